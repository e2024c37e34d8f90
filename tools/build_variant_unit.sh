#!/bin/bash
# Developer aid: A/B variant of ONE translation unit with extra -D flags; every other object is taken from the
# regular build.   tools/build_variant_unit.sh <tag> <unit (e.g. shw_ssw_coop)> [-DFLAG ...]
set -e
TAG=$1; UNIT=$2; shift; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/sphere-homeomorphic-wasserstein-distance-for-point-cloud-registration_amd/csrc
OUT=$ROOT/gpurun_variants; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c -o $OUT/${UNIT}_$TAG.o $SRC/$UNIT.hip
OBJS=$(ls $SRC/build/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libshw_hip_$TAG.so $OUT/${UNIT}_$TAG.o $OBJS
echo built $OUT/libshw_hip_$TAG.so
