#!/bin/bash
# Developer aid: A/B variant of the forward kernel (EPT=32 size class only) with extra -D flags; every other
# object is taken from the regular build.   tools/build_variant.sh <tag> [-DFLAG ...]
set -e
TAG=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/sphere-homeomorphic-wasserstein-distance-for-point-cloud-registration_amd/csrc
OUT=$ROOT/gpurun_variants; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSHW_DEV_ONLY_EPT=32 "$@" -c -o $OUT/fwd_$TAG.o $SRC/shw_ssw_fwd.hip
OBJS=$(ls $SRC/build/*.o | grep -v shw_ssw_fwd.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libshw_hip_$TAG.so $OUT/fwd_$TAG.o $OBJS
echo built $OUT/libshw_hip_$TAG.so
