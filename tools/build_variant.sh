#!/bin/bash
# Developer aid: build a variant of ONE translation unit with extra -D flags and link it with the other objects of the
# current build into gpurun_variants/<name>.so (selected at run time with SHW_LIB_PATH).
#   tools/build_variant.sh <name> <unit> [-D...]
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$(ls -d $ROOT/sphere*_amd/csrc)
mkdir -p $ROOT/gpurun_variants/obj
NOSLP=-fno-slp-vectorize; [ "$UNIT" = shw_ssw_grad2_m32 ] && NOSLP=      # as the Makefile (SLP_UNITS)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function $NOSLP "$@" \
  -c -o $ROOT/gpurun_variants/obj/$NAME.o $CSRC/$UNIT.hip
OTHERS=$(ls $CSRC/build/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/gpurun_variants/$NAME.so $ROOT/gpurun_variants/obj/$NAME.o $OTHERS
echo built $ROOT/gpurun_variants/$NAME.so
