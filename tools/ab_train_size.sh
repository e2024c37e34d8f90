#!/bin/bash
# runs on the GPU box: training-step bench at a given point count for every variant library ("base" = the in-tree
# build), twice, interleaved.   tools/ab_train_size.sh <points> <variant> [<variant> ...]
PTS=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset SHW_LIB_PATH; else export SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so; fi
  python bench.py --mode train --steps 100 --warmup 50 --points $PTS 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'N=$PTS train ms/step %.4f' % d['ms_per_step'])"
done; done
