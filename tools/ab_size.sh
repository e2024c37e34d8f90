#!/bin/bash
# runs on the GPU box: forward bench at a given point count for every variant library, twice, interleaved
# usage: tools/ab_size.sh <points> <variant> [<variant> ...]
PTS=$1; shift
for round in 1 2; do
for v in "$@"; do
  SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so python bench.py --no-cpu-baseline --steps 100 --warmup 50 --points $PTS 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'N=$PTS ms/step %.4f parity %.1e' % (d['ms_per_step'], d['parity_rel_err']))"
done; done
