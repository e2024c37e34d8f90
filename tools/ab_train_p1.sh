#!/bin/bash
# p = 1 training step at <points> (batch 131072 / points) for every variant library, twice, interleaved
PTS=$1; shift
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset SHW_LIB_PATH; else export SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so; fi
  python bench.py --mode train --p 1 --steps 50 --warmup 20 --points $PTS --batch $((131072/PTS)) 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'N=$PTS p=1 train ms/step %.4f' % d['ms_per_step'])"
done; done
