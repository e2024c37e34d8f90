#!/bin/bash
# runs on the GPU box: training-step bench (config 3) of every variant library, twice, interleaved
for round in 1 2; do
for v in "$@"; do
  SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so python bench.py --mode train --steps 100 --warmup 50 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'ms/step %.4f' % d['ms_per_step'])"
done; done
