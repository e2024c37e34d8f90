#!/bin/bash
# runs on the GPU box: training-step bench of every variant library at several sizes, twice, interleaved
# usage: tools/ab_train.sh <variant> [<variant> ...]
for round in 1 2; do
for size in "2048 512" "1024 256" "512 256" "256 128" "1000 256" "2000 512"; do
for v in "$@"; do
  pts=${size% *}; sl=${size#* }
  SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so python bench.py --mode train --no-cpu-baseline --steps 100 --warmup 50 --points $pts --slices $sl 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', '$size', 'ms/step %.4f' % d['ms_per_step'])"
done; done; done
