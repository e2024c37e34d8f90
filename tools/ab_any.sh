#!/bin/bash
# runs on the GPU box: bench.py with the given extra args for every variant library, twice, interleaved
# usage: tools/ab_any.sh "<bench args>" <variant> [<variant> ...]
ARGS=$1; shift
for round in 1 2; do
for v in "$@"; do
  SHW_BENCH_SKIP_PARITY=1 SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so python bench.py $ARGS --no-cpu-baseline --steps 100 --warmup 50 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'ms/step %.4f' % d['ms_per_step'])"
done; done
