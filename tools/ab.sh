#!/bin/bash
# runs on the GPU box: bench every variant library twice, interleaved
for round in 1 2; do
for v in "$@"; do
  SHW_BENCH_SKIP_PARITY=1 SHW_LIB_PATH=$PWD/gpurun_variants/libshw_hip_$v.so python bench.py --no-cpu-baseline --steps 100 --warmup 50 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v', 'ms/step %.4f kernel_ms %.4f parity %.2e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['parity_rel_err']))"
done; done
