#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel stats of the cooperative kernels above 2048 points, one bench.py run each.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_large; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, bench args...
  tag=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --steps 30 --warmup 20 --no-cpu-baseline "$@" > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $OUT/$tag.log; return; }
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "== $tag ($*)"; grep -E "shw::" "$f" | cut -d, -f1-4 | head -4; cp "$f" $OUT/${tag}_kernel_stats.csv; }
}
run train4096 --mode train --points 4096 --batch 32
run train8192 --mode train --points 8192 --batch 16
run train3000 --mode train --points 3000 --batch 43
run p1loss4096 --p 1 --points 4096 --batch 32
run p1train4096 --mode train --p 1 --points 4096 --batch 32
run p1loss2048 --p 1
run loss3000 --points 3000 --batch 43
