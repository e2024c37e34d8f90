#!/bin/bash
# Developer aid (GPU box): kernel stats of loss-only launches at the given sizes.  tools/prof_sizes.sh 3000 4096
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  OUT=$ROOT/gpurun_out/prof_$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/one_size.py $n > $OUT.log 2>&1 || { echo "trace $n failed"; tail -5 $OUT.log; exit 1; }
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $n $f"
  [ -n "$f" ] && cut -c1-200 "$f" | head -4
done
