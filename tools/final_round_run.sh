#!/bin/bash
# End-of-round GPU pass (run on the box via gpurun): full GPU suite, bench in every mode, size sweep, general-path
# kernel stats.  Everything lands under gpurun_out/final/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/gpu_tests.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cat $OUT/bench_default.json
for mode in train mirror chamfer config5; do
  python bench.py --mode $mode --no-cpu-baseline > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err; echo "bench $mode rc=$?"; cut -c1-400 $OUT/bench_$mode.json
done
python tools/size_sweep.py > $OUT/size_sweep.txt 2>&1; echo "sweep rc=$?"
python tools/general_time.py > $OUT/general_time.txt 2>&1; cat $OUT/general_time.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/general_trace -- python3 $ROOT/tools/general_time.py > $OUT/general_trace.log 2>&1; echo "general trace rc=$?"
f=$(find $OUT/general_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/general_kernel_stats.csv && cut -c1-150 $OUT/general_kernel_stats.csv | head -12
