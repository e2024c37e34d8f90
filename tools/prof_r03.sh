#!/bin/bash
# Round-3 measurement batch (GPU box, via gpurun): everything DESIGN.md / profiles/README.md quote for this round.
#   tools/prof_r03.sh [part ...]      parts: bench headline general notebook sweep modes nonuniform partial   (default: all)
# Outputs under gpurun_out/r03/; the summaries are copied into profiles/ by hand afterwards.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
PARTS=${@:-bench headline general notebook sweep modes nonuniform partial}
has() { [[ " $PARTS " == *" $1 "* ]]; }

if has bench; then
  echo "== bench.py, the driver's arguments and the defaults"
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.log || tail -5 $OUT/bench_driver_args.log
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log || tail -5 $OUT/bench_default.log
  cut -c1-400 $OUT/bench_driver_args.json
fi
if has headline; then
  echo "== headline kernel: kernel trace + PMC passes (tools/profile.sh)"
  bash tools/profile.sh r03_d_forward > $OUT/profile_forward.log 2>&1; tail -3 $OUT/profile_forward.log
  bash tools/profile.sh r03_d_train --mode train > $OUT/profile_train.log 2>&1; tail -3 $OUT/profile_train.log
fi
if has general; then
  echo "== general path: times, kernel stats, SQ counters"
  timeout -k 10 300 python3 tools/general_time.py 2>&1 | grep -v amdgpu.ids > $OUT/general_time.txt; cat $OUT/general_time.txt
  for kind in weighted unequal; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/general_trace_$kind -- python3 tools/general_prof.py $kind train > $OUT/general_trace_$kind.log 2>&1
    f=$(find $OUT/general_trace_$kind -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/general_${kind}_kernel_stats.csv
  done
  bash tools/prof_general_pmc.sh > $OUT/general_pmc.txt 2>&1; tail -12 $OUT/general_pmc.txt
fi
if has notebook; then
  echo "== the notebooks' step (N=1200, L=100): eager / graph / graph-fused, and kernel traces"
  timeout -k 10 200 python3 tools/notebook_flow_time.py all 2>&1 | grep gradient-flow > $OUT/notebook_flow_time.txt
  SHW_SMALL_GRID=0 timeout -k 10 200 python3 tools/notebook_flow_time.py graph-fused 2>&1 | grep gradient-flow | sed 's/$/   [SHW_SMALL_GRID=0]/' >> $OUT/notebook_flow_time.txt
  timeout -k 10 200 python3 tools/notebook_flow_time.py all 1 2>&1 | grep gradient-flow >> $OUT/notebook_flow_time.txt
  cat $OUT/notebook_flow_time.txt
  for mode in eager graph-fused; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/notebook_trace_$mode -- python3 tools/notebook_flow_time.py $mode > $OUT/notebook_trace_$mode.log 2>&1
    f=$(find $OUT/notebook_trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/notebook_${mode}_kernel_stats.csv
  done
fi
if has sweep; then
  echo "== size sweeps"
  timeout -k 10 600 python3 tools/size_sweep.py 2,1 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep.txt; cat $OUT/size_sweep.txt
  timeout -k 10 300 python3 tools/size_sweep_mid.py 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep_mid.txt
  timeout -k 10 300 python3 tools/size_sweep_small.py 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep_small.txt
  timeout -k 10 300 python3 tools/size_sweep_p1.py 2>&1 | grep -v amdgpu.ids > $OUT/size_sweep_p1.txt
fi
if has modes; then
  echo "== bench.py secondary modes"
  for args in "--mode train" "--mode train --p 1" "--p 1" "--mode mirror" "--mode chamfer" "--mode config5 --steps 12"; do
    python3 bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 >> $OUT/bench_modes.jsonl
  done
  cut -c1-200 $OUT/bench_modes.jsonl
fi
if has nonuniform; then
  echo "== clouds that are not uniform in angle (ADVICE r2)"
  timeout -k 10 200 python3 tools/nonuniform_time.py 2>&1 | grep -v amdgpu.ids > $OUT/nonuniform.txt
  if [ -f gpurun_variants/fwd_runlen.so ]; then
    SHW_RUNLEN=1 SHW_LIB_PATH=$ROOT/gpurun_variants/fwd_runlen.so timeout -k 10 200 python3 tools/nonuniform_time.py 2>&1 | grep -v amdgpu.ids >> $OUT/nonuniform.txt
  fi
  cat $OUT/nonuniform.txt
fi
if has partial; then
  echo "== partially filled classes vs a full one (SQ counters of the two-wave loss kernel)"
  bash tools/prof_partial.sh > $OUT/prof_partial.txt 2>&1; cat $OUT/prof_partial.txt
fi
find $OUT -name "*.csv" -size +2M -delete
echo done
