#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r1}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 20 --no-cpu-baseline $@"
cd $ROOT
echo "== kernel trace" 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -20 $OUT/trace.log; exit 1; }
echo "== pmc SQ"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || { echo pmc_sq failed; tail -20 $OUT/pmc_sq.log; exit 1; }
echo "== pmc SQ2"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 || { echo pmc_sq2 failed; tail -20 $OUT/pmc_sq2.log; }
echo "== pmc SQ3"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq3.log 2>&1 || { echo pmc_sq3 failed; tail -5 $OUT/pmc_sq3.log; }
echo "== pmc SQ4"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INSTS_WAVE32_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVE32_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH --output-format csv -d $OUT/pmc_sq4 -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq4.log 2>&1 || { echo pmc_sq4 failed; tail -5 $OUT/pmc_sq4.log; }
echo "== pmc FETCH"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || { echo pmc_fetch failed; tail -20 $OUT/pmc_fetch.log; }
echo "== pmc WRITE"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || { echo pmc_write failed; tail -20 $OUT/pmc_write.log; }
find $OUT -name "*.csv" | head -50
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
