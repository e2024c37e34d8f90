#!/bin/bash
# Developer aid (GPU box): SQ counters of the two-wave loss kernel on a full class (N=2048, SHW_FORWARD_KERNEL=twowave) and
# on partially filled ones (N=2000, 1280, 1200): where does the partially-filled-class code lose its ~30 %?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_partial; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 2048 2000 1280 1200; do
  export SHW_FORWARD_KERNEL=twowave
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_$n -- python3 $ROOT/tools/one_size.py $n > $OUT/pmc_$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $OUT/pmc_$n.log; }
  f=$(find $OUT/pmc_$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" $n <<'PY'
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name") or r.get("Kernel Name")
    if "ssw_forward" not in name: continue
    rows[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    avg = {m: sum(v) / len(v) for m, v in c.items()}
    w = avg["SQ_WAVES"]
    print("N=%s %s: per wave VALU %.0f LDS %.0f SALU %.0f wave-cycles %.0f wait-any %.0f wait-LDS %.0f busy %.3g" % (
        sys.argv[2], k[-40:], avg["SQ_INSTS_VALU"] / w, avg["SQ_INSTS_LDS"] / w, avg["SQ_INSTS_SALU"] / w,
        avg["SQ_WAVE_CYCLES"] / w, avg["SQ_WAIT_INST_ANY"] / w, avg["SQ_WAIT_INST_LDS"] / w, avg["SQ_BUSY_CYCLES"]))
PY
done
