#!/bin/bash
# Developer aid (GPU box): SQ counters of the weighted general-path kernels (tools/general_time_w.py), one --pmc pass.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_general_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc -- python3 $ROOT/tools/general_time_w.py > $OUT/pmc.log 2>&1 || { echo pmc failed; tail -5 $OUT/pmc.log; exit 1; }
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
echo "counter file: $f"
python3 - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name") or r.get("Kernel Name")
    if "ssw_general" not in name: continue
    rows[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    n = len(c["SQ_WAVES"])
    avg = {m: sum(v) / len(v) for m, v in c.items()}
    print(k, "launches", n)
    print("   ", {m: "%.4g" % v for m, v in avg.items()})
    if avg.get("SQ_WAVES"):
        w = avg["SQ_WAVES"]
        print("    per wave: VALU %.0f  LDS %.0f  SALU %.0f  wave cycles %.0f  wait-any %.0f  wait-LDS %.0f" % (
            avg["SQ_INSTS_VALU"] / w, avg["SQ_INSTS_LDS"] / w, avg["SQ_INSTS_SALU"] / w, avg["SQ_WAVE_CYCLES"] / w,
            avg["SQ_WAIT_INST_ANY"] / w, avg["SQ_WAIT_INST_LDS"] / w))
PY
