#!/bin/bash
# VALU / SALU / LDS instructions of the CSR top-K kernel per cumulative early exit (dev: needs make ABLATE=1)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc5_phases; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5.py 262144 101,102,103,104,105,0 0 > $OUT/a.log 2>&1 || exit 2
cd $ROOT && python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/a/*/*_counter_collection.csv"))[0]
rows = {}
for row in csv.DictReader(open(f)):
    if "item_sparse_topk" in row["Kernel_Name"]:
        rows.setdefault(int(row["Dispatch_Id"]), {}).setdefault(row["Counter_Name"], 0.0)
        rows[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
ids = sorted(rows)
names = ["101 plan", "102 +loads", "103 +insert", "104 +scan", "105 +select", "0 all"]
for g in range(0, len(ids), 4):
    r = rows[ids[g + 1]] if g + 1 < len(ids) else rows[ids[g]]
    print(names[g // 4] if g // 4 < len(names) else g, {k: round(v / 262144) for k, v in r.items()})
PY
