#!/bin/bash
# Instruction counts of the workgroup-per-user CSR kernel on the even vs the uniformly drawn W (dev tool)
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc5_random; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5_random.py > $OUT/a.log 2>&1 || exit 2
cd $ROOT && python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/a/*/*_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "item_sparse" in r["Kernel_Name"]]
    per = {}
    for r in rows:
        per.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:40]), {}).setdefault(r["Counter_Name"], 0.0)
        per[(int(r["Dispatch_Id"]), r["Kernel_Name"][:40])][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(per):
        print(k, {c: round(v / 262144) for c, v in per[k].items()})
PY
