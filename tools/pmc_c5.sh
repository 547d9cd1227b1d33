#!/bin/bash
# Dev: instruction / wait / LDS counters of item_sparse_topk_kernel on the C5 workload, per debug_mode:
#   bash tools/pmc_c5.sh <tag> [dbg list, e.g. 0,101,102,103,104,105]
TAG=$1; DBG=${2:-0}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc5_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5.py 262144 $DBG > $OUT/a.log 2>&1 || exit 2
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/tools/time_c5.py 262144 $DBG > $OUT/b.log 2>&1 || exit 3
cd $ROOT && python3 - <<PY
import csv, glob
modes = "$DBG".split(",")
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "item_sparse_topk" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), 0.0)
            acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for k, v in acc.items():
        d = [v[i] for i in sorted(v)]                 # 4 dispatches per mode (1 warm-up + 3)
        per = [sum(d[4 * m: 4 * m + 4]) / 4 for m in range(len(d) // 4)]
        print(k, " ".join(f"{modes[m] if m < len(modes) else m}:{x / 1e9:.3f}G" for m, x in enumerate(per)))
PY
