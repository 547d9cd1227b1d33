#!/bin/bash
# Instruction mix + wait states of the CSR top-K kernel on the C5 workload (dev tool): bash tools/pmc_c5c.sh <tag> [K]
TAG=$1; KK=${2:-10}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc5c_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5.py 262144 0 0 $KK > $OUT/a.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/tools/time_c5.py 262144 0 0 $KK > $OUT/b.log 2>&1 || exit 3
cd $ROOT && python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "item_sparse" in row["Kernel_Name"]:
            kn = "wave" if "wave" in row["Kernel_Name"] else "workgroup"
            acc.setdefault((kn, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(kn, k, sum(v) / 4 / 262144, "per user of the launch (4 launches of 262144 users)")
PY
