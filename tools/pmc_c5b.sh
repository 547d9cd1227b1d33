#!/bin/bash
# More SQ counters of the CSR top-K kernel on the C5 workload (dev tool): bash tools/pmc_c5b.sh <tag>
TAG=$1
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc5b_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $OUT/avail.txt | sort -u > $OUT/sq_names.txt
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5.py 262144 0 0 > $OUT/a.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/tools/time_c5.py 262144 0 0 > $OUT/b.log 2>&1 || exit 3
cd $ROOT && python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "item_sparse_topk" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(k, sum(v) / 4, "per launch (4 launches)")
PY
