#!/bin/bash
# Wait-state breakdown of the fused kernel (dev tool): bash tools/pmc_wait.sh <tag> [debug_mode]
TAG=$1; DEV="--dev debug_mode=${2:-0}"
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmcw_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/bench.py --no-cpu --no-extra --steps 3 --warmup 1 $DEV > $OUT/a.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/bench.py --no-cpu --no-extra --steps 3 --warmup 1 $DEV > $OUT/b.log 2>&1 || exit 3
cd $ROOT && python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
    acc = {}
    for row in csv.DictReader(open(f)):
        if "score_topk" in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in acc.items():
        # each dispatch appears once per counter after aggregation across dimensions? sum / launches
        print(k, sum(v) / 4)
PY
