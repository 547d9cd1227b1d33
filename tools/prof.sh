#!/bin/bash
# Profile one bench.py workload on the GPU box: kernel trace + PMC passes (each counter group in its own run).
# Usage (from the repo root on the box): bash tools/prof.sh <tag> <kernel-regex> [bench.py args...]
#   e.g. bash tools/prof.sh r01_c4 item_item_kernel --config c4_ease_mind --steps 3 --warmup 1
set -u
TAG=$1; KREGEX=$2; shift 2
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu --no-extra $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $B > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $B > $OUT/pmc_fetch.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $B > $OUT/pmc_write.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $B > $OUT/pmc_sq.log 2>&1 || exit 4
cd $ROOT && python3 tools/summarize_prof.py $OUT "$KREGEX" "$*" > $OUT/summary.json && cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv
echo done
