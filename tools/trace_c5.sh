#!/bin/bash
# Kernel times of the C5 ranking call (dev tool): bash tools/trace_c5.sh <tag> [K] [users]
TAG=$1; KK=${2:-10}; NU=${3:-1048576}
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/trace5_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 $ROOT/tools/time_c5.py $NU 0 0 $KK 1 > $OUT/a.log 2>&1 || exit 2
grep "item_sparse" $OUT/a/*/*_kernel_stats.csv | cut -d, -f1-4
grep "^dbg" $OUT/a.log
