#!/bin/bash
# 2 M-point VoxelFilter (0.3 m) and SetInputTarget: kernel trace + FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
# -> gpurun_out/<tag>_big_filter_kernel_stats.csv, <tag>_big_filter_pmc.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bf_trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bf_trace -- python3 $R/profiles/scripts/big_filter.py > $R/gpurun_out/${tag}_big_filter.log 2>&1
cp $(find /tmp/bf_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_big_filter_kernel_stats.csv
: > $R/gpurun_out/${tag}_big_filter_pmc.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/bf_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/bf_$c -- python3 $R/profiles/scripts/big_filter.py > $R/gpurun_out/${tag}_big_filter_pmc.log 2>&1
  python3 $R/profiles/scripts/pmc_summarize.py $(find /tmp/bf_$c -name "*counter_collection.csv" | head -1) >> $R/gpurun_out/${tag}_big_filter_pmc.csv
done
cat $R/gpurun_out/${tag}_big_filter_pmc.csv
