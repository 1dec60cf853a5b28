#!/bin/bash
# usage: trace.sh <tag> <script> [args]: rocprofv3 --kernel-trace --stats of a script; kernel stats land in gpurun_out/trace_<tag>_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
script=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -- python3 $R/$script "$@" > $R/gpurun_out/trace_$tag.log 2>&1
cp $(find /tmp/trace_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/trace_${tag}_kernel_stats.csv; cp $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) $R/gpurun_out/trace_${tag}_kernel_trace.csv
cat $R/gpurun_out/trace_$tag.log | tail -8
cut -d, -f1-8 $R/gpurun_out/trace_${tag}_kernel_stats.csv | head -24
