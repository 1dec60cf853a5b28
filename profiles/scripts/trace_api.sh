#!/bin/bash
# usage: trace_api.sh <tag> <script> [args]: kernel trace + HIP runtime API trace (no counters) to line host launches up with kernel starts
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
script=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tapi_$tag
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/tapi_$tag -- python3 $R/$script "$@" > $R/gpurun_out/tapi_$tag.log 2>&1
cp $(find /tmp/tapi_$tag -name "*kernel_trace.csv" | head -1) $R/gpurun_out/tapi_${tag}_kernel_trace.csv
cp $(find /tmp/tapi_$tag -name "*hip_api_trace.csv" | head -1) $R/gpurun_out/tapi_${tag}_hip_api_trace.csv
