#!/bin/bash
# kernel trace + stats of the device-resident FrontEnd loop (profiles/scripts/fe_timing.py): gpurun_out/<tag>_fe_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-fe}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fetrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fetrace -- python3 $R/profiles/scripts/fe_timing.py > $R/gpurun_out/${tag}_fe_trace.log 2>&1
cp $(find /tmp/fetrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_fe_kernel_stats.csv
