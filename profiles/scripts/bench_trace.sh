#!/bin/bash
# kernel-trace + stats of the bench command (no counters): gpurun_out/bench_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/btrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/btrace -- python3 $R/bench.py --steps 2000 --warmup 20 --ramp-seconds 0.5 --no-cpu-baseline --no-large --streams 0 > $R/gpurun_out/bench_trace.json 2> $R/gpurun_out/bench_trace.err
cp $(find /tmp/btrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/bench_kernel_stats.csv
cp $(find /tmp/btrace -name "*kernel_trace.csv" | head -1) $R/gpurun_out/bench_kernel_trace.csv
