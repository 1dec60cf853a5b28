#!/bin/bash
# kernel timeline of the scan-by-scan front-end loop (profiles/scripts/fe_loop_run.py) under rocprofv3 --kernel-trace:
#   gpurun_out/<tag>_fe_loop_timeline.txt, gpurun_out/<tag>_fe_loop_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-fe_loop}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fetrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fetrace -- python3 $R/profiles/scripts/fe_loop_run.py > $R/gpurun_out/${tag}_fe_loop_trace.log 2>&1
cp $(find /tmp/fetrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_fe_loop_kernel_stats.csv
f=$(find /tmp/fetrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/${tag}_fe_loop_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-330:-10]   # the last drive's steady state
t0 = int(rows[0]["Start_Timestamp"])
print("# start_us end_us dur_us queue kernel")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]))
PY
