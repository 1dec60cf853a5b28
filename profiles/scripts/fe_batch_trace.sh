#!/bin/bash
# kernel timeline of the batch front end (profiles/scripts/fe_batch_run.py) under rocprofv3 --kernel-trace:
#   gpurun_out/<tag>_fe_batch_timeline.txt   start / end / duration (us) / queue / kernel of ~400 dispatches in the steady state
#   gpurun_out/<tag>_fe_batch_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-fe_batch}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fetrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fetrace -- python3 $R/profiles/scripts/fe_batch_run.py > $R/gpurun_out/${tag}_fe_batch_trace.log 2>&1
cp $(find /tmp/fetrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_fe_batch_kernel_stats.csv
f=$(find /tmp/fetrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/${tag}_fe_batch_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-420:-20]   # the last run's steady state
t0 = int(rows[0]["Start_Timestamp"])
print("# start_us end_us dur_us queue kernel   (queue = HIP stream: one is the registration + local map, the other the frame filter)")
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f %7.1f q%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
