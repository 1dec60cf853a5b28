#!/bin/bash
# ICP (configs[2]) under rocprofv3 --kernel-trace: profiles/scripts/icp_timing.py (27 ScanMatches per mode)
#   gpurun_out/<tag>_icp_iteration_durations.txt   kernel duration and gap to the next kernel, per iteration, last 3 ScanMatches per mode
#   gpurun_out/<tag>_icp_kernel_stats.csv
#   gpurun_out/<tag>_icp_timing.log                the same script untraced: ms per ScanMatch, searches per iteration, armed counts
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-icp}
python3 $R/profiles/scripts/icp_timing.py > $R/gpurun_out/${tag}_icp_timing.log 2>&1
LSA_ICP_NO_ARMED=1 python3 $R/profiles/scripts/icp_timing.py >> $R/gpurun_out/${tag}_icp_timing.log 2>&1
LSA_ICP_NO_ARMED=1 LSA_ICP_NO_SKIP=1 python3 $R/profiles/scripts/icp_timing.py >> $R/gpurun_out/${tag}_icp_timing.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/icptrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icptrace -- python3 $R/profiles/scripts/icp_timing.py > $R/gpurun_out/${tag}_icp_trace.log 2>&1
cp $(find /tmp/icptrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_icp_kernel_stats.csv
f=$(find /tmp/icptrace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/${tag}_icp_iteration_durations.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_gn_pass" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name, label in (("ILb0E", "ICPMANUAL"), ("ILb1E", "point-to-plane")):
    rr = [r for r in rows if name in r["Kernel_Name"] or ("<false>" in r["Kernel_Name"] and label == "ICPMANUAL") or ("<true>" in r["Kernel_Name"] and label != "ICPMANUAL")]
    for sm in range(3):
        seg = rr[len(rr) - 30 * (sm + 1): len(rr) - 30 * sm]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
        gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
        print(label, "kernel us per iteration (begin -> end of the dispatch; an armed kernel's includes its wait for the pose):", " ".join("%.1f" % x for x in d))
        print(label, "gap to the next kernel us:", " ".join("%.1f" % x for x in gaps))
PY
