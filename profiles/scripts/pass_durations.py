"""Split the k_ndt_pass<0> launches of a bench.py kernel trace into the ones of the timed pipeline (they run UNDER the
overlapped frame filter of the next scan) and the last 200, which are bench.py's isolated back-to-back roofline
measurement (`roofline.us_per_launch`).  usage: pass_durations.py <kernel_trace.csv> <out.json>"""
import csv, json, sys
import numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_ndt_pass<0" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
out = {"kernel": "k_ndt_pass<0, true>", "launches": int(d.size),
       "avg_us_all": round(float(d.mean()), 3),
       "avg_us_pipeline_under_overlapped_filter": round(float(d[:-200].mean()), 3),
       "avg_us_last_200_isolated_roofline_measurement": round(float(d[-200:].mean()), 3),
       "median_us_last_200": round(float(np.median(d[-200:])), 3),
       "note": "bench.py's roofline.us_per_launch (HIP events around 200 back-to-back launches) is the isolated figure; the "
               "kernel-stats average over the whole run is dominated by the pipelined launches, which share the GPU with the "
               "filter kernels of the next scan"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
