import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:28]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
calls, cur = [], []
for k in ks:
    if not k[0].startswith("k_"): continue
    cur.append(k)
    if k[0].startswith("k_filter_finalize"):
        calls.append(cur); cur = []
# frame-filter calls: those whose kernels are the <512> variants
ff = [c for c in calls if any("512" in k[0] for k in c)]
import statistics
for c in ff[-2:]:
    t0 = c[0][1]
    print("call: %.1f us" % ((c[-1][2] - t0) / 1e3))
    for name, s, e in c:
        print("  %-30s start %6.1f dur %6.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3))
print("median chain over %d calls: %.1f us" % (len(ff), statistics.median((c[-1][2] - c[0][1]) / 1e3 for c in ff)))
