"""Kernel timeline of the last few 2 M-point filter calls out of a rocprofv3 --kernel-trace of profiles/scripts/big_filter.py (LEAVES=0.3):
start of each kernel relative to the call's first kernel, duration, gap to the previous kernel's end."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:28]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a call = the kernels up to and including k_filter_finalize (its first one differs by path: bounding box, a mark kernel, or the
# by-cell accumulate kernel)
calls, cur = [], []
for k in ks:
    if not k[0].startswith("k_"):
        continue
    cur.append(k)
    if k[0].startswith("k_filter_finalize"):
        calls.append(cur); cur = []
for c in calls[-3:]:
    t0 = c[0][1]
    prev = None
    print("call: %.1f us first kernel start -> last kernel end" % ((c[-1][2] - t0) / 1e3))
    for name, s, e in c:
        print("  %-28s start %6.1f  dur %6.1f  gap %5.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e
