"""Kernel (and fill / copy) timeline around the two one-time steps of a cold FrontEnd drive: the first key-frame (unfiltered
level-N build: k_lm_points<10, 1, false>) and the tenth (switch to the filtered map: k_lm_points<4, 0, false>).
usage: fe_cold_timeline.py <rocprofv3 output dir>   (a --kernel-trace run of fe_cold_drive_calls.py)"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
mc = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
for r in (csv.DictReader(open(mc[0])) if mc else []):
    rows.append({"Kernel_Name": "copy_" + r.get("Direction", "?") + "_%sB" % r.get("Size", "?"), "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"]})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    if n.startswith("copy_"): return n
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
ks = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
def around(marker, which, before_us=200.0, after_us=400.0):
    idx = [i for i, k in enumerate(ks) if k[0].startswith(marker)]
    if not idx: return
    # group marker launches closer than 1 ms into one event; take the `which`-th group from the end
    groups, cur = [], [idx[0]]
    for i in idx[1:]:
        if ks[i][1] - ks[cur[-1]][1] < 1_000_000: cur.append(i)
        else: groups.append(cur); cur = [i]
    groups.append(cur)
    g = groups[which]
    t0 = ks[g[0]][1]
    print("## around %s (group %d of %d)" % (marker, which % len(groups), len(groups)))
    prev_end = None
    for name, s, e in ks:
        if s < t0 - before_us * 1e3 or s > t0 + after_us * 1e3: continue
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print("  %-34s start %8.1f dur %6.1f  idle before %6.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, max(gap, 0.0)))
        prev_end = max(prev_end or e, e)
around("k_lm_points<10, 1, false>", -1)
around("k_lm_points<4, 0, false>", -1)
