# usage: pmc_summarize.py <counter_collection.csv> [kernel substring]  -> mean per dispatch of every counter
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"]
    if sub and sub not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("%s,%s,%d,%.6g" % (k, c, len(v), sum(v) / len(v)))
