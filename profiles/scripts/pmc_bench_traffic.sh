#!/bin/bash
# HBM traffic per launch of the dominant kernel of the bench workload (k_ndt_pass<0, true> at cfg2 size), collected exactly
# as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes with
# --kernel-trace only; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream), WRITE_SIZE as read; KB -> bytes.
# Writes gpurun_out/r02_pmc_traffic.json (copy it to profiles/).  Run from the repo root on the GPU box.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python3 $R/bench.py --steps 40 --warmup 5 --ramp-seconds 0.05 --no-cpu-baseline --no-large --streams 0 > $R/gpurun_out/pmcb_$c.log 2>&1
  python3 $R/profiles/scripts/pmc_summarize.py $(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmcb_$c.csv
done
python3 - <<PY
import json
def get(path, kern, ctr):
    for line in open(path):
        k, c, n, v = line.rstrip().rsplit(",", 3)
        if k.startswith(kern) and c == ctr:
            return int(n), float(v)
    return 0, 0.0
out = {}
for kern, key in (("k_ndt_pass<0, true>", "k_ndt_pass_kind0"), ("k_ndt_pass_armed<true>", "k_ndt_pass_armed")):
    nf, f = get("$R/gpurun_out/pmcb_FETCH_SIZE.csv", kern, "FETCH_SIZE")
    nw, w = get("$R/gpurun_out/pmcb_WRITE_SIZE.csv", kern, "WRITE_SIZE")
    out[key + "_bytes_per_launch"] = int((2.0 * f + w) * 1024)
    out[key + "_fetch_size_kb"] = f; out[key + "_write_size_kb"] = w; out[key + "_launches"] = nf
out["note"] = "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, mean per launch over the bench.py run (cfg2 size); separate --pmc passes"
json.dump(out, open("$R/gpurun_out/r02_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
