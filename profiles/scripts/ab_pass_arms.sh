#!/bin/bash
# usage: ab_pass_arms.sh <rounds> <name=dir> ...   (from the repo root on the GPU box)
# Same-box A/B of library builds on the pass-kernel figures: each arm is a source tree (relative to the repo root: `.` or a tree built under
# scratch/ by `git archive <rev> | tar -x` + `make -C lidar-slam_amd/csrc [EXTRA=-D...]`), one `bench.py --steps 2000 --no-extra` process per arm
# and round, arms alternating.  Wrote profiles/r04_pass_setprio_ab.log, r04_pass_cell_per_lane_ab.log.
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for rnd in $(seq 1 $rounds); do
  for arm in "$@"; do
    name=${arm%%=*}; d=$R/${arm#*=}
    (cd $d && timeout -k 10 200 python bench.py --steps 2000 --warmup 100 --no-extra --no-cpu-baseline --streams 0 2>/dev/null) > $R/gpurun_out/ab_${name}_${rnd}.json
    python3 - $R/gpurun_out/ab_${name}_${rnd}.json $name $rnd <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
r = j["roofline"]; c5 = j["roofline_cfg5"]; w = j["wide_map"]
print(sys.argv[2], sys.argv[3], "value", j["value"], "armed body", r["armed"]["dev_body_us"], "isolated", r["isolated"]["us_per_launch"], "cfg5 us", c5["us_per_launch"], "align_ms", c5.get("align_ms"), "wide", w["random_order"]["us_per_launch"], w["cell_sorted"]["us_per_launch"], flush=True)
PY
  done
done
