#!/bin/bash
# usage: ab_headline_arms.sh <rounds> <name=dir> ...   — as ab_pass_arms.sh, on the headline (10 000 timed steps, no large sections).  Wrote profiles/r04_rows_avx2_ab.log.
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for rnd in $(seq 1 $rounds); do
  for arm in "$@"; do
    name=${arm%%=*}; d=$R/${arm#*=}
    (cd $d && timeout -k 10 200 python bench.py --steps 10000 --warmup 200 --no-extra --no-large --no-cpu-baseline --streams 0 2>/dev/null) > $R/gpurun_out/abh_${name}_${rnd}.json
    python3 - $R/gpurun_out/abh_${name}_${rnd}.json $name $rnd <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
r = j["roofline"]; a = r["armed"]; c = j["config"]
print(sys.argv[2], sys.argv[3], "value", j["value"], "ms", j["ms_per_step"], "wait", a["dev_wait_us"], "body", a["dev_body_us"], "post_to_rows", a["host_post_to_rows_us"], "align_ms", round(c["align_ms"], 4), "filter_ms", round(c["filter_ms"], 4), "hostb", c.get("host_boundary_ms_per_scan") and round(c["host_boundary_ms_per_scan"], 4), flush=True)
PY
  done
done
