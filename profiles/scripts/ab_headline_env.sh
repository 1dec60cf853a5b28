#!/bin/bash
# usage: ab_headline_env.sh <rounds> <name=ENV=VALUE | name=> ...   — ONE tree, the arm chosen by an environment switch of the library
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
lscpu | grep -E "Model name|Flags" | cut -c1-400 | sed 's/Flags.*avx512f.*/Flags: ... avx512f present/'
for rnd in $(seq 1 $rounds); do
  for arm in "$@"; do
    name=${arm%%=*}; envs=${arm#*=}
    (cd $R && env $envs timeout -k 10 200 python bench.py --steps 10000 --warmup 200 --no-extra --no-large --no-cpu-baseline --streams 0 2>/dev/null) > $R/gpurun_out/abe_${name}_${rnd}.json
    python3 - $R/gpurun_out/abe_${name}_${rnd}.json $name $rnd <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
r = j["roofline"]; a = r["armed"]; c = j["config"]
print(sys.argv[2], sys.argv[3], "value", j["value"], "ms", j["ms_per_step"], "wait", a["dev_wait_us"], "body", a["dev_body_us"], "post_to_rows", a["host_post_to_rows_us"], "align_ms", round(c["align_ms"], 4), flush=True)
PY
  done
done
