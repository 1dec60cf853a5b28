#!/bin/bash
# A/B of the pre-armed first pass in a single FrontEnd::Update (front_end.hip: prearm_single): the scan-by-scan loop with LSA_FE_PREARM=1 (default)
# and 0, processes in rotating order on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
orders=("1 0" "0 1" "1 0" "0 1")
for rnd in 0 1 2 3; do
  for arm in ${orders[$rnd]}; do
    echo "== LSA_FE_PREARM=$arm round $rnd"
    LSA_FE_PREARM=$arm timeout -k 10 300 python $R/profiles/scripts/fe_loop_run.py 2>/dev/null | tail -3
  done
done
