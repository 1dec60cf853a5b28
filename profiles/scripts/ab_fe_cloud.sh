#!/bin/bash
# A/B of how FrontEnd::Update protects a key-frame's read of the caller's device cloud (front_end.hip: cloud_mode): the scan-by-scan
# front-end loop with the copy made by the frame filter's first kernel (keep, default), with the registration's stream ordered behind
# the map update's read (event) and with neither (none: the pre-round-5 contract); processes in rotating order on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
orders=("keep event none" "none keep event" "event none keep" "keep none event")
for rnd in 0 1 2 3; do
  for arm in ${orders[$rnd]}; do
    echo "== LSA_FE_CLOUD=$arm round $rnd"
    LSA_FE_CLOUD=$arm timeout -k 10 300 python $R/profiles/scripts/fe_loop_run.py 2>/dev/null | tail -3
  done
done
