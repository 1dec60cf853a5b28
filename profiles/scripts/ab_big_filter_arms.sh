#!/bin/bash
# usage: ab_big_filter_arms.sh <rounds> <name=dir> ...   — as ab_pass_arms.sh, on the 2 M-point filter / SetInputTarget (profiles/scripts/big_filter.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; shift
for rnd in $(seq 1 $rounds); do
  for arm in "$@"; do
    name=${arm%%=*}; d=$R/${arm#*=}
    echo "== $name $rnd"; (cd $d && REPS=200 timeout -k 10 200 python profiles/scripts/big_filter.py 2>/dev/null | cut -c1-110)
  done
done
