#!/bin/bash
# usage: ab_env.sh <rounds> <script> <VAR> <value> <value> ...   — one process per value and round, values in rotating order, on this tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; script=$2; var=$3; shift; shift; shift
vals=("$@"); n=${#vals[@]}
for rnd in $(seq 1 $rounds); do
  for k in $(seq 0 $((n - 1))); do
    v=${vals[$(((k + rnd) % n))]}
    echo "== $var=$v round $rnd"
    (cd $R && env $var=$v timeout -k 10 300 python $script 2>/dev/null)
  done
done
