#!/bin/bash
# usage: ab_arms.sh <rounds> <script relative to an arm's root> <name=dir> ...   — runs the script once per arm and round, arms in rotating order,
# each in its own process from the arm's own tree (two library builds side by side on one box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; script=$2; shift; shift
arms=("$@")
n=${#arms[@]}
for rnd in $(seq 1 $rounds); do
  for k in $(seq 0 $((n - 1))); do
    arm=${arms[$(((k + rnd) % n))]}
    name=${arm%%=*}; d=$R/${arm#*=}
    echo "== $name round $rnd"
    (cd $d && timeout -k 10 300 python $script 2>/dev/null)
  done
done
