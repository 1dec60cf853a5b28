#!/bin/bash
# usage: pmc_collect.sh <tag> <script> [args]   (run from the repo root on the GPU box)
# Three separate --pmc passes (FETCH_SIZE and WRITE_SIZE together exceed what the hardware collects at once) with
# --kernel-trace only, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes; summaries land in gpurun_out/pmc_<tag>.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
script=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_$tag.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  d=/tmp/pmc_${tag}_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/$script "$@" > $R/gpurun_out/pmc_${tag}.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/scripts/pmc_summarize.py $f k_ndt_pass >> $R/gpurun_out/pmc_$tag.csv
done
cat $R/gpurun_out/pmc_$tag.csv
