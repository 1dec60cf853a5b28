#!/bin/bash
# usage: pmc_collect_l1.sh <tag> <script> [args]   (run from the repo root on the GPU box)
# The address / L1 side of a kernel: texture-addresser (TA) and vector-L1 (TCP) busy and stall counters, VMEM / LDS issue counters —
# (at most two TA / TCP / TD counters per pass: four exceed what the hardware collects at once and rocprofv3 then aborts and hangs)
# one --pmc pass per group with --kernel-trace only; the per-dispatch means land in gpurun_out/pmc_l1_<tag>.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
script=$1; shift
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_l1_$tag.csv
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  d=/tmp/pmc_l1_${tag}_$(echo $grp | tr ' ' '_' | cut -c1-60)
  rm -rf $d
  if timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/$script "$@" > $R/gpurun_out/pmc_l1_${tag}.log 2>&1; then
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 $R/profiles/scripts/pmc_summarize.py $f k_ndt_pass >> $R/gpurun_out/pmc_l1_$tag.csv
  else
    echo "group failed: $grp" >> $R/gpurun_out/pmc_l1_$tag.csv; tail -3 $R/gpurun_out/pmc_l1_${tag}.log >> $R/gpurun_out/pmc_l1_$tag.csv
  fi
done
cat $R/gpurun_out/pmc_l1_$tag.csv
