#!/bin/bash
# four quick --pmc passes over the 2 M-point derivative pass (L1 accesses, instruction counts, TCP busy / pending stalls, cycles, TA busy)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
  d=/tmp/pmc2_$(echo $grp | tr ' ' '_' | cut -c1-50); rm -rf $d
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/profiles/scripts/pmc_pass.py > $R/gpurun_out/pmc2.log 2>&1 || echo "failed $grp"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/scripts/pmc_summarize.py $f k_ndt_pass
done
