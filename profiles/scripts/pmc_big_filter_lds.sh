#!/bin/bash
# LDS / atomic / wave-cycle counters of the 2 M-point filter kernels (k_mark, k_scan, k_accum, finalize), one --pmc pass per group
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
  d=/tmp/pmcbf_$(echo $grp | tr ' ' '_' | cut -c1-50); rm -rf $d
  REPS=5 timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/profiles/scripts/big_filter.py > $R/gpurun_out/pmcbf.log 2>&1 || echo "failed $grp"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/scripts/pmc_summarize.py $f | grep "k_mark\|k_accum\|k_scan\|k_filter_finalize"
done
