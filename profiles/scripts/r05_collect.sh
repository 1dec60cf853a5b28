#!/bin/bash
# usage: bash profiles/scripts/r05_collect.sh <tag>     (from the repo root on the GPU box; writes gpurun_out/<tag>_*)
# Round-5 evidence (same collection as r04_collect.sh, plus the phase stamps and the kernel timeline of the 2 M-point build) for the build as shipped, every counter group in its own pass (kernel trace only beside --pmc):
#   1. bench line of the default command                            -> <tag>_bench.json
#   2. kernel trace + stats of the bench command                    -> <tag>_bench_kernel_stats.csv, <tag>_bench_traced_run.json
#   3. FETCH_SIZE / WRITE_SIZE of the bench's pass kernels          -> <tag>_pmc_traffic.json
#   4. cfg5-sized pass: kernel trace + stats, then FETCH_SIZE, WRITE_SIZE, TCC hit/miss, SQ VALU group
#                                                                   -> <tag>_cfg5_pass_kernel_stats.csv, <tag>_cfg5_pass_pmc.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
cd /tmp && export TMPDIR=/tmp
echo "[r05_collect] bench" >&2
python3 $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
echo "[r05_collect] bench trace" >&2
rm -rf /tmp/btrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/btrace -- python3 $R/bench.py --steps 2000 --warmup 20 --ramp-seconds 0.5 --no-cpu-baseline --no-large --streams 0 > $R/gpurun_out/${tag}_bench_traced_run.json 2> $R/gpurun_out/${tag}_bench_trace.err
cp $(find /tmp/btrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_bench_kernel_stats.csv
echo "[r05_collect] bench pmc traffic" >&2
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcb_$c -- python3 $R/bench.py --steps 40 --warmup 5 --ramp-seconds 0.05 --no-cpu-baseline --no-large --streams 0 > $R/gpurun_out/pmcb_$c.log 2>&1
  python3 $R/profiles/scripts/pmc_summarize.py $(find /tmp/pmcb_$c -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmcb_$c.csv
done
python3 - <<PY
import json
def get(path, kern, ctr):
    for line in open(path):
        k, c, n, v = line.rstrip().rsplit(",", 3)
        if k.startswith(kern) and c == ctr:
            return int(n), float(v)
    return 0, 0.0
out = {}
for kern, key in (("k_ndt_pass<0, true>", "k_ndt_pass_kind0"), ("k_ndt_pass_armed<true>", "k_ndt_pass_armed")):
    nf, f = get("$R/gpurun_out/pmcb_FETCH_SIZE.csv", kern, "FETCH_SIZE")
    nw, w = get("$R/gpurun_out/pmcb_WRITE_SIZE.csv", kern, "WRITE_SIZE")
    out[key + "_bytes_per_launch"] = int((2.0 * f + w) * 1024)
    out[key + "_fetch_size_kb"] = f; out[key + "_write_size_kb"] = w; out[key + "_launches"] = nf
out["note"] = "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, mean per launch over the bench.py run (cfg2 size); separate --pmc passes"
json.dump(out, open("$R/gpurun_out/${tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
echo "[r05_collect] cfg5 pass trace" >&2
rm -rf /tmp/c5trace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5trace -- python3 $R/profiles/scripts/pmc_pass.py > $R/gpurun_out/${tag}_cfg5_trace.log 2>&1
cp $(find /tmp/c5trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_cfg5_pass_kernel_stats.csv
: > $R/gpurun_out/${tag}_cfg5_pass_pmc.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  echo "[r05_collect] cfg5 pmc $grp" >&2
  d=/tmp/pmc_c5_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/profiles/scripts/pmc_pass.py > $R/gpurun_out/${tag}_cfg5_pmc.log 2>&1 || { echo "group $grp failed" >&2; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 $R/profiles/scripts/pmc_summarize.py $f k_ndt_pass >> $R/gpurun_out/${tag}_cfg5_pass_pmc.csv
done
cat $R/gpurun_out/${tag}_cfg5_pass_pmc.csv
echo "[r05_collect] big filter" >&2
bash $R/profiles/scripts/r03_big_filter_pmc.sh ${tag} > $R/gpurun_out/${tag}_big_filter_collect.log 2>&1 || echo "big filter collection failed" >&2
echo "[r05_collect] front-end loop trace" >&2
bash $R/profiles/scripts/fe_loop_trace.sh ${tag} > $R/gpurun_out/${tag}_fe_loop_collect.log 2>&1 || echo "front-end loop trace failed" >&2
echo "[r05_collect] headline-only trace" >&2
rm -rf /tmp/htrace
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/htrace -- python3 $R/bench.py --steps 2000 --warmup 20 --ramp-seconds 0.5 --no-cpu-baseline --no-large --streams 0 --no-extra > $R/gpurun_out/${tag}_headline_traced_run.json 2> $R/gpurun_out/${tag}_headline_trace.err
cp $(find /tmp/htrace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_headline_kernel_stats.csv
echo "[r05_collect] 2 M-point build: phase stamps, kernel timeline, LDS counters" >&2
(python3 $R/profiles/scripts/vi_stamps.py; LEAF=1.3 python3 $R/profiles/scripts/vi_stamps.py) 2>/dev/null | grep -v amdgpu.ids > $R/gpurun_out/${tag}_vi_stamps.log
: > $R/gpurun_out/${tag}_big_filter_timeline.txt
for leaf in 0.3 1.3; do
  rm -rf /tmp/bf_tl
  LEAVES=$leaf NO_TARGET=1 REPS=20 rocprofv3 --kernel-trace --output-format csv -d /tmp/bf_tl -- python3 $R/profiles/scripts/big_filter.py > /dev/null 2>&1
  echo "# 2 M points, leaf $leaf m: the last three calls" >> $R/gpurun_out/${tag}_big_filter_timeline.txt
  python3 $R/profiles/scripts/big_filter_timeline.py /tmp/bf_tl >> $R/gpurun_out/${tag}_big_filter_timeline.txt
done
cd $R && bash profiles/scripts/pmc_big_filter_lds.sh > gpurun_out/${tag}_big_filter_lds_pmc.csv 2>&1 || echo "LDS counters failed" >&2
