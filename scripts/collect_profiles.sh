#!/bin/bash
# Round profiles on the GPU box: bench lines, kernel-trace summary of the bench command, hardware counters of the
# hot kernels (separate --pmc passes, never combined with trace domains other than --kernel-trace).
# usage: scripts/collect_profiles.sh OUTDIR   (run from the repo root; rocprofv3 runs from /tmp)
set -u
R=$(pwd); O=$R/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --steps 5 --warmup 2 --permute --no-cpu-baseline > $O/bench_permuted.json 2> /dev/null
timeout 600 python bench.py --steps 5 --warmup 2 --jitter 0.2 --no-cpu-baseline --no-pcie > $O/bench_jitter.json 2> /dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --mesh-n 100 --no-cpu-baseline --no-pcie > $O/bench_c2_n100.json 2> /dev/null
timeout 600 python bench.py --steps 3 --warmup 1 --pc jacobi --no-cpu-baseline --no-pcie > $O/bench_pc_jacobi.json 2> /dev/null
FEMO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_forced_dist_1rank.json 2> /dev/null
timeout 300 python scripts/run_nonlinear_c5.py > $O/config5_nonlinear_n2236.json 2> /dev/null
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie > $O/bench_under_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 5 k_spmv_sell 8 > $O/bench_kernel_stats.csv
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_rocprofv3.csv 2>/dev/null
rm -rf $O/trace
if [ "${SKIP_PMC:-0}" = "1" ]; then ls -la $O; exit 0; fi
for V in "215" "215 permute"; do
  T=$(echo $V | tr ' ' '_')
  for P in "FETCH_SIZE" "WRITE_SIZE L2CacheHit" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE VALUBusy MemUnitBusy MemUnitStalled"; do
    D=$O/pmc_${T}_$(echo $P | cut -d" " -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_pc_kernels.py $V > $D.log 2>&1)
  done
  python3 scripts/pmc_table.py $O/pmc_kernels_n$T.csv $O/pmc_${T}_FETCH_SIZE $O/pmc_${T}_WRITE_SIZE $O/pmc_${T}_SQ_WAVES $O/pmc_${T}_GRBM_GUI_ACTIVE > /dev/null
  rm -rf $O/pmc_${T}_*
done
python3 scripts/pmc_traffic.py $O/pmc_traffic.json $O/pmc_kernels_n215.csv $O/pmc_kernels_n215_permute.csv > /dev/null
ls -la $O
