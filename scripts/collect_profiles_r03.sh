#!/bin/bash
# Round-3 profiles on the GPU box: the driver's bench line (with the configs legs), its variants, kernel-trace summaries
# of the bench command and of the shell / nonlinear configurations, hardware counters of the hot kernels (separate --pmc
# passes, only ever combined with --kernel-trace).
# usage: scripts/collect_profiles_r03.sh OUTDIR   (run from the repo root)
set -u
R=$(pwd); O=$R/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --steps 5 --warmup 2 --permute --no-cpu-baseline --no-configs > $O/bench_permuted.json 2> /dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --permute --reorder --no-cpu-baseline --no-configs --no-pcie > $O/bench_permuted_reordered.json 2> /dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --reorder --force-morton --no-cpu-baseline --no-configs --no-pcie > $O/bench_morton.json 2> /dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --jitter 0.2 --no-cpu-baseline --no-configs --no-pcie > $O/bench_jitter.json 2> /dev/null
timeout 600 python bench.py --steps 3 --warmup 1 --pc jacobi --no-cpu-baseline --no-configs --no-pcie > $O/bench_pc_jacobi.json 2> /dev/null
FEMO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/bench_forced_dist_1rank.json 2> /dev/null
# kernel traces: the bench command itself (headline leg only: the trace of the other meshes would drown it), C3, C5
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-configs > $O/bench_under_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 5 k_spmv_sell 8 > $O/bench_kernel_stats.csv
python3 scripts/trace_gaps.py $O/trace 200 > $O/bench_trace_gaps.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_rocprofv3.csv 2>/dev/null
rm -rf $O/trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -- python3 $R/scripts/run_shell_c3.py 362 > $O/config3_shell_roof_n362.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace3 5 k_bsell_spmv 12 > $O/config3_shell_kernel_stats.csv
rm -rf $O/trace3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -- python3 $R/scripts/run_nonlinear_c5.py > $O/config5_nonlinear_n2236.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace5 5 k_spmv_sell 8 > $O/config5_kernel_stats.csv
rm -rf $O/trace5
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
# shell kernels: counters of a short solve on the 362 x 362 roof
for P in "FETCH_SIZE" "WRITE_SIZE L2CacheHit" "GRBM_GUI_ACTIVE VALUBusy MemUnitBusy MemUnitStalled"; do
  D=$O/pmc_shell_$(echo $P | cut -d" " -f1)
  (cd /tmp && FEMO_SHELL_PMC_ITS=4 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_shell_kernels.py 362 > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/shell_pmc_kernels_n362.csv $O/pmc_shell_FETCH_SIZE $O/pmc_shell_WRITE_SIZE $O/pmc_shell_GRBM_GUI_ACTIVE > /dev/null
rm -rf $O/pmc_shell_*
ls -la $O
