#!/bin/bash
# Round-6 profiles on the GPU box: the driver's bench line (configs, scaling models, checks), kernel-trace summaries of the
# headline / config 2 / the N-rank model leg / the shell / config 5, hardware counters of the hot kernels (separate --pmc
# passes, only ever combined with --kernel-trace), the 8-rank emulated record at the benchmark's size.
# usage: FEMO_COLLECT_COMMIT=<git rev> scripts/collect_profiles_r06.sh OUTDIR   (run from the repo root)
set -u
R=$(pwd); O=$R/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_cmd.json 2> $O/bench_n1.err
timeout 300 python bench.py --mesh-n 100 --steps 20 --warmup 3 --no-cpu-baseline --no-configs > $O/bench_c2_n100.json 2> /dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --jitter 0.2 --no-cpu-baseline --no-configs --no-pcie > $O/bench_jitter.json 2> /dev/null
FEMO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/bench_forced_dist_1rank.json 2> /dev/null
timeout 900 python scripts/run_emulated_ranks_bench.py 215 8 > $O/emulated_8ranks_n215.json 2> /dev/null
# kernel traces
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-configs > $O/bench_under_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 5 k_spmv_sell 7 > $O/bench_kernel_stats.csv
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_rocprofv3.csv 2>/dev/null
rm -rf $O/trace
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace2 -- python3 $R/bench.py --mesh-n 100 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-configs --no-check > /dev/null 2> /dev/null)
python3 scripts/trace_summary.py $O/trace2 3 k_spmv_sell 7 > $O/c2_kernel_stats.csv
python3 scripts/trace_timeline.py $O/trace2 k_load_walk 8 collapse > $O/c2_cycle_timeline.txt 2>&1
rm -rf $O/trace2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_sm -- python3 $R/scripts/run_scaling_model.py 215 6 --model-only > $O/scaling_model_under_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace_sm 3 k_spmv_sell 9 > $O/scaling_model_nrank_path_kernel_stats.csv
python3 scripts/trace_timeline.py $O/trace_sm k_load_walk 3 > $O/scaling_model_nrank_path_timeline.txt 2>&1
rm -rf $O/trace_sm
# the same leg with the round-5 ghost refresh (ncclSend/Recv-shaped: comm stream + events), same box: the A/B of round 6
(cd /tmp && FEMO_HALO_RCCL=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_sm2 -- python3 $R/scripts/run_scaling_model.py 215 6 --model-only > $O/scaling_model_rccl_path_under_rocprof.json 2> /dev/null)
python3 scripts/trace_timeline.py $O/trace_sm2 k_load_walk 3 > $O/scaling_model_rccl_path_timeline.txt 2>&1
rm -rf $O/trace_sm2
timeout 300 python scripts/run_scaling_model.py 215 6 --model-only > $O/scaling_model_direct.json 2> /dev/null
FEMO_HALO_RCCL=1 timeout 300 python scripts/run_scaling_model.py 215 6 --model-only > $O/scaling_model_rccl.json 2> /dev/null
timeout 600 python bench.py --permute --reorder --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-pcie > $O/bench_permuted_reordered.json 2> /dev/null
./scripts/probe_cellcentric/skeleton 215 > $O/cellcentric_skeleton_n215.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -- python3 $R/scripts/run_shell_c3.py 362 > $O/config3_shell_roof_n362.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace3 5 k_bsell_spmv 12 > $O/config3_shell_kernel_stats.csv
rm -rf $O/trace3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -- python3 $R/scripts/run_nonlinear_c5.py > $O/config5_nonlinear_n2236.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace5 5 k_spmv_sell 8 > $O/config5_kernel_stats.csv
rm -rf $O/trace5
if [ "${SKIP_PMC:-0}" = "1" ]; then ls -la $O; exit 0; fi
for P in "FETCH_SIZE" "WRITE_SIZE L2CacheHit" "GRBM_GUI_ACTIVE VALUBusy MemUnitBusy MemUnitStalled"; do
  D=$O/pmc_215_$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_pc_kernels.py 215 > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/pmc_kernels_n215.csv $O/pmc_215_FETCH_SIZE $O/pmc_215_WRITE_SIZE $O/pmc_215_GRBM_GUI_ACTIVE > /dev/null
rm -rf $O/pmc_215_*
# the assembly kernel's instruction mix (round 5: what bounds k_poisson_system_pipe)
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64"; do
  D=$O/pmc_sq_$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_assembly.py 215 > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/pmc_assembly_sq_n215.csv $O/pmc_sq_SQ_WAVE_CYCLES $O/pmc_sq_SQ_INSTS_VALU > /dev/null
rm -rf $O/pmc_sq_*
for P in "FETCH_SIZE" "WRITE_SIZE L2CacheHit" "GRBM_GUI_ACTIVE VALUBusy MemUnitBusy MemUnitStalled"; do
  D=$O/pmc_shell_$(echo $P | cut -d" " -f1)
  (cd /tmp && FEMO_SHELL_PMC_ITS=4 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_shell_kernels.py 362 > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/shell_pmc_kernels_n362.csv $O/pmc_shell_FETCH_SIZE $O/pmc_shell_WRITE_SIZE $O/pmc_shell_GRBM_GUI_ACTIVE > /dev/null
rm -rf $O/pmc_shell_*
# the texture-address / L1 / L2 / LDS side of the shell's lattice transfers (k_pc_restrict_h, k_pc_prolong_fused): the pass
# DESIGN.md section 9 of round 5 called for before touching them again
i=0
for P in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1)); D=$O/pmc_shellpc_$i
  (cd /tmp && FEMO_SHELL_PMC_ITS=4 timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_shell_kernels.py 362 > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/shell_pc_ta_l2_lds_n362.csv $O/pmc_shellpc_1 $O/pmc_shellpc_2 $O/pmc_shellpc_3 $O/pmc_shellpc_4 $O/pmc_shellpc_5 > /dev/null
rm -rf $O/pmc_shellpc_*
for P in "FETCH_SIZE" "WRITE_SIZE L2CacheHit"; do
  D=$O/pmc_100_$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_pc_kernels.py 100 > $D.log 2>&1)
  D=$O/pmc_sq_$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/pmc_pc_kernels.py 2236 square > $D.log 2>&1)
done
python3 scripts/pmc_table.py $O/pmc_kernels_n100.csv $O/pmc_100_FETCH_SIZE $O/pmc_100_WRITE_SIZE > /dev/null
python3 scripts/pmc_table.py $O/pmc_kernels_sq2236.csv $O/pmc_sq_FETCH_SIZE $O/pmc_sq_WRITE_SIZE > /dev/null
rm -rf $O/pmc_100_* $O/pmc_sq_*
python3 scripts/pmc_traffic.py $O/pmc_traffic.json $O/pmc_kernels_n215.csv --also "spmv_n100=$O/pmc_kernels_n100.csv:k_spmv_sell<1, true>" "spmv_sq2236=$O/pmc_kernels_sq2236.csv:k_spmv_sell<1, true>" "bsell_spmv_n362=$O/shell_pmc_kernels_n362.csv:k_bsell_spmv" > /dev/null
ls -la $O
