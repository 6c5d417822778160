set -u
R=$(pwd); O=$R/gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --mesh-n 100 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-configs --no-check > $O/bench_c2_rocprof.json 2> $O/err.log)
python3 scripts/trace_timeline.py $O/trace k_load_walk 8 > $O/c2_timeline_full.txt 2>&1
python3 scripts/trace_summary.py $O/trace 3 k_spmv_sell 8 > $O/c2_kernel_stats.csv
rm -rf $O/trace
