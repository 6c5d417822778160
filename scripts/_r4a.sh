set -u
R=$(pwd); O=$R/gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --mesh-n 100 --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-check > $O/bench_c2.json 2> $O/bench_c2.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --mesh-n 100 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-configs --no-check > $O/bench_c2_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 3 k_spmv_sell 8 > $O/c2_kernel_stats.csv
python3 scripts/trace_gaps.py $O/trace 200 > $O/c2_trace_gaps.txt 2>&1
rm -rf $O/trace
tail -3 $O/pytest.log
