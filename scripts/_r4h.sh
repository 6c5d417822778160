set -u
R=$(pwd); O=$R/gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bpx.py tests/test_gpu_emulated_ranks.py tests/test_gpu_round2.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python scripts/run_scaling_model.py 215 10 > $O/scaling_model.json 2> $O/sm.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_scaling_model.py 215 6 > $O/sm_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 3 k_spmv_sell 8 > $O/sm_kernel_stats.csv
rm -rf $O/trace
