set -u
R=$(pwd); O=$R/gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_scaling_model.py 215 6 > $O/sm_rocprof.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 3 k_spmv_sell 8 > $O/sm_kernel_stats.csv
rm -rf $O/trace
