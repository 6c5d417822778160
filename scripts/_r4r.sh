set -u
R=$(pwd); O=$R/gpurun_out/r4r; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_nonlinear_c5.py > $O/c5.json 2> $O/c5.err)
python3 scripts/trace_summary.py $O/trace 3 k_spmv_sell 10 > $O/c5_kernel_stats.csv
python3 scripts/trace_gaps.py $O/trace 500 > $O/c5_gaps.txt 2>&1
rm -rf $O/trace
