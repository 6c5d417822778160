set -u
R=$(pwd); O=$R/gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
for V in hermite trilinear; do
  if [ $V = trilinear ]; then export FEMO_SHELL_TRILINEAR=1; else unset FEMO_SHELL_TRILINEAR; fi
  timeout 600 python scripts/run_shell_c3.py 362 > $O/c3_$V.json 2> $O/c3_$V.err
  tail -2 $O/c3_$V.err
done
unset FEMO_SHELL_TRILINEAR
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_shell_c3.py 362 > /dev/null 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 3 k_bsell_spmv 14 > $O/c3_kernel_stats.csv
rm -rf $O/trace
