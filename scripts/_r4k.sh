set -u
R=$(pwd); O=$R/gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
for V in base dbg2; do
  unset FEMO_DEBUG_COARSE FEMO_COARSE_NO_STAGE
  [ $V = dbg1 ] && export FEMO_DEBUG_COARSE=1
  [ $V = dbg2 ] && export FEMO_DEBUG_COARSE=2
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$V -- python3 $R/scripts/run_scaling_model.py 215 4 > /dev/null 2> /dev/null)
  python3 scripts/trace_summary.py $O/trace_$V 3 | grep "k_lattice_coarse_m\|k_spmv_sell<1, true>\|k_lattice_restrict" > $O/stats_$V.txt
  rm -rf $O/trace_$V
  echo $V; cat $O/stats_$V.txt
done
