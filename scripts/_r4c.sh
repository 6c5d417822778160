set -u
R=$(pwd); O=$R/gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bpx.py tests/test_gpu_emulated_ranks.py tests/test_gpu_dist.py tests/test_gpu_round2.py tests/test_gpu_operators.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for V in merged classic; do
  if [ $V = classic ]; then export FEMO_PCG_CLASSIC=1; else unset FEMO_PCG_CLASSIC; fi
  timeout 300 python bench.py --mesh-n 100 --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-pcie > $O/bench_c2_$V.json 2> $O/bench_c2_$V.err
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-pcie > $O/bench_c4_$V.json 2> $O/bench_c4_$V.err
done
