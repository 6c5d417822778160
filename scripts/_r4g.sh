set -u
R=$(pwd); O=$R/gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --scaling-model --no-pcie > $O/bench_c4_sm.json 2> $O/bench_c4_sm.err
FEMO_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
