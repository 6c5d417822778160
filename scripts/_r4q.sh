set -u
R=$(pwd); O=$R/gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 1500 python bench.py --steps 10 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err
