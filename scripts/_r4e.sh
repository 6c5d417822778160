set -u
R=$(pwd); O=$R/gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python bench.py --mesh-n 100 --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-pcie > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-pcie > $O/bench_c4.json 2> $O/bench_c4.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --mesh-n 100 --steps 10 --warmup 2 --no-cpu-baseline --no-pcie --no-configs --no-check > $O/bench_c2_rocprof.json 2> $O/err.log)
python3 scripts/trace_timeline.py $O/trace k_load_walk 8 > $O/c2_timeline_full.txt 2>&1
rm -rf $O/trace
