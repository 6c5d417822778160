set -u
R=$(pwd); O=$R/gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python bench.py --steps 10 --warmup 2 > $O/bench_full.json 2> $O/bench_full.err; echo "rc=$?"; tail -3 $O/bench_full.err
