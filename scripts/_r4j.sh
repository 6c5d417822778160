set -u
R=$(pwd); O=$R/gpurun_out/r4j; mkdir -p $O
timeout 300 python scripts/run_scaling_model.py 215 4 > $O/sm.json 2> $O/sm.err; tail -5 $O/sm.err
