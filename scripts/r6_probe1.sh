#!/bin/bash
# round 6 probe: C4 cycle timeline with copies; the unstructured numbering as a run of its own
set -u
R=$(pwd); O=$R/gpurun_out/r6_p1; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-pcie --no-configs --no-check > $O/bench_traced.json 2> $O/bench_traced.err)
python3 scripts/trace_timeline.py $O/trace k_load_walk 5 collapse > $O/c4_cycle_timeline.txt 2>&1
python3 scripts/trace_timeline.py $O/trace k_load_walk 5 > $O/c4_cycle_timeline_full.txt 2>&1
rm -rf $O/trace
timeout 600 python bench.py --permute --reorder --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-pcie > $O/bench_permuted_reordered.json 2> $O/pr.err
tail -c 400 $O/c4_cycle_timeline.txt
