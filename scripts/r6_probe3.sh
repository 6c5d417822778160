#!/bin/bash
# round 6: the model rank's cycle with memory copies in the trace (what sits in the gaps of its output phase)
set -u
R=$(pwd); O=$R/gpurun_out/r6_p3; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/scripts/run_scaling_model.py 215 6 --model-only > $O/model.json 2> /dev/null)
python3 scripts/trace_timeline.py $O/trace k_load_walk 3 collapse > $O/timeline_copies.txt 2>&1
rm -rf $O/trace
cat $O/timeline_copies.txt | cut -c1-140
