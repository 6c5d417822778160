#!/bin/bash
# round 6: timelines of the N-rank path (model communicator), device-initiated vs ncclSend/Recv-shaped ghost refresh
set -u
R=$(pwd); O=$R/gpurun_out/r6_p2; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_d -- python3 $R/scripts/run_scaling_model.py 215 6 --model-only > $O/model_direct.json 2> /dev/null)
python3 scripts/trace_timeline.py $O/trace_d k_load_walk 3 > $O/timeline_direct.txt 2>&1
rm -rf $O/trace_d
export FEMO_HALO_RCCL=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_s -- python3 $R/scripts/run_scaling_model.py 215 6 --model-only > $O/model_staged.json 2> /dev/null)
python3 scripts/trace_timeline.py $O/trace_s k_load_walk 3 > $O/timeline_staged.txt 2>&1
rm -rf $O/trace_s
sed -n 60,110p $O/timeline_direct.txt
