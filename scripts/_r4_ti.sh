#!/bin/bash
O=gpurun_out/r4ti; mkdir -p $O; R=$(pwd)
python -m pytest tests/test_gpu_shell.py tests/test_gpu_shell_hermite.py -x -q 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/scripts/run_shell_c3.py 362 > $R/$O/c3.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 5 k_bsell_spmv 40 | grep "trinv\|chol\|total kernel"
rm -rf $O/trace
python3 -c "
import json; d=json.load(open('$O/c3.json')); print(d['forward_cg_iterations'], d['adjoint_cg_iterations'], d['forward_solve_device_ms'], d['adjoint_solve_device_ms'])"
