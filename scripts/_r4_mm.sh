#!/bin/bash
O=gpurun_out/r4mm; mkdir -p $O
python -m pytest tests/test_gpu_shell_hermite.py tests/test_gpu_shell.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
R=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/scripts/run_shell_c3.py 362 > $R/$O/c3.json 2> /dev/null)
python3 scripts/trace_summary.py $O/trace 5 k_bsell_spmv 12 > $O/c3_kernel_stats.csv
rm -rf $O/trace
head -14 $O/c3_kernel_stats.csv
python3 -c "
import json; d=json.load(open('$O/c3.json')); print(d['forward_cg_iterations'], d['adjoint_cg_iterations'], d['forward_solve_device_ms'], d['adjoint_solve_device_ms'])"
