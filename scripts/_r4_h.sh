#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests/test_gpu_bpx.py tests/test_gpu_emulated_ranks.py tests/test_gpu_shell_hermite.py tests/test_gpu_dist.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --mesh-n 100 --steps 40 --warmup 10 --no-cpu-baseline --no-configs --no-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'])"; done
timeout 300 python scripts/run_scaling_model.py 215 6 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sm', d['ms_per_cycle_block'], d['us_per_cg_iteration_wall'])"
