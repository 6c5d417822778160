set -u
R=$(pwd); O=$R/gpurun_out/r4t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shell_ranks.py tests/test_gpu_shell.py tests/test_gpu_shell_round3.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
