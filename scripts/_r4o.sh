set -u
R=$(pwd); O=$R/gpurun_out/r4o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_shell.py tests/test_gpu_shell_round3.py tests/test_gpu_shell_ranks.py tests/test_gpu_shell_hermite.py tests/test_gpu_fullsize.py -q -k "shell or hermite" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
