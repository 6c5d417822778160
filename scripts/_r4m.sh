set -u
R=$(pwd); O=$R/gpurun_out/r4m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shell_hermite.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
