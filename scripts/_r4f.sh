set -u
R=$(pwd); O=$R/gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_emulated_ranks.py tests/test_gpu_hostmem.py tests/test_gpu_bpx.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
