set -u
R=$(pwd); O=$R/gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bpx.py tests/test_gpu_emulated_ranks.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q -k "not shell" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
