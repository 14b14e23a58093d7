set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_gpu_ops.py tests/test_gpu_sfno.py tests/test_gpu_variants.py -q -m gpu --durations=8 > gpurun_out/r4i/gpu_tests_rest.txt 2>&1
tail -25 gpurun_out/r4i/gpu_tests_rest.txt
