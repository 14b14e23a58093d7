set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r4g/gpu_tests.txt 2>&1
tail -30 gpurun_out/r4g/gpu_tests.txt
