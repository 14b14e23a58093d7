set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
(time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r4k/gpu_tests.txt 2>&1) 2>&1 | tail -3
tail -20 gpurun_out/r4k/gpu_tests.txt
python tools/small_batch_profile.py 3 4 7 13 25 > gpurun_out/r4k/single_gpu_small_batch.json 2>/dev/null; tail -c 600 gpurun_out/r4k/single_gpu_small_batch.json
