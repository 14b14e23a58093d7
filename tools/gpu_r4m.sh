set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout 900 python tools/c4_rollout.py --steps 102 > gpurun_out/r4m/c4_rollout_102_steps.json 2> gpurun_out/r4m/c4.err; tail -c 700 gpurun_out/r4m/c4_rollout_102_steps.json
timeout 900 python tools/c4_rollout.py --steps 60 --ics 4 > gpurun_out/r4m/c5_shape_4ics_60_steps.json 2> gpurun_out/r4m/c5.err; tail -c 700 gpurun_out/r4m/c5_shape_4ics_60_steps.json
timeout 900 python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4m/bench_100_passes.json 2>/dev/null; head -c 300 gpurun_out/r4m/bench_100_passes.json
timeout 600 python bench.py > gpurun_out/r4m/bench_default.json 2>/dev/null; head -c 300 gpurun_out/r4m/bench_default.json
