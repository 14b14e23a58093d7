#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats of the same command.
# Usage: bash tools/gpu_profile.sh <tag> [bench args...]
TAG=${1:-r1}; shift
ARGS=${@:---steps 2 --warmup 1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd $R
python3 bench.py $ARGS > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"; cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"; tail -2 $R/gpurun_out/prof_$TAG.log
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'
# keep only the small summaries (kernel_trace.csv can be large)
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
