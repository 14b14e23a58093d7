# End-of-round measurement set (GPU box): default bench line, small-batch stage profile, the C4 rollout by wall time, a two-rank
# run of bench.py on one GPU (test mode of the N > 1 path).     bash tools/gpu_r4_final.sh <tag>
TAG=${1:-r4d}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 400 $O/bench_default.json; echo
python tools/small_batch_profile.py 1 3 4 7 13 25 > $O/single_gpu_small_batch.json 2>/dev/null; echo "small batch rc=$?"
python tools/c4_rollout.py --steps 102 --members 25 > $O/c4_rollout_102_steps.json 2>/dev/null; echo "c4 rc=$?"; tail -c 500 $O/c4_rollout_102_steps.json; echo
python bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_two_ranks_one_gpu_test_mode.json 2>$O/two.err; echo "two ranks rc=$?"; head -c 300 $O/bench_two_ranks_one_gpu_test_mode.json; echo
