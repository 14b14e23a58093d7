set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_ops.py tests/test_gpu_dyffusion.py tests/test_gpu_sfno.py -q -m gpu -k "full_size or wide or injected or ensemble_statistics or predict_step or mlp_fused or c2_full" > gpurun_out/r4b/new_tests.txt 2>&1
tail -15 gpurun_out/r4b/new_tests.txt
python tools/precision_probe.py > gpurun_out/r4b/precision_tiled.txt 2>&1; tail -1 gpurun_out/r4b/precision_tiled.txt
for round in 1 2 3; do
 for kv in rowmajor=1 tiled=; do
  n=${kv%%=*}; v=${kv#*=}
  echo "== round $round $n" >> gpurun_out/r4b/e2e_ab_tiled.txt
  if [ -n "$v" ]; then export SDY_NO_XF_TILED=1; else unset SDY_NO_XF_TILED; fi
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras >> gpurun_out/r4b/e2e_ab_tiled.txt 2>&1
 done
done
unset SDY_NO_XF_TILED
grep -E "==|value" gpurun_out/r4b/e2e_ab_tiled.txt | sed 's/"unit".*//'
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4b/bench_tiled.json 2>/dev/null; SDY_NO_XF_TILED=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4b/bench_rowmajor.json 2>/dev/null
python tools/show_bench.py gpurun_out/r4b/bench_tiled.json; python tools/show_bench.py gpurun_out/r4b/bench_rowmajor.json
python tools/small_batch_profile.py 3 4 25 > gpurun_out/r4b/small_batch.json 2> gpurun_out/r4b/small_batch.err; tail -c 3000 gpurun_out/r4b/small_batch.json
