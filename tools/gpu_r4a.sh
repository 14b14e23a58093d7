set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
V=build/variants
( python tools/precision_probe.py; SDY_AMD_LIB=$PWD/$V/libsdy_amd_sx1.so python tools/precision_probe.py; SDY_AMD_LIB=$PWD/$V/libsdy_amd_r3.so python tools/precision_probe.py ) > gpurun_out/r4a/precision_probe.txt 2>&1
tail -3 gpurun_out/r4a/precision_probe.txt
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_ops.py tests/test_gpu_dyffusion.py -x -q -m gpu -k "full_size or wide or injected or ensemble_statistics or predict_step or mlp_fused" > gpurun_out/r4a/new_tests.txt 2>&1
tail -15 gpurun_out/r4a/new_tests.txt
for round in 1 2; do
 for kv in r3=$PWD/$V/libsdy_amd_r3.so new= sx1=$PWD/$V/libsdy_amd_sx1.so; do
  n=${kv%%=*}; l=${kv#*=}
  echo "== round $round $n" >> gpurun_out/r4a/e2e_ab.txt
  SDY_AMD_LIB=$l timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras >> gpurun_out/r4a/e2e_ab.txt 2>&1
 done
done
grep -E "==|value" gpurun_out/r4a/e2e_ab.txt | sed 's/"unit".*//'
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r4a/bench_first.json 2> gpurun_out/r4a/bench_first.err
tail -c 1500 gpurun_out/r4a/bench_first.json
