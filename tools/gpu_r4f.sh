set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_dyffusion.py tests/test_gpu_golden.py -q -m gpu -k "shared_encoder or sampler_vs_reference or fused_interpolator or sample_matches or full_size" 2>&1 | tail -5
O=gpurun_out/r4f/e2e_ab_encoder_reuse.txt
for round in 1 2 3; do
 for v in off on; do
  echo "== round $round encoder reuse $v" >> $O
  if [ "$v" = off ]; then export SDY_NO_ENCODER_REUSE=1; else unset SDY_NO_ENCODER_REUSE; fi
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | sed 's/"unit".*//;s/.*"value"/value/' >> $O
 done
done
cat $O
