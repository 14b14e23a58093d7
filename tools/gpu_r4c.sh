set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
for v in 3 2; do
  SDY_LEG_PERSIST=$v timeout 600 python -m pytest tests/test_gpu_golden.py -q -m gpu -k "full_size" 2>&1 | tail -3
done
for round in 1 2; do
 for v in base 3 2; do
  echo "== round $round persist=$v" >> gpurun_out/r4c/e2e_ab_leg_persist.txt
  if [ "$v" = base ]; then unset SDY_LEG_PERSIST; else export SDY_LEG_PERSIST=$v; fi
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4c/b.json 2>/dev/null
  python tools/show_bench.py gpurun_out/r4c/b.json | grep -E "^\{|legendre" | sed "s/'unit'.*//" >> gpurun_out/r4c/e2e_ab_leg_persist.txt
 done
done
cat gpurun_out/r4c/e2e_ab_leg_persist.txt
