set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_sfno.py -q -m gpu -k "full_size or wide or c2_full or tiny" 2>&1 | tail -4
for round in 1 2 3; do
 for v in nchw tiled; do
  echo "== round $round z=$v" >> gpurun_out/r4d/e2e_ab_z_tiled.txt
  if [ "$v" = nchw ]; then export SDY_NO_Z_TILED=1; else unset SDY_NO_Z_TILED; fi
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4d/b.json 2>/dev/null
  python tools/show_bench.py gpurun_out/r4d/b.json | grep -E "^\{|mlp fused|inner-skip" | sed "s/'unit'.*//" >> gpurun_out/r4d/e2e_ab_z_tiled.txt
 done
done
cat gpurun_out/r4d/e2e_ab_z_tiled.txt
