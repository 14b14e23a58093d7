set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sfno.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -25
python tools/precision_probe.py 2>/dev/null | tail -1
O=gpurun_out/r4n/e2e_ab_pair_dynamic_scale.txt
for round in 1 2 3; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4n/b.json 2>/dev/null
  echo "== round $round dynamic" >> $O; python tools/show_bench.py gpurun_out/r4n/b.json | grep -E "^\{|fused pair" | sed "s/'unit'.*//" >> $O
done
cat $O
