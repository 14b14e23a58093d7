set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sfno.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -25
O=gpurun_out/r4o/e2e_ab_pair_dynamic_scale.txt
for round in 1 2 3; do
  for kv in base=build_ab/libsdy_base.so dynamic=spherical-dyffusion_amd/libsdy_amd.so; do
    n=${kv%%=*}; l=$GRAFT_REPO_ROOT/${kv#*=}
    SDY_AMD_LIB=$l timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4o/b.json 2>/dev/null
    echo "== round $round $n" >> $O; python tools/show_bench.py gpurun_out/r4o/b.json | grep -E "^\{|fused pair" | sed "s/'unit'.*//" >> $O
  done
done
cat $O
