set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
SDY_LEG_PREFETCH=256 timeout 600 python -m pytest tests/test_gpu_golden.py -q -m gpu -k "full_size" 2>&1 | tail -2
O=gpurun_out/r4l/e2e_ab_leg_l2_prefetch.txt
for round in 1 2; do
 for v in 0 128 256 512 768; do
  echo "== round $round prefetch distance $v" >> $O
  SDY_LEG_PREFETCH=$v timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4l/b.json 2>/dev/null
  python tools/show_bench.py gpurun_out/r4l/b.json | grep -E "^\{|legendre" | sed "s/'unit'.*//" >> $O
 done
done
cat $O
for kv in base= k2=$PWD/build/variants/libsdy_amd_fft_k2.so; do
  n=${kv%%=*}; l=${kv#*=}
  echo "== fft $n" >> gpurun_out/r4l/e2e_ab_fft_kpw2.txt
  SDY_AMD_LIB=$l timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4l/b.json 2>/dev/null
  python tools/show_bench.py gpurun_out/r4l/b.json | grep -E "^\{|fft" | sed "s/'unit'.*//" >> gpurun_out/r4l/e2e_ab_fft_kpw2.txt
done
cat gpurun_out/r4l/e2e_ab_fft_kpw2.txt
