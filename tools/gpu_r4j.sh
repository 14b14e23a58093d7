set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_sfno.py -q -m gpu -k "full_size or wide or c2_full or tiny" 2>&1 | tail -4
O=gpurun_out/r4j/e2e_ab_cs_tiled.txt
for round in 1 2 3; do
 for v in rowmajor tiled; do
  echo "== round $round cs=$v" >> $O
  if [ "$v" = rowmajor ]; then export SDY_NO_CS_TILED=1; else unset SDY_NO_CS_TILED; fi
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r4j/b.json 2>/dev/null
  python tools/show_bench.py gpurun_out/r4j/b.json | grep -E "^\{|legendre|dhconv" | sed "s/'unit'.*//" >> $O
 done
done
cat $O
unset SDY_NO_CS_TILED
(time timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3_full_size") 2>&1 | tail -6
