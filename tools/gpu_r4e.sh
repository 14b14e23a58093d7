set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
O=gpurun_out/r4e/two_process_overlap.txt
for round in 1 2; do
  echo "== round $round single process B=25" >> $O
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | sed 's/"unit".*//' >> $O
  echo "== round $round two processes (13 + 12 members) on one GPU, no CU mask" >> $O
  timeout 900 python bench.py --gpus 2 --share-gpu --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | sed 's/"unit".*//' >> $O
  echo "== round $round two processes, disjoint CU halves (HSA_CU_MASK)" >> $O
  SDY_BENCH_CU_SPLIT=1 timeout 900 python bench.py --gpus 2 --share-gpu --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | sed 's/"unit".*//' >> $O
done
cat $O
