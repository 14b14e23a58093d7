# Same-device A/B of an environment switch through bench.py, interleaved rounds:
#   bash tools/gpu_ab_env.sh <out file> <stage regex> <ENV_NAME>          (run on the GPU box; "off" = the variable set to 1)
cd $GRAFT_REPO_ROOT
O=$1; PAT=$2; V=$3
mkdir -p $(dirname $O); : > $O
for round in 1 2 3; do
  for mode in default $V; do
    if [ $mode = default ]; then timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    else env $V=1 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ab.json 2>/dev/null; fi
    echo "== round $round $mode" >> $O; python tools/show_bench.py /tmp/ab.json | grep -E "^\{|$PAT" | sed "s/'unit'.*//" >> $O
  done
done
cat $O
