# Same-device A/B of two library builds through bench.py, interleaved rounds:
#   bash tools/gpu_ab_e2e.sh <out file> <stage regex> name=lib [name=lib ...]        (run on the GPU box)
cd $GRAFT_REPO_ROOT
O=$1; PAT=$2; shift 2
mkdir -p $(dirname $O); : > $O
for round in 1 2 3; do
  for kv in "$@"; do
    n=${kv%%=*}; l=$GRAFT_REPO_ROOT/${kv#*=}
    SDY_AMD_LIB=$l timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    echo "== round $round $n" >> $O; python tools/show_bench.py /tmp/ab.json | grep -E "^\{|$PAT" | sed "s/'unit'.*//" >> $O
  done
done
cat $O
