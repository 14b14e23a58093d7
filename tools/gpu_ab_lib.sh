cd $GRAFT_REPO_ROOT
O=$1; PAT=$2; LIB=$3
mkdir -p $(dirname $O); : > $O
for round in 1 2 3; do
  for mode in default prev; do
    if [ $mode = default ]; then timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ab.json 2>/dev/null
    else env SDY_AMD_LIB=$LIB timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ab.json 2>/dev/null; fi
    echo "== round $round $mode" >> $O; python tools/show_bench.py /tmp/ab.json | grep -E "^\{|$PAT" | sed "s/'unit'.*//" >> $O
  done
done
cat $O
