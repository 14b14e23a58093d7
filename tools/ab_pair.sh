#!/bin/bash
# A/B of pair_h3 builds on one device, interleaved rounds: bash tools/ab_pair.sh <out> name=lib ...
OUT=$1; shift
mkdir -p $(dirname $OUT)
: > $OUT
for round in 1 2 3; do
  for kv in "$@"; do
    n=${kv%%=*}; l=${kv#*=}
    echo "== round $round $n" >> $OUT
    SDY_AMD_LIB=$l timeout 300 python tools/pair_bench.py >> $OUT 2>&1
  done
done
grep -E "==|fused" $OUT
