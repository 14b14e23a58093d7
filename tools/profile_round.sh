#!/bin/bash
# Per-round profile set (GPU box, via gpurun): bash tools/profile_round.sh <tag>     (r2a, r2b, r3a, ...)
#   1. bench line (default run) + rocprofv3 --kernel-trace --stats of the same command
#   2. PMC passes over one interpolator + one forecaster forward at B = 25 (tools/pmc_forward25.py), each counter group in its
#      own pass with --kernel-trace only: FETCH_SIZE, WRITE_SIZE (HBM traffic), matrix-pipe / issue counters
TAG=${1:-r3b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p $OUT
cd $R
python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; head -c 600 $OUT/bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/prof.log 2>&1
echo "rocprof stats rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" -delete
head -16 $OUT/kernel_stats.csv | cut -c1-150
agg() {
python3 - "$1" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if any(t in r["Kernel_Name"] for t in ("mlp", "gemm", "fft", "leg_", "dh_h3", "conv_h3", "pair_h3")):
        agg[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:66s} {c:26s} avg {sum(v)/len(v):16.1f} over {len(v)} dispatches")
PY
}
# (the counter passes keep every launch at 25 rows: bench.py scales the per-launch traffic to the rows its launches covered)
export SDY_NO_DROP_SKIP=1
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"; do
  T=$(echo $CTRS | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$T -- python3 $R/tools/pmc_forward25.py > $OUT/pmc_$T.log 2>&1
  echo "rc=$? $CTRS"
  F=$(find $OUT/pmc_$T -name "*counter_collection.csv" | head -1)
  agg "$F" > $OUT/pmc_$T.txt
  find $OUT/pmc_$T -name "*.csv" -size +5M -delete
  head -60 $OUT/pmc_$T.txt
done
unset SDY_NO_DROP_SKIP
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>/dev/null; head -20 $OUT/pmc_summary.txt
