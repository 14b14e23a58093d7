# Same-device A/B of environment switches at small batches (the 8-GPU operating point), interleaved rounds:
#   bash tools/gpu_ab_small_batch.sh <out file> "<B list>" ENV_A ENV_B ...      (run on the GPU box; each ENV is set to 1 in turn)
cd $GRAFT_REPO_ROOT
O=$1; BS=$2; shift 2
mkdir -p $(dirname $O); : > $O
for round in 1 2; do
  for mode in default "$@"; do
    if [ $mode = default ]; then python tools/small_batch_profile.py $BS 2>/dev/null | tail -1 > /tmp/sb.json
    else env $mode=1 python tools/small_batch_profile.py $BS 2>/dev/null | tail -1 > /tmp/sb.json; fi
    python - "$round" "$mode" >> $O <<'PY'
import json, sys
d = json.load(open("/tmp/sb.json"))
print("== round", sys.argv[1], sys.argv[2], {B: o["pass_ms"] for B, o in d.items()})
PY
  done
done
cat $O
