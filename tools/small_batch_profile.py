"""Per-stage kernel time per MEMBER of one horizon-6 sampling pass at several batch sizes (the 8-GPU operating point of the
strong-scaling metric is B = 3-4 per GPU): which stages lose efficiency at small batches.  Run on the GPU box.
    python tools/small_batch_profile.py [B ...]            -> one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
Bs = [int(a) for a in sys.argv[1:]] or [3, 4, 25]
out = {}
for B in Bs:
    x, f = bench.synthetic_state(0, B, dev)
    exp.set_batch_offset(0)
    for _ in range(2):
        x = bench.one_pass(exp, x, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3 if B > 8 else 6
    for _ in range(n):
        x = bench.one_pass(exp, x, f)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    rows, total = bench.profile_pass(exp, x, f, B)
    out[str(B)] = {"pass_ms": round(wall, 2), "ms_per_member": round(wall / B, 3), "kernel_ms": round(total, 2),
                   "stages_ms_per_member": {r["name"]: round(r["ms"] * r["launches_per_step"] / B, 3) for r in rows}}
ref = out.get("25")
if ref:
    for B in Bs:
        if B != 25:
            o = out[str(B)]
            o["vs_b25"] = {k: round(v / ref["stages_ms_per_member"][k], 3) for k, v in o["stages_ms_per_member"].items()
                           if k in ref["stages_ms_per_member"] and ref["stages_ms_per_member"][k] > 0}
            o["per_member_vs_b25"] = round(o["ms_per_member"] / ref["ms_per_member"], 3)
print(json.dumps(out))
