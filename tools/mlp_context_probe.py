"""Which part of the in-network gap between the fused MLP with and without dropout is the dropout?  Stage times of
(a) an interpolator forward with the dropout stream on, (b) the same forward with it off, (c) a forecaster forward, B = 25
(run on the GPU box; SDY_NO_DROP_SKIP=1 keeps every launch at 25 rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import sdy_amd

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else bench.MEMBERS
x, f = bench.synthetic_state(0, B, dev)
ip = exp.model.interpolator
inp = torch.cat([x, x], dim=1)
t = torch.full((B,), 3.0, device=dev)


def interp(drop):
    with ip.inference_dropout_scope(condition=drop):
        ip.predict_packed(inp, time=t, static_condition=f)


def forecast():
    exp.model.model(x, time=torch.zeros(B, device=dev), static_condition=f)


for name, fn in (("interpolator, dropout on", lambda: interp(True)), ("interpolator, dropout off", lambda: interp(False)),
                 ("forecaster", forecast)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with sdy_amd.ops.stage_timer() as tm:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    print("==", name)
    for k, (cnt, ms) in sorted(tm.stages.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:26s} x{cnt:4d} {ms / cnt:8.4f} ms  rows {tm.rows[k] / cnt:5.1f}")
