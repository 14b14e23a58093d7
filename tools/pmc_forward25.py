"""One full-size interpolator forward at the bench shape (B = 25, dropout stream on) + one forecaster forward: every hot
kernel runs in its network context (norm coefficients, statistics epilogues, Philox dropout) for the PMC passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
x, f = bench.synthetic_state(0, bench.MEMBERS, dev)
ip = exp.model.interpolator
inp = torch.cat([x, x], dim=1)
t = torch.full((bench.MEMBERS,), 3.0, device=dev)
for _ in range(2):
    with ip.inference_dropout_scope(condition=True):
        ip.predict_packed(inp, time=t, static_condition=f)
    exp.model.model(x, time=torch.zeros(bench.MEMBERS, device=dev), static_condition=f)
torch.cuda.synchronize()
