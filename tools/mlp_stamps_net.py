"""Per-phase timeline of one wave of the fused MLP kernel INSIDE the network (the last block's launch of an interpolator
forward at B = 25, tile-major input, lazy residual), dropout stream on and off.  Needs a -DSDY_STAMPS build (csrc/Makefile)
selected with SDY_AMD_LIB; run on the GPU box."""
import ctypes as C
import os
import sys

os.environ["SDY_MLP_STAMPS"] = "1"
os.environ.setdefault("SDY_NO_DROP_SKIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sdy_amd._lib import lib as L

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
B = bench.MEMBERS
x, f = bench.synthetic_state(0, B, dev)
ip = exp.model.interpolator
inp = torch.cat([x, x], dim=1)
t = torch.full((B,), 3.0, device=dev)
names = ["x regs -> LDS + barrier", "fc1(0)", "chain(0)", "barrier", "fc1(1)", "fc2(0)||chain(1)", "B+fc1(2)",
         "fc2(1)||chain(2)", "B+fc1(3)", "fc2(2)||chain(3)", "barrier", "x prefetch + fc2(3)", "resid req + epilogue VALU->LDS",
         "barrier", "store phase", "end barrier -> next tile"]
L.sdy_mlp_h3_debug_stamps.argtypes = [C.c_void_p]
L.sdy_mlp_h3_debug_stamps.restype = C.c_int
for drop in (False, True):
    for _ in range(3):
        with ip.inference_dropout_scope(condition=drop):
            ip.predict_packed(inp, time=t, static_condition=f)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    assert L.sdy_mlp_h3_debug_stamps(buf) == 0
    v = list(buf)
    print(f"dropout={drop}")
    rows = []
    for k in range(3):
        s_ = v[k * 16:(k + 1) * 16]
        nxt = v[(k + 1) * 16]
        rows.append([s_[i + 1] - s_[i] for i in range(15)] + [nxt - s_[15]])
    print("  %-34s %8s %8s %8s" % ("phase", "tile0", "tile1", "tile2"))
    for i in range(16):
        print("  %-34s %8d %8d %8d" % (names[i], rows[0][i], rows[1][i], rows[2][i]))
    print("  %-34s %8d %8d %8d" % ("total", sum(rows[0]), sum(rows[1]), sum(rows[2])))
