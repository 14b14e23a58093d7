"""Per-phase timeline of one wave of the inner-skip convolution (conv_h3) INSIDE the network: the last block's launch of an
interpolator forward at B = 25 (affine prologue from norm0, addend y, GELU, norm1 statistics, tile-major output).  Needs a
-DSDY_STAMPS build (csrc/Makefile) selected with SDY_AMD_LIB; run on the GPU box."""
import ctypes as C
import os
import sys

os.environ["SDY_CONV_STAMPS"] = "1"
os.environ.setdefault("SDY_NO_DROP_SKIP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from sdy_amd._lib import lib as L

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else bench.MEMBERS
x, f = bench.synthetic_state(0, B, dev)
ip = exp.model.interpolator
inp = torch.cat([x, x], dim=1)
t = torch.full((B,), 3.0, device=dev)
names = ["x regs->LDS+barrier", "prefetch issue", "MFMA", "barrier", "acc->LDS+barrier", "store loop (GELU)", "stats",
         "end barrier->next"]
L.sdy_conv256_h3_debug_stamps.argtypes = [C.c_void_p]
L.sdy_conv256_h3_debug_stamps.restype = C.c_int
for rep in range(3):
    for _ in range(2):
        with ip.inference_dropout_scope(condition=True):
            ip.predict_packed(inp, time=t, static_condition=f)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 64)()
    assert L.sdy_conv256_h3_debug_stamps(buf) == 0
    v = list(buf)
    print(f"B={B} sample {rep}")
    for tl in range(3):
        s_ = v[tl * 8:(tl + 1) * 8]
        nxt = v[(tl + 1) * 8]
        d = [s_[i + 1] - s_[i] for i in range(7)] + [nxt - s_[7]]
        print(f"  tile {tl}: total {sum(d):6d}: " + ", ".join(f"{n} {x}" for n, x in zip(names, d)))
