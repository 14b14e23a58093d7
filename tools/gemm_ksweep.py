"""Fixed-overhead vs per-k-tile cost of the conv GEMM kernels: time vs Cin at fixed Cout (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdy_amd

def t(B, Cin, Cout, h3, reps=5):
    H, W = 180, 360
    dev = torch.device("cuda")
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    wt = w.t().contiguous()
    out = torch.empty(B, Cout, H, W, device=dev)
    kw = dict(out=out, wt_prepared=wt)
    if h3:
        kw["h3_prepared"] = sdy_amd.ops.pack_h3(w, dev)
    f = lambda: sdy_amd.ops.conv1x1(x, w, None, **kw)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B = 8
for h3 in (False, True):
    for Cout in (128, 256):
        row = []
        for Cin in (64, 128, 256, 512, 1024):
            row.append(t(B, Cin, Cout, h3))
        print(("h3 " if h3 else "f32"), f"Cout={Cout}", " ".join(f"K={k}:{v:.3f}ms" for k, v in zip((64, 128, 256, 512, 1024), row)), flush=True)
