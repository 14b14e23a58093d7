"""Micro-benchmark of the conv1x1 GEMM variants at production shapes (run on the GPU box)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdy_amd

def run(name, B, Cin, Cout, reps=5, **kw):
    H, W = 180, 360
    dev = torch.device("cuda")
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    bias = torch.randn(Cout, device=dev) * 0.1
    wt = w.t().contiguous()
    out = torch.empty(B, Cout, H, W, device=dev)
    extra = {}
    if kw.pop("affine", False):
        extra["pre_affine"] = (torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1)
    if kw.pop("addpre", False):
        extra["add"] = torch.randn(B, Cout, H, W, device=dev); extra["add_mode"] = 1
    if kw.pop("addpost", False):
        extra["add"] = torch.randn(B, Cout, H, W, device=dev); extra["add_mode"] = 2
    extra.update(kw)
    if extra.pop('h3', False):
        extra['h3_prepared'] = sdy_amd.ops.pack_h3(w, dev)
    f = lambda: sdy_amd.ops.conv1x1(x, w, bias, out=out, wt_prepared=wt, **extra)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * Cin * Cout * H * W * B
    print(f"{name:34s} B={B:2d} {Cin:3d}->{Cout:3d}  {ms:8.3f} ms  {fl/ms/1e9:7.1f} TF/s", flush=True)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
run("plain 256->256", B, 256, 256)
run("plain 256->512", B, 256, 512)
run("plain 512->256", B, 512, 256)
run("skip: bias+addpre+gelu", B, 256, 256, addpre=True, gelu=True, kernel_tag=3)
run("fc1: affine+bias+gelu", B, 256, 512, affine=True, gelu=True, kernel_tag=1)
run("fc1: +dropout(philox)", B, 256, 512, affine=True, gelu=True, kernel_tag=1, drop_p=0.1, seed=1, call=1)
run("fc2: bias+addpost", B, 512, 256, addpost=True, kernel_tag=2)
run("fc2: +dropout(philox)", B, 512, 256, addpost=True, kernel_tag=2, drop_p=0.1, seed=1, call=1)
run("enc 65->256 gelu", B, 65, 256, gelu=True)
for nm, ci, co, kw in [("h3 plain 256->256", 256, 256, {}), ("h3 plain 256->512", 256, 512, {}), ("h3 plain 512->256", 512, 256, {}),
                       ("h3 skip", 256, 256, dict(addpre=True, gelu=True, kernel_tag=3)),
                       ("h3 fc1 +dropout", 256, 512, dict(affine=True, gelu=True, kernel_tag=1, drop_p=0.1, seed=1, call=1)),
                       ("h3 fc2 +dropout", 512, 256, dict(addpost=True, kernel_tag=2, drop_p=0.1, seed=1, call=1)),
                       ("h3 enc 65->256", 65, 256, dict(gelu=True)), ("h3 dec 256->63", 256, 63, {})]:
    run(nm, B, ci, co, h3=True, **kw)
run("dec 256->63", B, 256, 63)
