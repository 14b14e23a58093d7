"""fc1 / fc2 / inner-skip conv launches at the bench shape (B=25) for the PMC traffic pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd
dev = torch.device("cuda"); B, H, W = 25, 180, 360
def run(Cin, Cout, tag, **kw):
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, device=dev) / 16
    bias = torch.randn(Cout, device=dev)
    out = torch.empty(B, Cout, H, W, device=dev)
    h3 = sdy_amd.ops.pack_h3(w, dev); wt = w.t().contiguous()
    extra = {}
    if kw.get("affine"): extra["pre_affine"] = (torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1)
    if kw.get("add"): extra["add"] = torch.randn(B, Cout, H, W, device=dev); extra["add_mode"] = kw["add"]
    for _ in range(2):
        sdy_amd.ops.conv1x1(x, w, bias, out=out, wt_prepared=wt, h3_prepared=h3, kernel_tag=tag, gelu=kw.get("gelu", False), **extra)
    torch.cuda.synchronize()
run(256, 512, 1, affine=True, gelu=True)
run(512, 256, 2, add=2)
run(256, 256, 3, add=1, gelu=True)
