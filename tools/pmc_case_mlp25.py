"""Fused MLP launches at the bench shape (B=25) for the PMC traffic pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd as sdy
B, E, Hd, H, W = 25, 256, 512, 180, 360
g = torch.Generator().manual_seed(0)
w1 = torch.randn(Hd, E, generator=g) / 16; b1 = (torch.randn(Hd, generator=g) * .1).cuda()
w2 = torch.randn(E, Hd, generator=g) / 22; b2 = (torch.randn(E, generator=g) * .1).cuda()
x = torch.randn(B, E, H, W, device="cuda"); res = torch.randn(B, E, H, W, device="cuda")
pa = torch.ones(B, E, device="cuda"); pd = torch.zeros(B, E, device="cuda")
prep = sdy.ops.pack_mlp_h3(w1, w2, "cuda"); out = torch.empty_like(x)
for _ in range(3):
    sdy.ops.mlp_fused(x, w1, b1, w2, b2, pre_affine=(pa, pd), add=res, out=out, prepared=prep)
torch.cuda.synchronize()
