import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdy_amd as sdy
E, Hd, B, H, W = 256, 512, 1, 4, 16
F = torch.nn.functional
g = torch.Generator().manual_seed(0)
x = torch.randn(B, E, H, W, generator=g)
w1 = torch.zeros(Hd, E); w1[torch.arange(Hd), torch.arange(Hd) % E] = 1.0
b1 = torch.zeros(Hd); b2 = torch.zeros(E)
ref = F.gelu(x.double())
for k0 in (0, 64, 128, 192, 256, 320, 384, 448):
    w2 = torch.zeros(E, Hd); w2[torch.arange(E), (torch.arange(E) + k0) % Hd] = 1.0
    r = ref[:, (torch.arange(E) + k0) % E]
    got = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2).cpu().double()
    d = (got - r).abs().reshape(8, 32, H * W)
    nan = torch.isnan(got).reshape(8, 32, H * W)
    print(f"k0={k0}: nan per 32-row block {nan.any(2).sum(1).tolist()}  maxerr per block {[f'{v:.1e}' for v in torch.nan_to_num(d, nan=0).amax((1,2)).tolist()]}")
