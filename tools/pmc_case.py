import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd
dev = torch.device("cuda"); B, Cin, Cout, H, W = 8, 256, 256, 180, 360
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, device=dev) / 16
out = torch.empty(B, Cout, H, W, device=dev)
h3 = sdy_amd.ops.pack_h3(w, dev); wt = w.t().contiguous()
for _ in range(3):
    sdy_amd.ops.conv1x1(x, w, None, out=out, wt_prepared=wt, h3_prepared=h3)
    sdy_amd.ops.conv1x1(x, w, None, out=out, wt_prepared=wt)
torch.cuda.synchronize()
