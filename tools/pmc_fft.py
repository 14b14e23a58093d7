import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd as sdy
B, C, H, W = 8, 256, 180, 360
x = torch.randn(B, C, H, W, device="cuda")
sht = sdy.RealSHT(H, W, lmax=180, mmax=181, grid="equiangular").cuda() if hasattr(sdy.RealSHT(H, W), "cuda") else sdy.RealSHT(H, W, lmax=180, mmax=181, grid="equiangular")
isht = sdy.InverseRealSHT(H, W, lmax=180, mmax=181, grid="equiangular")
for _ in range(3):
    c = sht(x)
    y = isht(c)
torch.cuda.synchronize()
