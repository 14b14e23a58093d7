"""Times the fused MLP kernel against the two-launch path (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sdy_amd as sdy

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def main():
    E, Hd, H, W = 256, 512, 180, 360
    g = torch.Generator().manual_seed(0)
    w1 = torch.randn(Hd, E, generator=g) / 16; b1 = torch.randn(Hd, generator=g) * .1
    w2 = torch.randn(E, Hd, generator=g) / 22; b2 = torch.randn(E, generator=g) * .1
    for B in (int(a) for a in (sys.argv[1:] or ["8", "25"])):
        x = torch.randn(B, E, H, W, device="cuda"); res = torch.randn(B, E, H, W, device="cuda")
        pa = torch.ones(B, E, device="cuda"); pd = torch.zeros(B, E, device="cuda")
        prep = sdy.ops.pack_mlp_h3(w1, w2, "cuda")
        p1, p2 = sdy.ops.pack_h3(w1, "cuda"), sdy.ops.pack_h3(w2, "cuda")
        out = torch.empty_like(x); hid = torch.empty(B, Hd, H, W, device="cuda")
        b1c, b2c = b1.cuda(), b2.cuda()
        for drop in (0.0, 0.1):
            f = lambda: sdy.ops.mlp_fused(x, w1, b1c, w2, b2c, pre_affine=(pa, pd), add=res, drop_p=drop, out=out, prepared=prep)
            def u():
                sdy.ops.conv1x1(x, w1, b1c, pre_affine=(pa, pd), gelu=True, drop_p=drop, h3_prepared=p1, out=hid, wt_prepared=hid, kernel_tag=1)
                sdy.ops.conv1x1(hid, w2, b2c, add=res, add_mode=2, drop_p=drop, stream_id=1, h3_prepared=p2, out=out, wt_prepared=hid, kernel_tag=2)
            tf, tu = timeit(f), timeit(u)
            alg = 3 * B * E * H * W * 4
            print(f"B={B} drop={drop}: fused {tf:.3f} ms ({alg/tf/1e6:.0f} GB/s alg, {2*2*E*Hd*H*W*B/tf/1e9:.1f} TF/s) | two launches {tu:.3f} ms  -> x{tu/tf:.2f}")

main()
