"""Per-phase timeline of one wave of the fused MLP kernel (dev aid; run with SDY_MLP_STAMPS=1 on the GPU box)."""
import sys, os, ctypes as C
os.environ["SDY_MLP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd as sdy
B, E, Hd, H, W = 25, 256, 512, 180, 360
g = torch.Generator().manual_seed(0)
w1 = torch.randn(Hd, E, generator=g) / 16; b1 = (torch.randn(Hd, generator=g) * .1).cuda()
w2 = torch.randn(E, Hd, generator=g) / 22; b2 = (torch.randn(E, generator=g) * .1).cuda()
x = torch.randn(B, E, H, W, device="cuda"); res = torch.randn(B, E, H, W, device="cuda")
pa = torch.ones(B, E, device="cuda"); pd = torch.zeros(B, E, device="cuda")
prep = sdy.ops.pack_mlp_h3(w1, w2, "cuda"); out = torch.empty_like(x)
names = ["x regs -> LDS + barrier", "fc1(0)", "chain(0)", "barrier", "fc1(1)", "fc2(0)||chain(1)", "B+fc1(2)",
         "fc2(1)||chain(2)", "B+fc1(3)", "fc2(2)||chain(3)", "barrier", "x prefetch + fc2(3)", "resid req + epilogue VALU->LDS",
         "barrier", "store phase", "end barrier -> next tile"]
for drop in (0.0, 0.1):
    for _ in range(2):
        sdy.ops.mlp_fused(x, w1, b1, w2, b2, pre_affine=(pa, pd), add=res, out=out, prepared=prep, drop_p=drop)
    buf = (C.c_uint64 * 64)()
    lib = sdy._lib.lib if hasattr(sdy, "_lib") else None
    from importlib import import_module
    L = import_module("sdy_amd._lib").lib
    L.sdy_mlp_h3_debug_stamps.argtypes = [C.c_void_p]; L.sdy_mlp_h3_debug_stamps.restype = C.c_int
    assert L.sdy_mlp_h3_debug_stamps(buf) == 0
    v = list(buf)
    print(f"drop={drop}")
    rows = []
    for t in range(3):
        s_ = v[t * 16:(t + 1) * 16]; nxt = v[(t + 1) * 16]
        rows.append([s_[i + 1] - s_[i] for i in range(15)] + [nxt - s_[15]])
    print("  %-34s %8s %8s %8s" % ("phase", "tile0", "tile1", "tile2"))
    for i in range(16):
        print("  %-34s %8d %8d %8d" % (names[i], rows[0][i], rows[1][i], rows[2][i]))
    print("  %-34s %8d %8d %8d" % ("total", sum(rows[0]), sum(rows[1]), sum(rows[2])))
