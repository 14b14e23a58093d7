"""Times the persistent 256->256 conv kernel against the tile GEMM (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd as sdy
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
B, E, H, W = 25, 256, 180, 360
g = torch.Generator().manual_seed(0)
w = torch.randn(E, E, generator=g) / 16; bias = (torch.randn(E, generator=g) * .1).cuda()
x = torch.randn(B, E, H, W, device="cuda"); y = torch.randn(B, E, H, W, device="cuda")
pa = torch.ones(B, E, device="cuda"); pd = torch.zeros(B, E, device="cuda")
frag = sdy.ops.pack_conv256(w, "cuda"); h3 = sdy.ops.pack_h3(w, "cuda")
st = torch.zeros(B, E, 2, dtype=torch.float64, device="cuda")
out = torch.empty_like(x)
f = lambda: sdy.ops.conv1x1(x, w, bias, pre_affine=(pa, pd), add=y, add_mode=1, gelu=True, frag_prepared=frag, stats=st, out=out, wt_prepared=out)
u = lambda: sdy.ops.conv1x1(x, w, bias, pre_affine=(pa, pd), add=y, add_mode=1, gelu=True, h3_prepared=h3, out=out, wt_prepared=out, kernel_tag=3)
s = lambda: sdy.ops.instnorm_coeffs(out, bias, bias)
tf, tu, ts = timeit(f), timeit(u), timeit(s)
print(f"conv_h3 (with stats) {tf:.3f} ms ({3*B*E*H*W*4/tf/1e6:.0f} GB/s) | tile GEMM {tu:.3f} ms + stats pass {ts:.3f} ms")
if os.environ.get("SDY_CONV_STAMPS"):
    import ctypes as C
    from importlib import import_module
    L = import_module("sdy_amd._lib").lib
    L.sdy_conv256_h3_debug_stamps.argtypes = [C.c_void_p]; L.sdy_conv256_h3_debug_stamps.restype = C.c_int
    f(); buf = (C.c_uint64 * 64)(); assert L.sdy_conv256_h3_debug_stamps(buf) == 0
    v = list(buf); names = ["x regs->LDS+barrier", "prefetch issue", "MFMA", "barrier", "acc->LDS+barrier", "store loop (GELU)", "stats", "end barrier->next"]
    for t in range(3):
        s_ = v[t * 8:(t + 1) * 8]; nxt = v[(t + 1) * 8]
        d = [s_[i + 1] - s_[i] for i in range(7)] + [nxt - s_[7]]
        print(f"tile {t}: total {sum(d)}: " + ", ".join(f"{n} {x}" for n, x in zip(names, d)))
