"""Times the fused encoder / decoder pair (sdy_pair_h3) against the two launches it replaces (run on the GPU box)."""
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
for name, cin, cout in [("encoder  65", 65, 256), ("encoder 128", 128, 256), ("decoder 321", 321, 63), ("decoder 384", 384, 63)]:
    x = torch.randn(B, cin, H, W, device="cuda")
    w1 = torch.randn(E, cin, generator=g) / cin ** 0.5; b1 = (torch.randn(E, generator=g) * .1).cuda()
    w2 = torch.randn(cout, E, generator=g) / 16
    pos = torch.randn(1, cout, H, W, device="cuda") if cout == 256 else None
    st = torch.zeros(B, cout, 2, dtype=torch.float64, device="cuda") if cout == 256 else None
    prep = sdy.ops.pack_pair_h3(w1, w2, "cuda")
    out = torch.empty(B, cout, H, W, device="cuda"); hid = torch.empty(B, E, H, W, device="cuda")
    fr1 = sdy.ops.pack_conv256(w1, "cuda") if cin <= 384 else None
    h31 = sdy.ops.pack_h3(w1, "cuda"); 
    fr2 = sdy.ops.pack_conv256(w2, "cuda") if cout == 256 else None
    h32 = sdy.ops.pack_h3(w2, "cuda")
    f = lambda: sdy.ops.conv_pair(x, w1, b1, w2, add=pos, out=out, prepared=prep, stats=st)
    def two():
        sdy.ops.conv1x1(x, w1, b1, gelu=True, frag_prepared=fr1, h3_prepared=h31, out=hid, wt_prepared=hid)
        sdy.ops.conv1x1(hid, w2, None, add=pos, add_mode=2, frag_prepared=fr2, h3_prepared=h32, stats=st if fr2 is not None else None,
                        out=out, wt_prepared=out)
    tf, tt = timeit(f), timeit(two)
    gb = B * H * W * 4 * (cin + cout) / 1e9
    print(f"{name} -> 256 -> {cout}: fused {tf:.3f} ms ({gb / tf * 1e3:.0f} GB/s alg) | two launches {tt:.3f} ms  -> x{tt / tf:.2f}")
    if os.environ.get("SDY_PAIR_STAMPS"):
        import ctypes as C
        from importlib import import_module
        L = import_module("sdy_amd._lib").lib
        L.sdy_pair_h3_debug_stamps.argtypes = [C.c_void_p]; L.sdy_pair_h3_debug_stamps.restype = C.c_int
        f(); buf = (C.c_uint64 * 64)(); assert L.sdy_pair_h3_debug_stamps(buf) == 0
        v = list(buf)
        names = {1: "x split p0 + barrier", 3: "fc1 p0, split p1 + barriers", 4: "fc1 (last part)", 5: "chain alone", 6: "barrier",
                 7: "fc2(0) [|| chain(1)]", 8: "fc2(1) + barrier + prefetch issue", 9: "acc -> LDS + barriers", 10: "stores (+ stats)"}
        for t in range(3):
            s_ = v[t * 16:(t + 1) * 16]; nxt = v[(t + 1) * 16]
            idx = [i for i in range(11) if i == 0 or i in names and (i != 3 or cin > 256)]
            parts = [f"{names[b]} {s_[b] - s_[a]}" for a, b in zip(idx[:-1], idx[1:])]
            print(f"    (7 -> 11 prefetch issue {s_[11] - s_[7]}, 11 -> 12 fc2(1) {s_[12] - s_[11]}, 12 -> 8 holes + reset {s_[8] - s_[12]})")
            print(f"  tile {t}: total {nxt - s_[0]}: " + ", ".join(parts) + f", end barrier -> next {nxt - s_[10]}")
