"""Per-phase timeline of one wave of the dhconv fragment kernel (dev aid; run on the GPU box)."""
import sys, os, ctypes as C
os.environ["SDY_DH_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdy_amd
from sdy_amd._lib import lib, ptr, check, current_stream
B, E, L, mtr = (int(sys.argv[1]) if len(sys.argv) > 1 else 25), 256, 180, 180
dev = torch.device("cuda")
Cs = torch.randn(L * mtr * B * 2 * E, device=dev); Cs2 = torch.zeros_like(Cs)
w = torch.randn(E, E, L, 2) / 16
wf = torch.empty(lib.sdy_dhconv_frag_pack_bytes(L), dtype=torch.uint8, device=dev)
sc = C.c_float()
check(lib.sdy_dhconv_frag_pack(ptr(w.contiguous()), L, ptr(wf), C.byref(sc)))
for _ in range(3):
    check(lib.sdy_dhconv_frag(ptr(Cs), ptr(wf), sc.value, ptr(Cs2), L, mtr, B, current_stream()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    check(lib.sdy_dhconv_frag(ptr(Cs), ptr(wf), sc.value, ptr(Cs2), L, mtr, B, current_stream()))
e1.record(); torch.cuda.synchronize()
print("ms", e0.elapsed_time(e1) / 5)
buf = (C.c_uint64 * 384)()
lib.sdy_dhconv_frag_debug_stamps.argtypes = [C.c_void_p]; lib.sdy_dhconv_frag_debug_stamps.restype = C.c_int
assert lib.sdy_dhconv_frag_debug_stamps(buf) == 0
v = list(buf)
names = ["x regs -> LDS", "barrier", "MFMA phase", "epilogue stores", "end barrier", "-> next tile"]
t0 = min(v[w * 8] for w in range(8))
print("tile 0 of the window: stamp times relative to the earliest wave (ticks); columns = waves 0..7")
for i in range(6):
    print("  stamp %d %-18s" % (i, "(before " + names[i] + ")"), " ".join("%7d" % (v[w * 8 + i] - t0) for w in range(8)))
print("  next tile start          ", " ".join("%7d" % (v[(8 + w) * 8] - t0) for w in range(8)))

ts, lt = v[256:320], v[320:384]
n = max(i for i in range(64) if ts[i]) if any(ts) else 0
print("every tile of the sampled workgroup: (degree l, tile t) and its duration in ticks")
print("  " + "  ".join("(%d,%d) %d" % (lt[i] // 1000, lt[i] % 1000, ts[i + 1] - ts[i]) for i in range(n) if ts[i + 1]))
print("  total ticks first -> last stamp", ts[n] - ts[0], "over", n, "tiles")
