"""Per-stage micro-benchmark of the spectral path at production shapes (run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdy_amd
from sdy_amd._lib import lib, ptr, check, current_stream
from sdy_amd.sht import ShtPlan

B = int(sys.argv[1]) if len(sys.argv) > 1 else int(os.environ.get("SDY_STAGE_B", "8"))
E, H, W, L, M = 256, 180, 360, 180, 181
dev = torch.device("cuda")
plan = ShtPlan.get(H, W, L, M, "legendre-gauss", 0)
mtr = plan.mtr
x = torch.randn(B, E, H, W, device=dev)
a = torch.rand(B, E, device=dev) + 0.5
d = torch.randn(B, E, device=dev) * 0.1
xn = torch.empty_like(x)
y = torch.empty_like(x)
Xf = torch.zeros(mtr * H * B * 2 * E, device=dev)
Cs = torch.zeros(L * mtr * B * 2 * E, device=dev)
Cs2 = torch.zeros_like(Cs)
bias = torch.randn(E, device=dev)
gamma = torch.ones(E, device=dev); beta = torch.zeros(E, device=dev)
w = torch.randn(E, E, L, 2) / 16
wp = torch.empty(L * 2 * E * E, device=dev)
check(lib.sdy_dhconv_pack_weight(ptr(w.contiguous()), E, E, L, ptr(wp), 0))
st = lambda: current_stream()
T = 66.355200e6 * B  # bytes of one activation tensor

def bench(name, f, nbytes, flops=0, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:28s} B={B:2d} {ms:8.3f} ms  {nbytes/ms/1e9:7.2f} TB/s" + (f"  {flops/ms/1e9:7.1f} TF/s" if flops else ""), flush=True)

bench("instnorm_coeffs", lambda: check(lib.sdy_instnorm_coeffs(ptr(x), B, E, H*W, ptr(gamma), ptr(beta), None, 0, 1e-6, ptr(a), ptr(d), st())), T)
a = torch.rand(B, E, device=dev) + 0.5
bench("rfft (affine, store xn)", lambda: check(lib.sdy_rfft_lon(plan.handle, ptr(x), ptr(a), ptr(d), ptr(xn), ptr(Xf), B, E, st())), 3 * T)
bench("rfft (no xn store)", lambda: check(lib.sdy_rfft_lon(plan.handle, ptr(x), ptr(a), ptr(d), None, ptr(Xf), B, E, st())), 2 * T)
bench("legendre_fwd", lambda: check(lib.sdy_legendre_fwd(plan.handle, ptr(Xf), ptr(Cs), B, E, st())), 1.5 * T, 3.0e9 * B)
bench("dhconv", lambda: check(lib.sdy_dhconv(ptr(Cs), ptr(wp), ptr(Cs2), L, mtr, B, E, E, st())), T + 94.4e6, 8.54e9 * B)
wf = torch.empty(lib.sdy_dhconv_frag_pack_bytes(L), dtype=torch.uint8, device=dev)
import ctypes as _C
_sc = _C.c_float()
check(lib.sdy_dhconv_frag_pack(ptr(w.contiguous()), L, ptr(wf), _C.byref(_sc)))
bench("dhconv (fragment stream)", lambda: check(lib.sdy_dhconv_frag(ptr(Cs), ptr(wf), _sc.value, ptr(Cs2), L, mtr, B, st())), T + 94.4e6 * 2, 8.54e9 * B)
bench("legendre_inv", lambda: check(lib.sdy_legendre_inv(plan.handle, ptr(Cs2), ptr(Xf), B, E, st())), 1.5 * T, 3.0e9 * B)
bench("irfft (+bias)", lambda: check(lib.sdy_irfft_lon(plan.handle, ptr(Xf), ptr(bias), ptr(y), B, E, st())), 2 * T)
