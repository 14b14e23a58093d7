"""Phase timeline of the folded Legendre kernel (leg_par.hip) for four sampled workgroups: SDY_LEG_STAMPS=1 python tools/leg_stamps.py"""
import ctypes as C
import os
import sys

os.environ["SDY_LEG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdy_amd
from sdy_amd._lib import lib, ptr, check, current_stream
from sdy_amd.sht import ShtPlan

B, E, H, W, L, M = 25, 256, 180, 360, 180, 181
plan = ShtPlan.get(H, W, L, M, "legendre-gauss", 0)
mtr = plan.mtr
Xf = torch.randn(mtr * H * B * 2 * E, device="cuda")
Cs = torch.zeros(L * mtr * B * 2 * E, device="cuda")
lib.sdy_leg_par_debug_stamps.argtypes = [C.c_void_p]
lib.sdy_leg_par_debug_stamps.restype = C.c_int
names = ["issue loads", "loads arrive", "split + barrier", "MFMA", "stores issued"]
for fwd in (1, 0):
    for _ in range(3):
        if fwd:
            check(lib.sdy_legendre_fwd(plan.handle, ptr(Xf), ptr(Cs), B, E, current_stream()))
        else:
            check(lib.sdy_legendre_inv(plan.handle, ptr(Cs), ptr(Xf), B, E, current_stream()))
    buf = (C.c_uint64 * 96)()
    assert lib.sdy_leg_par_debug_stamps(buf) == 0
    v = list(buf)
    print("analysis" if fwd else "synthesis")
    for wg in range(4):
        for w in range(3):
            s = v[(wg * 3 + w) * 8:(wg * 3 + w) * 8 + 8]
            d = [s[i + 1] - s[i] for i in range(5)]
            print(f"  wg(x={'150' if wg >= 2 else '37'}, m={'90' if wg & 1 else '20'}) wave {w}: total {s[5] - s[0]:6d}: " +
                  ", ".join(f"{n} {x}" for n, x in zip(names, d)))
