"""dhconv fragment kernel at small batches (the multi-GPU operating point): time per launch and an output digest, so that
runs with SDY_DH_NJ=1 / 2 (32- / 64-row tiles) can be compared bit for bit.  Run on the GPU box.
    [SDY_DH_NJ=1|2] python tools/dh_small_bench.py [B ...]"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdy_amd  # noqa: F401
from sdy_amd._lib import check, current_stream, lib, ptr

E, L, mtr = 256, 180, 180
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(7)
w = torch.randn(E, E, L, 2, generator=g) / 16
wf = torch.empty(lib.sdy_dhconv_frag_pack_bytes(L), dtype=torch.uint8, device=dev)
sc = C.c_float()
check(lib.sdy_dhconv_frag_pack(ptr(w.contiguous()), L, ptr(wf), C.byref(sc)))
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 6, 8, 25]:
    Cs = torch.randn(L * mtr * B * 2 * E, generator=g).to(dev)
    Cs2 = torch.zeros_like(Cs)
    run = lambda: check(lib.sdy_dhconv_frag(ptr(Cs), ptr(wf), sc.value, ptr(Cs2), L, mtr, B, current_stream()))  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    # rows m <= l only are written: digest those
    out = Cs2.view(L, mtr, B, 2 * E)
    mask = (torch.arange(mtr, device=dev)[None, :] <= torch.arange(L, device=dev)[:, None])
    vals = out[mask].contiguous()
    dig = hashlib.sha1(vals.cpu().numpy().tobytes()).hexdigest()[:12]
    print("B=%2d  %8.1f us per launch   digest %s  |out| %.6e" % (B, e0.elapsed_time(e1) / 20 * 1e3, dig, float(vals.double().norm())), flush=True)
