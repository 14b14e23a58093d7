"""Operator-level host wrappers (one HIP launch each) mirroring the reference callables of the SFNO block.

These exist for drop-in use at the reference's inner boundary and for the per-op parity tests; the network
(`sfno.py`) does not go through them -- it calls the fused native forward.  torch is used for buffer ownership
and layout views only; all arithmetic is in libsdy_amd.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from ._lib import SdyConvArgs, SdyMlpArgs, SdyPairArgs, check, current_stream, lib, ptr


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("sdy_amd ops run on the GPU only (no CPU fallback)")
    return t.to(torch.float32).contiguous()


def _aux(t: torch.Tensor, dev) -> torch.Tensor:
    """small side inputs (weights, biases, coefficients) may arrive on the host; move them next to x"""
    return t.detach().to(dev, torch.float32).contiguous()


def contract_dhconv(x: torch.Tensor, weight: torch.Tensor, separable: bool = False,
                    operator_type: str = "dhconv", gemm_mode: Optional[str] = None) -> torch.Tensor:
    """`_contract_dense_pytorch(x, weight, separable=False, operator_type="dhconv")`
    (`src/models/sfno/factorizations.py:165-186` -> `contractions.py:159-169`):
    x (B, Ci, L, M) complex64, weight (Ci, Co, L, 2) real -> (B, Co, L, M) complex64,
    out[b,o,l,m] = sum_i x[b,i,l,m] * w[i,o,l].  Entries with m > l are treated as structurally zero
    (they are for every SHT output) and returned as zero.
    """
    if separable or operator_type != "dhconv":
        raise NotImplementedError("only the non-separable dhconv contraction is on the hot path")
    B, Ci, L, M = x.shape
    Co = weight.shape[1]
    assert weight.shape == (Ci, Co, L, 2), f"weight shape {tuple(weight.shape)}"
    mtr = min(M, L)
    xr = torch.view_as_real(x.to(torch.complex64).contiguous())          # (B,Ci,L,M,2)
    cs_in = xr[:, :, :, :mtr].permute(2, 3, 0, 4, 1).contiguous()         # [l][m][b][ri][c]
    cs_out = torch.zeros(L, mtr, B, 2, Co, dtype=torch.float32, device=x.device)
    w_host = weight.detach().to(torch.float32).cpu().contiguous()
    from ._lib import default_gemm_mode
    with torch.cuda.device(x.device):
        mode = gemm_mode or default_gemm_mode()
        if mode == "h3" and lib.sdy_dhconv_frag_supported(Ci, Co):
            wp = torch.empty(lib.sdy_dhconv_frag_pack_bytes(L), dtype=torch.uint8, device=x.device)
            sc = C.c_float()
            check(lib.sdy_dhconv_frag_pack(ptr(w_host), L, ptr(wp), C.byref(sc)), "sdy_dhconv_frag_pack")
            check(lib.sdy_dhconv_frag(ptr(cs_in), ptr(wp), sc.value, ptr(cs_out), L, mtr, B, current_stream()),
                  "sdy_dhconv_frag")
        elif mode == "h3":
            wp = torch.empty(lib.sdy_dhconv_h3_pack_bytes(Ci, Co, L), dtype=torch.uint8, device=x.device)
            sc = C.c_float()
            check(lib.sdy_dhconv_h3_pack_weight(ptr(w_host), Ci, Co, L, ptr(wp), C.byref(sc)), "sdy_dhconv_h3_pack_weight")
            check(lib.sdy_dhconv_h3(ptr(cs_in), ptr(wp), sc.value, ptr(cs_out), L, mtr, B, Ci, Co, current_stream()),
                  "sdy_dhconv_h3")
        else:
            wp = torch.empty(L * 2 * Ci * Co, dtype=torch.float32, device=x.device)
            check(lib.sdy_dhconv_pack_weight(ptr(w_host), Ci, Co, L, ptr(wp), current_stream()), "sdy_dhconv_pack_weight")
            check(lib.sdy_dhconv(ptr(cs_in), ptr(wp), ptr(cs_out), L, mtr, B, Ci, Co, current_stream()), "sdy_dhconv")
    out = torch.zeros(B, Co, L, M, 2, dtype=torch.float32, device=x.device)
    out[:, :, :, :mtr] = cs_out.permute(2, 4, 0, 1, 3)
    tri = (torch.arange(M, device=x.device)[None, :] <= torch.arange(L, device=x.device)[:, None])
    out = out * tri[None, None, :, :, None]
    return torch.view_as_complex(out.contiguous())


def instnorm_coeffs(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                    scale_shift: Optional[torch.Tensor] = None, eps: float = 1e-6):
    """Per-(b,c) coefficients (a, d) with InstanceNorm2d(x)*(1+scale)+shift == a*x+d
    (`src/models/sfno/sfnonet.py:280-299,641-648`).  scale_shift: (B, 2C) = (scale | shift) or None."""
    x = _f32c(x)
    B, Cc, H, W = x.shape
    a = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    d = torch.empty_like(a)
    g, b_ = _aux(gamma, x.device), _aux(beta, x.device)
    ss = _aux(scale_shift, x.device) if scale_shift is not None else None
    with torch.cuda.device(x.device):
        check(lib.sdy_instnorm_coeffs(ptr(x), B, Cc, H * W, ptr(g), ptr(b_), ptr(ss), 2 * Cc, eps, ptr(a), ptr(d),
                                      current_stream()), "sdy_instnorm_coeffs")
    return a, d


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            pre_affine=None, add: Optional[torch.Tensor] = None, add_mode: int = 0, gelu: bool = False,
            drop_p: float = 0.0, keep_mask: Optional[torch.Tensor] = None, seed: int = 0, call: int = 0,
            stream_id: int = 0, batch_offset: int = 0, batch_scale: Optional[torch.Tensor] = None,
            kernel_tag: int = 0, out: Optional[torch.Tensor] = None, wt_prepared=None, h3: bool = False,
            h3_prepared=None, frag_prepared=None, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Conv2d(kernel_size=1) with the block's fused prologue/epilogue (see include/sdy_amd.h, sdy_conv1x1).
    x (B,Cin,H,W), weight (Cout,Cin[,1,1])."""
    x = _f32c(x)
    B, Cin, H, W = x.shape
    w2 = weight.detach().to(torch.float32).reshape(weight.shape[0], -1)
    Cout = w2.shape[0]
    assert w2.shape[1] == Cin
    ldw = (Cout + 3) // 4 * 4
    if wt_prepared is not None:      # (Cin, ldw) transposed weight already on the device (benchmark loops)
        wt = wt_prepared
    else:
        wt = torch.zeros(Cin, ldw, dtype=torch.float32, device=x.device)
        wt[:, :Cout] = w2.t().to(x.device)
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    a = SdyConvArgs()
    a.x, a.x_bstride = ptr(x), Cin * H * W
    a.wt, a.ldw = ptr(wt), ldw
    a.out, a.out_bstride = ptr(out), Cout * H * W
    a.B, a.Cin, a.Cout, a.HW = B, Cin, Cout, H * W
    keep = [x, wt, out]
    if h3 or h3_prepared is not None:     # split-fp16 3-pass MFMA path (include/sdy_amd.h, sdy_h3_pack_weight)
        if h3_prepared is None:
            h3_prepared = pack_h3(w2, x.device)
        a.w_h3, a.w_h3_scale = ptr(h3_prepared[0]), h3_prepared[1]
        keep.append(h3_prepared[0])
    if pre_affine is not None:
        pa, pd = _aux(pre_affine[0], x.device), _aux(pre_affine[1], x.device)
        a.pa, a.pd = ptr(pa), ptr(pd)
        keep += [pa, pd]
    if bias is not None:
        bb = _aux(bias, x.device)
        a.bias = ptr(bb)
        keep.append(bb)
    if add is not None:
        ad = _aux(add, x.device)
        a.add = ptr(ad)
        a.add_bstride = 0 if ad.shape[0] == 1 and B > 1 else Cout * H * W
        a.add_mode = add_mode
        keep.append(ad)
    a.act = 1 if gelu else 0
    a.drop_p = drop_p
    if keep_mask is not None:
        km = _aux(keep_mask, x.device)
        a.keep_mask = ptr(km)
        keep.append(km)
    a.seed, a.call, a.stream_id, a.batch_offset = seed, call, stream_id, batch_offset
    a.kernel_tag = kernel_tag
    if frag_prepared is not None:   # persistent 256 -> 256 kernel (pack_conv256); the only path that can emit statistics
        a.w_frag, a.w_frag_scale = ptr(frag_prepared[0]), frag_prepared[1]
        keep.append(frag_prepared[0])
    if stats is not None:       # (B, Cout, 2) float64 on the device, zeroed by the caller: (sum, sumsq) of `out` are added
        assert stats.dtype == torch.float64 and stats.is_cuda and stats.is_contiguous() and stats.numel() == B * Cout * 2
        a.stats = ptr(stats)
    if batch_scale is not None:
        bs = _aux(batch_scale, x.device)
        a.batch_scale = ptr(bs)
        keep.append(bs)
    with torch.cuda.device(x.device):
        check(lib.sdy_conv1x1(C.byref(a), current_stream()), "sdy_conv1x1")
    return out


def pack_conv256(weight: torch.Tensor, device):
    """(256, Cin <= 384) fp32 weight -> (per-wave MFMA fragment stream, scale) for conv1x1(..., frag_prepared=...)."""
    w = weight.detach().to("cpu", torch.float32).reshape(weight.shape[0], -1).contiguous()
    if not lib.sdy_conv256_h3_supported(w.shape[1], w.shape[0]):
        raise NotImplementedError("the persistent conv kernel needs 256 output and at most 384 input channels")
    buf = torch.empty(lib.sdy_conv256_h3_pack_bytes_cin(w.shape[1]), dtype=torch.uint8, device=device)
    sc = C.c_float()
    with torch.cuda.device(device):
        check(lib.sdy_conv256_h3_pack_cin(ptr(w), w.shape[1], ptr(buf), C.byref(sc)), "sdy_conv256_h3_pack_cin")
    return buf, sc.value


def pack_h3(weight: torch.Tensor, device):
    """(Cout, Cin) fp32 weight -> (device buffer, scale) for the split-fp16 conv path."""
    w = weight.detach().to("cpu", torch.float32).reshape(weight.shape[0], -1).contiguous()
    Cout, Cin = w.shape
    nbytes = lib.sdy_h3_pack_bytes(Cout, Cin)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    sc = C.c_float()
    with torch.cuda.device(device):
        check(lib.sdy_h3_pack_weight(ptr(w), Cout, Cin, ptr(buf), C.byref(sc)), "sdy_h3_pack_weight")
    return buf, sc.value


def pack_mlp_h3(w1: torch.Tensor, w2: torch.Tensor, device):
    """fc1 (hidden, E) and fc2 (E, hidden) weights -> (per-wave fragment stream, scale1, scale2) for `mlp_fused`."""
    a = w1.detach().to("cpu", torch.float32).reshape(w1.shape[0], -1).contiguous()
    b = w2.detach().to("cpu", torch.float32).reshape(w2.shape[0], -1).contiguous()
    hidden, E = a.shape
    assert b.shape == (E, hidden)
    if not lib.sdy_mlp_h3_supported(E, hidden):
        raise NotImplementedError(f"fused MLP kernel supports E=256, hidden=512 (got {E}, {hidden}); use conv1x1")
    buf = torch.empty(lib.sdy_mlp_h3_pack_bytes(E, hidden), dtype=torch.uint8, device=device)
    s1, s2 = C.c_float(), C.c_float()
    with torch.cuda.device(device):
        check(lib.sdy_mlp_h3_pack(ptr(a), ptr(b), E, hidden, ptr(buf), C.byref(s1), C.byref(s2)), "sdy_mlp_h3_pack")
    return (buf, s1.value, s2.value)


def mlp_fused(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, *,
              pre_affine=None, add: Optional[torch.Tensor] = None, drop_p: float = 0.0, seed: int = 0, call: int = 0,
              stream_fc1: int = 0, stream_fc2: int = 1, batch_offset: int = 0,
              batch_scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
              prepared=None, add_affine=None, stats: Optional[torch.Tensor] = None,
              keep_masks=None) -> torch.Tensor:
    """The block's MLP (`src/models/sfno/layers.py:73-80`) with the norm affine, both dropouts, DropPath scale and the
    residual add in one launch (include/sdy_amd.h, sdy_mlp_h3).  Same arithmetic and dropout stream as
    conv1x1(fc1, gelu, stream_fc1) -> conv1x1(fc2, stream_fc2, add_mode=2).  `keep_masks=(hidden mask (B, hidden, H, W),
    output mask (B, E, H, W))` (tests): 0/1 keep decisions replace the Philox stream, as `keep_mask` does in conv1x1."""
    x = _f32c(x)
    B, E, H, W = x.shape
    if prepared is None:
        prepared = pack_mlp_h3(w1, w2, x.device)
    hidden = b1.numel()
    if out is None:
        out = torch.empty_like(x)
    a = SdyMlpArgs()
    a.x, a.x_bstride = ptr(x), E * H * W
    a.w, a.w1_scale, a.w2_scale = ptr(prepared[0]), prepared[1], prepared[2]
    bb1, bb2 = _aux(b1, x.device), _aux(b2, x.device)
    a.b1, a.b2 = ptr(bb1), ptr(bb2)
    a.out, a.out_bstride = ptr(out), E * H * W
    keep = [x, bb1, bb2, out, prepared]
    if pre_affine is not None:
        pa, pd = _aux(pre_affine[0], x.device), _aux(pre_affine[1], x.device)
        a.pa, a.pd = ptr(pa), ptr(pd)
        keep += [pa, pd]
    if add is not None:
        ad = _aux(add, x.device)
        a.add, a.add_bstride = ptr(ad), E * H * W
        keep.append(ad)
        if add_affine is not None:      # residual = add_affine[0][b, c] * add + add_affine[1][b, c]
            ra, rd = _aux(add_affine[0], x.device), _aux(add_affine[1], x.device)
            a.add_a, a.add_d = ptr(ra), ptr(rd)
            keep += [ra, rd]
    a.B, a.E, a.hidden, a.HW = B, E, hidden, H * W
    a.drop_p = drop_p
    a.seed, a.call, a.stream_fc1, a.stream_fc2, a.batch_offset = seed, call, stream_fc1, stream_fc2, batch_offset
    if batch_scale is not None:
        bs = _aux(batch_scale, x.device)
        a.batch_scale = ptr(bs)
        keep.append(bs)
    if stats is not None:       # (B, E, 2) float64 on the device, zeroed by the caller: (sum, sumsq) of `out` are added
        assert stats.dtype == torch.float64 and stats.is_cuda and stats.is_contiguous() and stats.numel() == B * E * 2
        a.stats = ptr(stats)
    if keep_masks is not None:
        kh, ko = _aux(keep_masks[0], x.device), _aux(keep_masks[1], x.device)
        assert kh.numel() == B * hidden * H * W and ko.numel() == B * E * H * W
        a.keep_hidden, a.keep_out = ptr(kh), ptr(ko)
        keep += [kh, ko]
    with torch.cuda.device(x.device):
        check(lib.sdy_mlp_h3(C.byref(a), current_stream()), "sdy_mlp_h3")
    return out


def pack_pair_h3(w1: torch.Tensor, w2: torch.Tensor, device):
    """(hidden, Cin) and (Cout, hidden) weights -> (per-wave fragment stream, scale1, scale2) for `conv_pair`."""
    a = w1.detach().to("cpu", torch.float32).reshape(w1.shape[0], -1).contiguous()
    b = w2.detach().to("cpu", torch.float32).reshape(w2.shape[0], -1).contiguous()
    hidden, Cin = a.shape
    Cout = b.shape[0]
    assert b.shape == (Cout, hidden)
    if not lib.sdy_pair_h3_supported(Cin, hidden, Cout):
        raise NotImplementedError(f"fused conv pair supports hidden=256 with (Cin<=144, Cout=256) or (Cin<=416, Cout<=64); "
                                  f"got {Cin}->{hidden}->{Cout}; use conv1x1 twice")
    buf = torch.empty(lib.sdy_pair_h3_pack_bytes(Cin, hidden, Cout), dtype=torch.uint8, device=device)
    s1, s2 = C.c_float(), C.c_float()
    with torch.cuda.device(device):
        check(lib.sdy_pair_h3_pack(ptr(a), ptr(b), Cin, hidden, Cout, ptr(buf), C.byref(s1), C.byref(s2)), "sdy_pair_h3_pack")
    return (buf, s1.value, s2.value)


def conv_pair(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, *,
              add: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, prepared=None,
              stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv(w2) . GELU . conv(w1, b1) + add in one launch (include/sdy_amd.h, sdy_pair_h3): the encoder
    (`src/models/sfno/sfnonet.py:609-618`, add = position embedding of shape (1, Cout, H, W)) and the decoder (`:734-744`)."""
    x = _f32c(x)
    B, Cin, H, W = x.shape
    if prepared is None:
        prepared = pack_pair_h3(w1, w2, x.device)
    hidden, Cout = w1.shape[0], w2.shape[0]
    if out is None:
        out = torch.empty(B, Cout, H, W, dtype=torch.float32, device=x.device)
    a = SdyPairArgs()
    a.x, a.x_bstride = ptr(x), Cin * H * W
    a.w, a.w1_scale, a.w2_scale = ptr(prepared[0]), prepared[1], prepared[2]
    keep = [x, out, prepared]
    if b1 is not None:
        bb1 = _aux(b1, x.device)
        a.b1 = ptr(bb1)
        keep.append(bb1)
    a.out, a.out_bstride = ptr(out), Cout * H * W
    if add is not None:
        ad = _aux(add, x.device)
        a.add, a.add_bstride = ptr(ad), (0 if ad.shape[0] == 1 and B > 1 else Cout * H * W)
        keep.append(ad)
    a.B, a.Cin, a.hidden, a.Cout, a.HW = B, Cin, hidden, Cout, H * W
    if stats is not None:
        assert stats.dtype == torch.float64 and stats.is_cuda and stats.is_contiguous() and stats.numel() == B * Cout * 2
        a.stats = ptr(stats)
    with torch.cuda.device(x.device):
        check(lib.sdy_pair_h3(C.byref(a), current_stream()), "sdy_pair_h3")
    return out


def instnorm_from_stats(stats: torch.Tensor, HW: int, gamma: torch.Tensor, beta: torch.Tensor, scale_shift=None,
                        eps: float = 1e-6):
    """(B, C, 2) float64 (sum, sumsq) statistics written by a producer epilogue -> the (a, d) of `instnorm_coeffs`;
    clears `stats`."""
    B, Cc = stats.shape[0], stats.shape[1]
    dev = stats.device
    g, b_ = _aux(gamma, dev), _aux(beta, dev)
    ss = _aux(scale_shift, dev) if scale_shift is not None else None
    a = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    d = torch.empty(B, Cc, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.sdy_instnorm_from_stats(ptr(stats), B, Cc, HW, ptr(g), ptr(b_), ptr(ss), 2 * Cc if ss is not None else 0,
                                          eps, ptr(a), ptr(d), current_stream()), "sdy_instnorm_from_stats")
    return a, d


def cold_update(x_s: torch.Tensor, x_ip_next: torch.Tensor, x_ip_s: Optional[torch.Tensor]) -> torch.Tensor:
    """x_s + (x_ip_next - x_ip_s)  (`src/diffusion/dyffusion.py:517-519`); x_ip_s None means x_ip_s == x_s."""
    a, b = _f32c(x_s), _f32c(x_ip_next)
    c = _f32c(x_ip_s) if x_ip_s is not None else None
    out = torch.empty_like(a)
    with torch.cuda.device(a.device):
        check(lib.sdy_cold_update(ptr(a), ptr(b), ptr(c), ptr(out), a.numel(), current_stream()), "sdy_cold_update")
    return out


def concat_channels(tensors) -> torch.Tensor:
    """torch.cat(tensors, dim=1) for up to 4 NCHW float32 tensors (`src/diffusion/dyffusion.py:655-661`)."""
    ts = [_f32c(t) for t in tensors]
    if len(ts) == 1:
        return ts[0]
    assert 1 <= len(ts) <= 4
    B, _, H, W = ts[0].shape
    chans = [t.shape[1] for t in ts]
    out = torch.empty(B, sum(chans), H, W, dtype=torch.float32, device=ts[0].device)
    srcs = (C.c_void_p * len(ts))(*[ptr(t) for t in ts])
    ch = (C.c_int * len(ts))(*chans)
    with torch.cuda.device(out.device):
        check(lib.sdy_concat_channels(srcs, ch, len(ts), ptr(out), B, H * W, current_stream()), "sdy_concat_channels")
    return out


FLAG_NONFINITE, FLAG_F16_RANGE = 1, 2


def status_flags(reset: bool = True, device=None) -> int:
    """Sticky status word of a device (`sdy_status_flags`): FLAG_NONFINITE | FLAG_F16_RANGE bits set by kernels since the last
    reset.  Synchronises the current stream of that device."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = C.c_uint(0)
    with torch.cuda.device(dev):
        check(lib.sdy_status_flags(C.byref(out), 1 if reset else 0, current_stream()), "sdy_status_flags")
    return int(out.value)


RANGE_CLASSES = ("conv_h3 x tile", "mlp_h3 x tile", "dh_h3 coefficient rows", "Legendre analysis input (at rfft360's stores)",
                 "Legendre synthesis input (at dh_h3's stores)")


class range_headroom:
    """Context manager around `sdy_range_headroom*` (include/sdy_amd.h): while the block runs, the split-precision kernels of
    the default path record the largest magnitude they stage as fp16 (pre-scale included: the number that must stay below
    65504), per consumer class.  Afterwards `.max_staged` = {class: value} and `.factor` = {class: 65504 / value} -- "x N below
    the cliff" for a checkpoint's first contact with the library, instead of SDY_FLAG_F16_RANGE's pass / fail.  Debug aid: one
    atomic per tile while enabled."""

    LIMIT = 65504.0

    def __enter__(self):
        buf = (C.c_float * len(RANGE_CLASSES))()
        with torch.cuda.device(torch.cuda.current_device()):
            check(lib.sdy_range_headroom(buf, 1, current_stream()), "sdy_range_headroom")     # drop stale maxima
        check(lib.sdy_range_headroom_enable(1))
        self.max_staged, self.factor = {}, {}
        return self

    def __exit__(self, *exc):
        check(lib.sdy_range_headroom_enable(0))
        buf = (C.c_float * len(RANGE_CLASSES))()
        with torch.cuda.device(torch.cuda.current_device()):
            check(lib.sdy_range_headroom(buf, 1, current_stream()), "sdy_range_headroom")
        self.max_staged = {n: float(buf[i]) for i, n in enumerate(RANGE_CLASSES)}
        self.factor = {n: (self.LIMIT / v if v > 0.0 else float("inf")) for n, v in self.max_staged.items()}
        return False


def raise_on_status_flags(flags: int, where: str) -> None:
    if not flags:
        return
    why = []
    if flags & FLAG_F16_RANGE:
        why.append("an activation exceeded the fp16 range of the split-precision kernels (|value| * 16 >= 65504, i.e. "
                   "|value| >= 4094 after normalisation)")
    if flags & FLAG_NONFINITE:
        why.append("an InstanceNorm statistic came out inf / NaN (a tensor inside the network holds non-finite values)")
    from ._lib import SdyError
    raise SdyError(f"{where}: " + "; ".join(why) + ".  Inputs are expected standardised; if the data really need the range, "
                   "run the fp32-MFMA path: SDY_GEMM_MODE=f32 (or gemm_mode='f32').")


class stage_timer:
    """Context manager around `sdy_profile_*`: per-stage HIP-event timing of every SFNO forward issued inside the block
    (events on the launch stream).  `.stages` afterwards: {stage name: (launches, total_ms)}; `.rows`: {stage name: batch rows
    summed over its launches} (with the drop-path skip a block's kernels run on the kept trajectories only).  Measurement only."""

    def __enter__(self):
        n = lib.sdy_profile_stage_count()
        check(lib.sdy_profile_read((C.c_double * n)(), (C.c_long * n)(), n))      # drop stale records
        check(lib.sdy_profile_enable(1))
        self.stages, self.rows = {}, {}
        return self

    def __exit__(self, *exc):
        check(lib.sdy_profile_enable(0))
        n = lib.sdy_profile_stage_count()
        ms, cnt, rows = (C.c_double * n)(), (C.c_long * n)(), (C.c_long * n)()
        check(lib.sdy_profile_read_rows(ms, cnt, rows, n), "sdy_profile_read_rows")
        self.stages = {lib.sdy_profile_stage_name(i).decode(): (int(cnt[i]), float(ms[i])) for i in range(n) if cnt[i]}
        self.rows = {lib.sdy_profile_stage_name(i).decode(): int(rows[i]) for i in range(n) if cnt[i]}
        return False
