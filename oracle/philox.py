"""Oracle: numpy restatement of the product's counter-based dropout stream.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference draws its inference-time dropout / drop-path masks from torch's
global generator (`nn.Dropout` in `src/models/sfno/layers.py:76-78`,
`torch.rand` in `src/models/modules/drop_path.py:19`); bitwise reproduction of
that stream on another device is impossible, so the product defines its own
stream: Philox4x32-7 (Salmon et al., SC'11: seven rounds is the smallest count of
the family that passes BigCrush; same constants as Random123 / cuRAND / torch's
CUDA generator, which run ten), keyed and countered as documented in
`include/sdy_amd.h` ("Dropout stream").  This file restates it so the parity
tests can compare the device masks bit for bit.

  key      = (seed_lo, seed_hi)
  counter  = (c0, c1, stream, call)
  element dropout (kind 0/1):  c0 = pixel index p (h*W + w)
                               c1 = b_global * (C/4) + (ch >> 2);  word = ch & 3
                               stream = 2*layer + kind
  drop path:                   c0 = b_global, c1 = 0xFFFFFFFF, stream = 0x1000 + layer, word 0
  keep  <=>  word >= floor(p * 2**32)
"""
from __future__ import annotations

import os

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


ROUNDS = 7   # SDY_PHILOX_ROUNDS of csrc/common.h


def philox4x32(c0, c1, c2, c3, k0, k1, rounds: int = ROUNDS):
    """Vectorised Philox4x32-`rounds`.  All inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    shape = np.broadcast(c0, c1, c2, c3).shape
    c0, c1, c2, c3 = (np.broadcast_to(c, shape).copy() for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(rounds):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if r < rounds - 1:
                k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
                k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def drop_threshold(p: float) -> int:
    """uint32 threshold: keep <=> word >= threshold.  p is taken at fp32 precision, as the C ABI carries it."""
    return min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF)


def drop_threshold16(p: float) -> int:
    """16-bit threshold of the element dropout: keep <=> half-word >= threshold (p at fp32 precision)."""
    p32 = float(np.float32(p))
    return min(max(int(p32 * 65536.0), 1 if p32 > 0.0 else 0), 0xFFFF)   # p > 0 never rounds to "no dropout"


def _global_rows(B: int, batch_offset: int, rows) -> np.ndarray:
    """Global trajectory index of every batch row: `batch_offset + b` (what the C ABI does), or an explicit table for an
    oracle that walks the trajectories in another order than the device batch."""
    if rows is None:
        return np.arange(B, dtype=np.uint64) + np.uint64(batch_offset)
    rows = np.asarray(rows, dtype=np.uint64)
    assert rows.shape == (B,)
    return rows


def element_keep_mask(seed: int, call: int, layer: int, kind: int, p: float,
                      B: int, C: int, H: int, W: int, batch_offset: int = 0, rows=None) -> np.ndarray:
    """Keep mask (B, C, H, W) of 0/1 float32 for MLP dropout (kind 0 = hidden, 1 = output).

    One Philox call covers 4 channels x the pixel pair (n & ~32, n | 32): word = ch & 3, low half-word for the pixel
    with bit 5 clear, high half-word for its partner (include/sdy_amd.h)."""
    assert C % 4 == 0
    thr = np.uint32(drop_threshold16(p))
    HW = H * W
    npix = np.arange(HW, dtype=np.uint32)
    base = npix[(npix & np.uint32(32)) == 0]          # one generator call per pixel PAIR: the pixel with bit 5 clear ...
    partner = base | np.uint32(32)                    # ... serves its partner from the high half-words
    has_partner = partner < HW
    b = _global_rows(B, batch_offset, rows)[:, None, None]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    keep = np.empty((B, C // 4, 4, HW), dtype=np.float32)

    def fill(g0, g1):   # channel groups g0 .. g1 - 1 (numpy releases the GIL inside its loops: the chunks run in parallel)
        g = np.arange(g0, g1, dtype=np.uint64)[None, :, None]
        c1 = ((b * np.uint64(C // 4) + g) & MASK32).astype(np.uint32)
        words = philox4x32(base[None, None, :], c1, np.uint32(2 * layer + kind), np.uint32(call & 0xFFFFFFFF), k0, k1)
        w = np.stack(words, axis=2)                   # (B, g1 - g0, 4, pairs)
        keep[:, g0:g1, :, base] = (w & np.uint32(0xFFFF)) >= thr
        keep[:, g0:g1, :, partner[has_partner]] = ((w >> np.uint32(16)) >= thr)[..., has_partner]

    ngroups = C // 4
    nthreads = min(ngroups, max(1, min(32, (os.cpu_count() or 1))))
    if nthreads == 1 or B * C * HW < (1 << 22):
        fill(0, ngroups)
    else:
        from concurrent.futures import ThreadPoolExecutor

        edges = np.linspace(0, ngroups, nthreads + 1).astype(int)
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            list(pool.map(lambda i: fill(int(edges[i]), int(edges[i + 1])), range(nthreads)))
    return keep.reshape(B, C, H, W)


def drop_path_keep(seed: int, call: int, layer: int, p: float, B: int, batch_offset: int = 0, rows=None) -> np.ndarray:
    """Keep flags (B,) of 0/1 float32 for drop path of `layer`."""
    thr = np.uint32(drop_threshold(p))
    b = (_global_rows(B, batch_offset, rows) & MASK32).astype(np.uint32)
    w0, _, _, _ = philox4x32(b, np.uint32(0xFFFFFFFF), np.uint32(0x1000 + layer),
                                np.uint32(call & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return (w0 >= thr).astype(np.float32)


def _mulhilo32_torch(m: int, c):
    """(hi32, lo32) of m * c for an int64 tensor c holding uint32 values: 16-bit limbs keep every product below 2^63."""
    cl, ch = c & 0xFFFF, c >> 16
    p, q = cl * m, ch * m
    s = p + ((q & 0xFFFF) << 16)
    return (q >> 16) + (s >> 32), s & 0xFFFFFFFF


def element_keep_mask_torch(seed: int, call: int, layer: int, kind: int, p: float, B: int, C: int, H: int, W: int,
                            batch_offset: int = 0, rows=None, device="cpu"):
    """`element_keep_mask` restated on torch int64 tensors (same stream definition, same result bit for bit:
    tests/test_capi_cpu.py) -- the full-depth, full-size parity tests need 160 masks of 33 M decisions each, minutes of numpy;
    on `device="cuda"` torch evaluates the same arithmetic in milliseconds.  Still test infrastructure: plain torch integer
    ops, nothing of the product.  Returns a float32 CPU tensor (B, C, H, W)."""
    import torch

    assert C % 4 == 0
    thr = int(drop_threshold16(p))
    HW = H * W
    dev = torch.device(device)
    npix = torch.arange(HW, dtype=torch.int64, device=dev)
    base = npix[(npix & 32) == 0]
    partner = base | 32
    has_partner = partner < HW
    b = torch.as_tensor(_global_rows(B, batch_offset, rows).astype(np.int64), device=dev)[:, None, None]
    g = torch.arange(C // 4, dtype=torch.int64, device=dev)[None, :, None]
    c0 = base[None, None, :].expand(B, C // 4, -1).clone()
    c1 = ((b * (C // 4) + g) & 0xFFFFFFFF).expand(-1, -1, base.numel()).clone()
    c2 = torch.full_like(c0, (2 * layer + kind) & 0xFFFFFFFF)
    c3 = torch.full_like(c0, call & 0xFFFFFFFF)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for r in range(ROUNDS):
        hi0, lo0 = _mulhilo32_torch(int(M0), c0)
        hi1, lo1 = _mulhilo32_torch(int(M1), c2)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        if r < ROUNDS - 1:
            k0 = (k0 + int(W0)) & 0xFFFFFFFF
            k1 = (k1 + int(W1)) & 0xFFFFFFFF
    w = torch.stack((c0, c1, c2, c3), dim=2)                  # (B, C/4, 4, pairs)
    keep = torch.empty(B, C // 4, 4, HW, dtype=torch.float32, device=dev)
    keep[..., base] = ((w & 0xFFFF) >= thr).to(torch.float32)
    keep[..., partner[has_partner]] = ((w >> 16) >= thr)[..., has_partner].to(torch.float32)
    return keep.reshape(B, C, H, W).cpu()
