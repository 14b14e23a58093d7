"""C-ABI checks that need no GPU: the library loads, exports every symbol the header declares, host-only entry points
agree with the oracle, and argument validation returns the documented error codes before touching the device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sdy():
    import sdy_amd

    return sdy_amd


def test_header_symbols_are_exported_and_bound(sdy):
    hdr = open(os.path.join(ROOT, "include", "sdy_amd.h")).read()
    declared = set(re.findall(r"\b(sdy_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = C.CDLL(sdy.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sdy_amd.h but not exported"
    # ... and nothing else: the product library is built with -fvisibility=hidden behind csrc/exports.map, and the in-kernel
    # stamp read-backs (sdy_*_debug_stamps) exist in -DSDY_STAMPS measurement builds only
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", sdy.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, f"exported but not declared: {sorted(exported - declared)}; missing: {sorted(declared - exported)}"
    from sdy_amd import _lib

    assert declared == set(_lib.SIGNATURES), sorted(declared ^ set(_lib.SIGNATURES))
    assert sdy.lib.sdy_version() >= 100


@pytest.mark.parametrize("grid,gid", [("equiangular", 0), ("legendre-gauss", 1)])
@pytest.mark.parametrize("nlat,nlon", [(32, 64), (180, 360)])
def test_host_tables_match_oracle(sdy, grid, gid, nlat, nlon):
    from oracle.sht import quadrature, sht_tables

    L, M = nlat, nlon // 2 + 1
    pct = np.zeros((M, L, nlat))
    w = np.zeros(nlat)
    th = np.zeros(nlat)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert sdy.lib.sdy_sht_tables_host(nlat, nlon, L, M, gid, vp(pct), vp(w), vp(th)) == 0
    p_ref, wq_ref, _, _ = sht_tables(nlat, nlon, L, M, grid)
    th_ref, w_ref = quadrature(nlat, grid)
    assert np.abs(th - th_ref).max() < 1e-14 and np.abs(w - w_ref).max() < 1e-14
    assert np.abs(pct - p_ref).max() < 1e-12
    # what the device sees: fp32 casts of pct and pct*w are identical to the oracle's
    assert (pct.astype(np.float32) == p_ref.astype(np.float32)).mean() > 0.9999
    assert np.abs((pct * w).astype(np.float32) - wq_ref.astype(np.float32)).max() < 1e-9


def test_error_codes_without_gpu(sdy):
    lib = sdy.lib
    assert lib.sdy_sht_tables_host(1, 64, 4, 4, 0, None, None, None) == -1          # SDY_ERR_ARG
    assert lib.sdy_sht_tables_host(16, 32, 16, 17, 7, None, None, None) == -1       # bad grid
    h = C.c_void_p()
    assert lib.sdy_sht_plan_create(30, 62, 30, 32, 0, C.byref(h)) == -3             # nlon % 4 -> SDY_ERR_ALIGN
    assert lib.sdy_sht_plan_create(32, 64, 32, 40, 0, C.byref(h)) == -1             # mmax > nlon/2+1
    assert lib.sdy_sht_plan_create(32, 4 * 7, 32, 15, 0, C.byref(h)) == -2          # nlon/2 = 14 has a factor 7
    assert lib.sdy_cold_update(None, None, None, None, 8, None) == -1
    assert lib.sdy_sfno_workspace_floats(None, 4) == 0
    assert b"multiple of 4" in lib.sdy_error_string(-3)
    with pytest.raises(sdy.SdyError):
        sdy._lib.check(-2, "x")


def test_no_cpu_fallback(sdy):
    import torch

    with pytest.raises(RuntimeError, match="GPU only"):
        sdy.RealSHT(32, 64)(torch.zeros(1, 4, 32, 64))
    with pytest.raises(RuntimeError, match="GPU only"):
        sdy.ops.cold_update(torch.zeros(4), torch.zeros(4), None)
    net = sdy.SphericalFourierNeuralOperatorNet(4, 4, spatial_shape_in=(32, 64), embed_dim=8, num_layers=1)
    with pytest.raises(RuntimeError, match="GPU only"):
        net(torch.zeros(1, 4, 32, 64))


def test_philox_known_answers():
    """Random123 known-answer vectors of Philox4x32-10 pin the round function, the constants and the key schedule of the
    oracle's generator; the dropout stream runs the same code for seven rounds (oracle.philox.ROUNDS, SDY_PHILOX_ROUNDS), for
    which Random123 publishes known-answer vectors as well."""
    from oracle.philox import ROUNDS, philox4x32

    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]
    for ctr, key, exp in kat:
        got = tuple(int(x) for x in philox4x32(*ctr, *key, rounds=10))
        assert got == exp
    assert ROUNDS == 7
    # Random123's known-answer vectors of philox4x32 with SEVEN rounds (kat_vectors: "philox4x32 7 ..."), the count the dropout
    # stream runs: held by the oracle's generator AND by the library's own (the __host__ __device__ function the kernels
    # inline, evaluated on the host through sdy_dropout_stream_words; no GPU needed)
    kat7 = [((0, 0, 0, 0), (0, 0), (0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48)),
            ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662)),
            ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
             (0x4DFCCABA, 0x190A87F0, 0xC47362BA, 0xB6B5242A))]
    import ctypes as C

    import sdy_amd

    assert sdy_amd.lib.sdy_dropout_stream_rounds() == ROUNDS          # a rebuilt library cannot silently change seeded results
    for ctr, key, exp in kat7:
        assert tuple(int(x) for x in philox4x32(*ctr, *key)) == exp
        out = (C.c_uint32 * 4)()
        assert sdy_amd.lib.sdy_dropout_stream_words(*ctr, *key, out) == 0
        assert tuple(out) == exp


def test_element_dropout_stream_definition():
    """The documented element-dropout stream (include/sdy_amd.h), re-derived value by value from the generator:
    one Philox call per (4 channels, pixel pair n / n + 32), word = channel & 3, half-word = bit 5 of the pixel,
    keep <=> half-word >= floor(p * 2^16)."""
    import numpy as np

    from oracle.philox import drop_threshold16, element_keep_mask, philox4x32

    seed, call, layer, kind, p = 0x1234_5678_9ABC, 7, 3, 1, 0.13
    B, C, H, W, boff = 2, 8, 3, 40, 5
    mask = element_keep_mask(seed, call, layer, kind, p, B, C, H, W, batch_offset=boff)
    thr = drop_threshold16(p)
    assert thr == int(np.float32(p) * 65536.0)
    rng = np.random.default_rng(0)
    for _ in range(200):
        b, c, pix = int(rng.integers(B)), int(rng.integers(C)), int(rng.integers(H * W))
        words = philox4x32(np.uint32(pix & ~32), np.uint32((b + boff) * (C // 4) + (c >> 2)), np.uint32(2 * layer + kind),
                              np.uint32(call), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        w = int(words[c & 3])
        half = (w >> 16) if (pix >> 5) & 1 else (w & 0xFFFF)
        assert mask.reshape(B, C, H * W)[b, c, pix] == (1.0 if half >= thr else 0.0)
    # the two pixels of a pair come from the same call: flipping bit 5 changes only the half-word
    assert 0.05 < 1.0 - mask.mean() < 0.25


def test_dropout_streams_are_independent_across_members_kinds_calls_and_layers():
    """The documented stream (include/sdy_amd.h), as the device replays it bit for bit (tests/test_gpu_sfno.py): keep rates
    match p, and the decisions of different trajectories, of the hidden / output dropout of one layer, of consecutive calls
    and of different layers are uncorrelated (|r| < 4 / sqrt(n)) -- what 25 ensemble members sharing one (seed, call) need in
    order to be independent samples (reference: independent draws of torch's generator, layers.py:76-78, drop_path.py:19)."""
    from oracle.philox import drop_path_keep, element_keep_mask

    seed, p, C, H, W = 0x5EED_1234, 0.1, 32, 16, 64
    n = C * H * W
    mem = element_keep_mask(seed, 3, 2, 0, p, 25, C, H, W, batch_offset=0).reshape(25, -1)      # 25 members, one call
    assert np.all(np.abs(mem.mean(1) - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n))
    r = np.corrcoef(mem)
    off = r[~np.eye(25, dtype=bool)]
    assert np.abs(off).max() < 4.5 / np.sqrt(n), f"members correlate: max |r| {np.abs(off).max():.4f}"
    # a rank's rows draw the streams of their GLOBAL trajectory index: rows 7.. of a whole-job batch == batch_offset = 7
    part = element_keep_mask(seed, 3, 2, 0, p, 5, C, H, W, batch_offset=7).reshape(5, -1)
    assert np.array_equal(part, mem[7:12])

    def corr(a, b):
        return abs(float(np.corrcoef(a.reshape(-1), b.reshape(-1))[0, 1]))

    base = element_keep_mask(seed, 3, 2, 0, p, 2, C, H, W)
    for other in (element_keep_mask(seed, 3, 2, 1, p, 2, C, H, W),      # output dropout of the same layer
                  element_keep_mask(seed, 4, 2, 0, p, 2, C, H, W),      # next call
                  element_keep_mask(seed, 3, 3, 0, p, 2, C, H, W),      # next layer
                  element_keep_mask(seed + 1, 3, 2, 0, p, 2, C, H, W)):  # another seed
        assert corr(base, other) < 4.5 / np.sqrt(2 * n)
        assert not np.array_equal(base, other)
    # drop path: per-trajectory flags at the documented rate, independent of the layer / call
    dp = np.stack([drop_path_keep(seed, c, l, 0.3, 4096) for c in range(2) for l in range(2)])
    assert np.all(np.abs(dp.mean(1) - 0.7) < 4 * np.sqrt(0.21 / 4096))
    rr = np.corrcoef(dp)
    assert np.abs(rr[~np.eye(4, dtype=bool)]).max() < 4.5 / np.sqrt(4096)


def test_torch_restatement_of_the_dropout_stream_equals_the_numpy_one():
    """oracle.philox.element_keep_mask_torch (int64 tensors, 16-bit limbs; what the full-depth GPU parity test evaluates on the
    device for speed) against the numpy restatement, bit for bit, including a ragged pixel count and a batch offset."""
    import torch

    from oracle.philox import element_keep_mask, element_keep_mask_torch

    for args in ((0xABCDEF0123456789, 3, 5, 1, 0.13, 2, 16, 7, 50, 5), (77, 1, 2, 0, 0.1, 1, 64, 45, 64, 0),
                 (1000, 9, 7, 0, 0.1, 3, 8, 3, 21, 40)):
        a = element_keep_mask(*args[:9], batch_offset=args[9])
        b = element_keep_mask_torch(*args[:9], batch_offset=args[9])
        assert torch.equal(torch.from_numpy(a), b), args
