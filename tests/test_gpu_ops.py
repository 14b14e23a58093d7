"""Per-op parity of the HIP kernels against the CPU oracle (all through the C ABI via the host wrappers).

Tolerances (fp32 path; north_star bound is 1e-4 relative L2 end to end):
  single op      : 2e-6 relative L2  (fp32 FMA-chain GEMMs / FFT vs torch CPU fp32)
"""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_OP = 2e-6
GRIDS = [(32, 64), (180, 360)]


@pytest.fixture(scope="module")
def sdy():
    import sdy_amd

    return sdy_amd


def _gen(seed):
    return torch.Generator(device="cpu").manual_seed(seed)


MODES = ["f32", "h3"]   # fp32-input MFMA / split-fp16 3-pass MFMA: same tolerances


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
@pytest.mark.parametrize("nlat,nlon", GRIDS)
def test_real_sht_forward(sdy, nlat, nlon, grid, mode):
    from oracle.sht import RealSHT as ORef

    C = 8 if nlat > 64 else 12
    x = torch.randn(2, C, nlat, nlon, generator=_gen(1))
    ref = ORef(nlat, nlon, lmax=nlat, mmax=nlon // 2 + 1, grid=grid).float()(x)
    got = sdy.RealSHT(nlat, nlon, lmax=nlat, mmax=nlon // 2 + 1, grid=grid, gemm_mode=mode).float()(x.cuda())
    assert got.shape == ref.shape and got.dtype == torch.complex64
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"RealSHT {nlat}x{nlon} {grid}: rel L2 {err:.3e}"
    # structural zeros (m > l) must be exact zeros
    l = torch.arange(nlat)[:, None]
    m = torch.arange(nlon // 2 + 1)[None, :]
    assert (got.cpu()[..., (m > l)] == 0).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
@pytest.mark.parametrize("nlat,nlon", GRIDS)
def test_inverse_real_sht(sdy, nlat, nlon, grid, mode):
    from oracle.sht import InverseRealSHT as ORef

    C = 8 if nlat > 64 else 12
    L, M = nlat, nlon // 2 + 1
    c = torch.randn(2, C, L, M, dtype=torch.complex64, generator=_gen(2))  # dense, incl. m > l and imag of m = 0
    ref = ORef(nlat, nlon, lmax=L, mmax=M, grid=grid).float()(c)
    got = sdy.InverseRealSHT(nlat, nlon, lmax=L, mmax=M, grid=grid, gemm_mode=mode).float()(c.cuda())
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"InverseRealSHT {nlat}x{nlon} {grid}: rel L2 {err:.3e}"


def test_sht_roundtrip_bandlimited(sdy):
    """size-independent property at the full grid: SHT(ISHT(c)) == c for band-limited c on the Gauss grid."""
    nlat, nlon = 180, 360
    L, M = nlat, nlon // 2 + 1
    c = torch.randn(1, 4, L, M, dtype=torch.complex64, generator=_gen(3))
    l = torch.arange(L)[:, None]
    m = torch.arange(M)[None, :]
    c = c * (m <= l)
    c[..., 0] = c[..., 0].real + 0j
    isht = sdy.InverseRealSHT(nlat, nlon, lmax=L, mmax=M, grid="legendre-gauss")
    sht = sdy.RealSHT(nlat, nlon, lmax=L, mmax=M, grid="legendre-gauss")
    c2 = sht(isht(c.cuda()))
    err = rel_l2(c2, c)
    assert err < 2e-5, f"round trip rel L2 {err:.3e}"


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
def test_sht_vs_scipy_known_answers_180x360(sdy, grid):
    """The HIP transforms against scipy directly (no oracle in between), production size: a field assembled from
    `scipy.special.sph_harm_y` has known coefficients; synthesis of those coefficients gives the field back."""
    from scipy import special

    nlat, nlon = 180, 360
    L, M = nlat, nlon // 2 + 1
    if grid == "legendre-gauss":
        x, _ = special.roots_legendre(nlat)
        theta = np.arccos(x)[::-1].copy()
    else:
        theta = np.linspace(0.0, np.pi, nlat)
    phi = 2 * np.pi * np.arange(nlon) / nlon
    modes = [(0, 0, 1.5, 0.0), (5, 2, 0.5, -0.25), (40, 40, -1.0, 2.0), (89, 17, 0.75, 0.5), (60, 0, 2.0, 0.0)]
    if grid == "legendre-gauss":
        modes += [(179, 179, 1.0, -1.0), (179, 1, -0.5, 0.25), (150, 97, 0.3, 0.6)]
    f = np.zeros((nlat, nlon))
    want = np.zeros((L, M), dtype=np.complex128)
    for l, m, cr, ci in modes:
        Y = special.sph_harm_y(l, m, theta[:, None], phi[None, :])
        f += (complex(cr, ci) * Y).real * (2.0 if m > 0 else 1.0)
        want[l, m] = complex(cr, ci)
    xf = torch.from_numpy(f).float()[None, None].expand(1, 4, -1, -1).contiguous()
    got = sdy.RealSHT(nlat, nlon, grid=grid)(xf.cuda()).cpu().to(torch.complex128).numpy()[0, 1]
    ok = np.ones(L, bool) if grid == "legendre-gauss" else (np.arange(L) <= (nlat - 1) - max(l for l, *_ in modes))
    err = np.linalg.norm(got[ok] - want[ok]) / np.linalg.norm(want)
    assert err < 2e-6, f"RealSHT vs scipy known answer ({grid}): {err:.3e}"
    c = torch.from_numpy(want).to(torch.complex64)[None, None].expand(1, 4, -1, -1).contiguous()
    back = sdy.InverseRealSHT(nlat, nlon, grid=grid)(c.cuda()).cpu().double().numpy()[0, 2]
    err = np.linalg.norm(back - f) / np.linalg.norm(f)
    assert err < 2e-6, f"InverseRealSHT vs scipy field ({grid}): {err:.3e}"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,E,L,M", [(2, 8, 32, 33), (1, 256, 180, 181), (3, 16, 20, 11), (3, 256, 20, 11), (5, 256, 9, 10)])
def test_dhconv(sdy, B, E, L, M, mode):
    g = _gen(4)
    x = torch.randn(B, E, L, M, dtype=torch.complex64, generator=g)
    l = torch.arange(L)[:, None]
    m = torch.arange(M)[None, :]
    x = x * (m <= l)
    w = torch.randn(E, E, L, 2, generator=g) / np.sqrt(E)
    ref = torch.einsum("bixy,iox->boxy", x, torch.view_as_complex(w))
    got = sdy.ops.contract_dhconv(x.cuda(), w.cuda(), gemm_mode=mode)
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"dhconv rel L2 {err:.3e}"


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 32, 64), (1, 16, 180, 360)])
def test_instnorm_coeffs(sdy, B, C, H, W):
    g = _gen(5)
    x = torch.randn(B, C, H, W, generator=g) * 3 + 1.5
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ss = 0.3 * torch.randn(B, 2 * C, generator=g)
    xn = torch.nn.functional.instance_norm(x, weight=gamma, bias=beta, eps=1e-6)
    scale, shift = ss[:, :, None, None].chunk(2, dim=1)
    ref = xn * (scale + 1) + shift
    a, d = sdy.ops.instnorm_coeffs(x.cuda(), gamma.cuda(), beta.cuda(), ss.cuda())
    got = a.cpu()[:, :, None, None] * x + d.cpu()[:, :, None, None]
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"instnorm rel L2 {err:.3e}"
    a2, d2 = sdy.ops.instnorm_coeffs(x.cuda(), gamma.cuda(), beta.cuda(), None)
    got2 = a2.cpu()[:, :, None, None] * x + d2.cpu()[:, :, None, None]
    assert rel_l2(got2, xn) < TOL_OP


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 8, 16, 32, 64), (1, 70, 256, 180, 360), (2, 256, 34, 32, 64),
                                            (1, 36, 130, 20, 12)])
def test_conv1x1_plain_and_epilogues(sdy, B, Cin, Cout, H, W):
    g = _gen(6)
    F = torch.nn.functional
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    add = torch.randn(B, Cout, H, W, generator=g)
    pa = 1 + 0.2 * torch.randn(B, Cin, generator=g)
    pd = 0.2 * torch.randn(B, Cin, generator=g)
    bs = torch.tensor([0.0, 1.25][:B]) if B == 2 else torch.tensor([1.25])
    xc = x.cuda()
    # plain
    err = rel_l2(sdy.ops.conv1x1(xc, w), F.conv2d(x, w))
    assert err < TOL_OP, f"plain {err:.3e}"
    # bias + GELU
    err = rel_l2(sdy.ops.conv1x1(xc, w, b, gelu=True), F.gelu(F.conv2d(x, w, b)))
    assert err < TOL_OP, f"bias+gelu {err:.3e}"
    # pre-activation add (inner skip): GELU(conv + bias + add)
    err = rel_l2(sdy.ops.conv1x1(xc, w, b, add=add.cuda(), add_mode=1, gelu=True), F.gelu(F.conv2d(x, w, b) + add))
    assert err < TOL_OP, f"add_pre {err:.3e}"
    # prologue affine + batch scale + post add (fc2 / residual)
    xa = x * pa[:, :, None, None] + pd[:, :, None, None]
    ref = F.conv2d(xa, w, b) * bs[:, None, None, None] + add
    got = sdy.ops.conv1x1(xc, w, b, pre_affine=(pa.cuda(), pd.cuda()), add=add.cuda(), add_mode=2, batch_scale=bs.cuda())
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"affine+scale+add_post {err:.3e}"
    # broadcast add (pos_embed)
    pe = torch.randn(1, Cout, H, W, generator=g)
    if B > 1:
        err = rel_l2(sdy.ops.conv1x1(xc, w, add=pe.cuda(), add_mode=2), F.conv2d(x, w) + pe)
        assert err < TOL_OP, f"pos_embed add {err:.3e}"


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 8, 16, 32, 64), (1, 256, 512, 45, 64)])
def test_conv1x1_dropout_matches_philox_oracle(sdy, B, Cin, Cout, H, W):
    from oracle.philox import element_keep_mask

    g = _gen(7)
    F = torch.nn.functional
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    p, seed, call, layer, kind, boff = 0.1, 0x1234567887654321, 7, 3, 1, 5
    keep = torch.from_numpy(element_keep_mask(seed, call, layer, kind, p, B, Cout, H, W, batch_offset=boff))
    frac = float(keep.mean())
    assert abs(frac - 0.9) < 0.01, f"keep fraction {frac}"
    ref = F.gelu(F.conv2d(x, w, b)) * keep * (1.0 / (1.0 - p))
    got = sdy.ops.conv1x1(x.cuda(), w, b, gelu=True, drop_p=p, seed=seed, call=call, stream_id=2 * layer + kind,
                          batch_offset=boff)
    gc = got.cpu()
    assert (gc[keep == 0] == 0).all(), "a dropped element is non-zero: device mask differs from the Philox oracle"
    big = (ref.abs() > 1e-3)    # kept elements that are not tiny must be non-zero on the device too
    assert (gc[big] != 0).all(), "a kept element is zero: device mask differs from the Philox oracle"
    assert rel_l2(got, ref) < TOL_OP
    # injected mask path
    km = (torch.rand(B, Cout, H, W, generator=g) > 0.3).float()
    got2 = sdy.ops.conv1x1(x.cuda(), w, b, gelu=True, drop_p=0.3, keep_mask=km.cuda())
    ref2 = F.gelu(F.conv2d(x, w, b)) * km * (1.0 / (1.0 - 0.3))
    assert rel_l2(got2, ref2) < TOL_OP


def test_cold_update_and_concat(sdy):
    g = _gen(8)
    a, b, c = (torch.randn(2, 5, 18, 36, generator=g) for _ in range(3))
    got = sdy.ops.cold_update(a.cuda(), b.cuda(), c.cuda()).cpu()
    assert torch.equal(got, a + (b - c))
    got0 = sdy.ops.cold_update(a.cuda(), b.cuda(), None).cpu()
    assert torch.equal(got0, a + (b - a))
    cat = sdy.ops.concat_channels([a.cuda(), b.cuda()[:, :1], c.cuda()]).cpu()
    assert torch.equal(cat, torch.cat([a, b[:, :1], c], 1))


def test_errors_are_reported(sdy):
    with pytest.raises(sdy.SdyError):
        sdy.RealSHT(30, 62, grid="equiangular")(torch.zeros(1, 4, 30, 62).cuda())  # nlon % 4 != 0
    with pytest.raises(RuntimeError):
        sdy.RealSHT(32, 64)(torch.zeros(1, 4, 32, 64))  # CPU tensor: no fallback


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 8, 16, 32, 64), (1, 70, 256, 45, 64), (2, 256, 34, 32, 64),
                                            (1, 512, 256, 45, 64), (1, 36, 130, 20, 12)])
def test_conv1x1_split_fp16_mode(sdy, B, Cin, Cout, H, W):
    """gemm_mode "h3": 3-pass split-fp16 MFMA must hold the same per-op tolerance as the fp32-MFMA kernel."""
    g = _gen(9)
    F = torch.nn.functional
    x = torch.randn(B, Cin, H, W, generator=g) * 1.7 + 0.3
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = 0.1 * torch.randn(Cout, generator=g)
    add = torch.randn(B, Cout, H, W, generator=g)
    pa = 1 + 0.2 * torch.randn(B, Cin, generator=g)
    pd = 0.2 * torch.randn(B, Cin, generator=g)
    xd = x.double() * pa.double()[:, :, None, None] + pd.double()[:, :, None, None]
    ref64 = F.conv2d(xd, w.double(), b.double())
    got = sdy.ops.conv1x1(x.cuda(), w, b, pre_affine=(pa.cuda(), pd.cuda()), h3=True)
    got32 = sdy.ops.conv1x1(x.cuda(), w, b, pre_affine=(pa.cuda(), pd.cuda()))
    e_h3, e_f32 = rel_l2(got, ref64), rel_l2(got32, ref64)
    assert e_h3 < TOL_OP, f"h3 vs fp64: {e_h3:.3e} (fp32 kernel: {e_f32:.3e})"
    assert e_h3 < 4 * e_f32 + 2e-7, f"h3 {e_h3:.3e} much worse than fp32 kernel {e_f32:.3e}"
    ref = F.gelu(F.conv2d(x * pa[:, :, None, None] + pd[:, :, None, None], w, b) + add)
    got = sdy.ops.conv1x1(x.cuda(), w, b, pre_affine=(pa.cuda(), pd.cuda()), add=add.cuda(), add_mode=1, gelu=True, h3=True)
    assert rel_l2(got, ref) < TOL_OP
    # tiny and large magnitudes (lo parts in the fp16 subnormal range / hi parts near the top of the range)
    for s in (1e-3, 50.0):
        got = sdy.ops.conv1x1((x * s).cuda(), w, None, h3=True)
        err = rel_l2(got, F.conv2d(x.double() * s, w.double()))
        assert err < 5e-6, f"scale {s}: {err:.3e}"


@pytest.mark.parametrize("B,H,W,drop", [(2, 8, 40, 0.0), (2, 8, 40, 0.1), (1, 45, 64, 0.1), (3, 6, 36, 0.25), (3, 87, 96, 0.1)])
def test_mlp_fused_matches_fp64_and_unfused(sdy, B, H, W, drop):
    """sdy_mlp_h3 (hidden activation kept on the CU) == fc1 -> GELU -> dropout -> fc2 -> dropout -> drop-path scale ->
    + residual, against an fp64 restatement with the Philox oracle's masks, and against the two-launch path."""
    from oracle.philox import element_keep_mask

    E, Hd = 256, 512
    g = _gen(21)
    F = torch.nn.functional
    x = torch.randn(B, E, H, W, generator=g) * 1.3 + 0.2
    w1 = torch.randn(Hd, E, 1, 1, generator=g) / np.sqrt(E)
    b1 = 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(E, Hd, 1, 1, generator=g) / np.sqrt(Hd)
    b2 = 0.1 * torch.randn(E, generator=g)
    res = torch.randn(B, E, H, W, generator=g)
    pa = 1 + 0.2 * torch.randn(B, E, generator=g)
    pd = 0.2 * torch.randn(B, E, generator=g)
    bs = torch.tensor([1.25, 0.0, 0.8][:B])
    seed, call, layer, boff = 0xABCDEF0123456789, 3, 5, 7
    hid = F.gelu(F.conv2d(x.double() * pa.double()[:, :, None, None] + pd.double()[:, :, None, None], w1.double(),
                          b1.double()))
    if drop > 0:
        k1 = torch.from_numpy(element_keep_mask(seed, call, layer, 0, drop, B, Hd, H, W, batch_offset=boff)).double()
        k2 = torch.from_numpy(element_keep_mask(seed, call, layer, 1, drop, B, E, H, W, batch_offset=boff)).double()
        hid = hid * k1 / (1.0 - drop)
    o = F.conv2d(hid, w2.double(), b2.double())
    if drop > 0:
        o = o * k2 / (1.0 - drop)
    ref = o * bs.double()[:, None, None, None] + res.double()
    kw = dict(drop_p=drop, seed=seed, call=call, batch_offset=boff)
    got = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, pre_affine=(pa.cuda(), pd.cuda()), add=res.cuda(),
                            stream_fc1=2 * layer, stream_fc2=2 * layer + 1, batch_scale=bs.cuda(), **kw)
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"fused MLP vs fp64: {err:.3e}"
    h = sdy.ops.conv1x1(x.cuda(), w1, b1, pre_affine=(pa.cuda(), pd.cuda()), gelu=True, stream_id=2 * layer, h3=True, **kw)
    two = sdy.ops.conv1x1(h, w2, b2, add=res.cuda(), add_mode=2, batch_scale=bs.cuda(), stream_id=2 * layer + 1, h3=True,
                          **kw)
    assert rel_l2(got, two) < 5e-6
    if drop > 0:   # identical masks: without residual / drop-path exactly the same elements are zero
        f0 = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, stream_fc1=2 * layer, stream_fc2=2 * layer + 1, **kw).cpu()
        assert torch.equal(f0 == 0, k2 == 0), "fc2 dropout mask differs from the Philox oracle"
    # residual given as an affine of another tensor (norm folded into its consumer)
    ra, rd = 1 + 0.3 * torch.randn(B, E, generator=g), 0.3 * torch.randn(B, E, generator=g)
    got = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, pre_affine=(pa.cuda(), pd.cuda()), add=res.cuda(),
                            add_affine=(ra.cuda(), rd.cuda()), stream_fc1=2 * layer, stream_fc2=2 * layer + 1,
                            batch_scale=bs.cuda(), **kw)
    ref_aff = o * bs.double()[:, None, None, None] + res.double() * ra.double()[:, :, None, None] + rd.double()[:, :, None, None]
    assert rel_l2(got, ref_aff) < TOL_OP
    # statistics of the stored output (the next block's InstanceNorm) from the epilogue
    st = torch.zeros(B, E, 2, dtype=torch.float64, device="cuda")
    got_s = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, pre_affine=(pa.cuda(), pd.cuda()), add=res.cuda(),
                              stream_fc1=2 * layer, stream_fc2=2 * layer + 1, batch_scale=bs.cuda(), stats=st, **kw)
    gd = got_s.double().cpu()
    want = torch.stack([gd.sum((2, 3)), (gd * gd).sum((2, 3))], -1)
    assert torch.allclose(st.cpu(), want, rtol=1e-5, atol=1e-6 * H * W)
    gamma, beta = 1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    a_s, d_s = sdy.ops.instnorm_from_stats(st, H * W, gamma, beta)
    a_r, d_r = sdy.ops.instnorm_coeffs(got_s, gamma, beta)
    assert rel_l2(a_s, a_r) < 1e-5 and rel_l2(d_s, d_r) < 1e-5
    assert float(st.abs().max()) == 0.0, "statistics buffer must be cleared for reuse"
    # no affine / no residual / no scale
    got = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2)
    ref = F.conv2d(F.gelu(F.conv2d(x.double(), w1.double(), b1.double())), w2.double(), b2.double())
    assert rel_l2(got, ref) < TOL_OP


@pytest.mark.parametrize("B,H,W", [(2, 8, 40), (1, 45, 64)])
def test_mlp_fused_with_injected_masks(sdy, B, H, W):
    """The fused MLP driven by INJECTED keep masks (sdy_mlp_args.keep_hidden / keep_out), drawn here by torch's own
    `nn.functional.dropout` the way the reference's layers draw them (src/models/sfno/layers.py:76-78) -- no Philox anywhere:
    the fused kernel's dropout plumbing (chain beside fc2, hidden tile in LDS, output dropout in the epilogue, 1 / (1 - p)
    folded into the accumulator scales) against fp64, on full and ragged tiles."""
    E, Hd, drop = 256, 512, 0.1
    g = _gen(23)
    F = torch.nn.functional
    x = torch.randn(B, E, H, W, generator=g) * 1.3 + 0.2
    w1 = torch.randn(Hd, E, 1, 1, generator=g) / np.sqrt(E)
    b1 = 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(E, Hd, 1, 1, generator=g) / np.sqrt(Hd)
    b2 = 0.1 * torch.randn(E, generator=g)
    res = torch.randn(B, E, H, W, generator=g)
    torch.manual_seed(5)
    k1 = (F.dropout(torch.ones(B, Hd, H, W), p=drop, training=True) != 0).float()
    k2 = (F.dropout(torch.ones(B, E, H, W), p=drop, training=True) != 0).float()
    assert 0.05 < 1 - float(k1.mean()) < 0.15
    hid = F.gelu(F.conv2d(x.double(), w1.double(), b1.double())) * k1.double() / (1.0 - drop)
    ref = F.conv2d(hid, w2.double(), b2.double()) * k2.double() / (1.0 - drop) + res.double()
    got = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, add=res.cuda(), drop_p=drop, keep_masks=(k1.cuda(), k2.cuda()))
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"fused MLP with injected masks vs fp64: {err:.3e}"
    # exactly the injected output mask: without the residual the zeros of the result are the zeros of k2
    f0 = sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, drop_p=drop, keep_masks=(k1.cuda(), k2.cuda())).cpu()
    assert torch.equal(f0 == 0, k2 == 0)
    # one mask without the other, or masks without dropout, is an argument error
    with pytest.raises(sdy.SdyError):
        sdy.ops.mlp_fused(x.cuda(), w1, b1, w2, b2, drop_p=0.0, keep_masks=(k1.cuda(), k2.cuda()))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 65, 256, 8, 40), (3, 36, 256, 87, 96), (2, 130, 256, 45, 64), (1, 144, 256, 6, 36),
                                            (2, 321, 63, 8, 40), (3, 292, 34, 87, 96), (2, 386, 63, 45, 64), (1, 416, 64, 6, 36),
                                            (2, 17, 5, 6, 36)])
def test_conv_pair_matches_fp64_and_two_launches(sdy, B, Cin, Cout, H, W):
    """sdy_pair_h3 (encoder: Cin -> 256 -> 256 + position embedding + InstanceNorm statistics; decoder: 256 + Cin -> 256 ->
    Cout) == conv -> GELU -> conv in fp64, and the two-launch path it replaces; ragged last tiles, several tiles per
    persistent workgroup (87 x 96), inputs in one and in two parts, every shape class of sdy_pair_h3_supported."""
    g = _gen(77)
    F = torch.nn.functional
    Hd = 256
    x = torch.randn(B, Cin, H, W, generator=g) * 1.3 + 0.2
    w1 = torch.randn(Hd, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b1 = 0.1 * torch.randn(Hd, generator=g)
    w2 = torch.randn(Cout, Hd, 1, 1, generator=g) / np.sqrt(Hd)
    pos = 0.5 * torch.randn(1, Cout, H, W, generator=g)
    hid = F.gelu(F.conv2d(x.double(), w1.double(), b1.double()))
    ref = F.conv2d(hid, w2.double())
    got = sdy.ops.conv_pair(x.cuda(), w1, b1, w2)
    assert got.shape == (B, Cout, H, W)
    err = rel_l2(got, ref)
    assert err < TOL_OP, f"fused pair vs fp64: {err:.3e}"
    h = sdy.ops.conv1x1(x.cuda(), w1, b1, gelu=True, h3=True)
    two = sdy.ops.conv1x1(h, w2, None, h3=True)
    assert rel_l2(got, two) < 5e-6
    # addend broadcast over the batch (position embedding) and per image; no bias
    got = sdy.ops.conv_pair(x.cuda(), w1, None, w2, add=pos.cuda())
    ref_nb = F.conv2d(F.gelu(F.conv2d(x.double(), w1.double())), w2.double())
    assert rel_l2(got, ref_nb + pos.double()) < TOL_OP
    res = torch.randn(B, Cout, H, W, generator=g)
    got = sdy.ops.conv_pair(x.cuda(), w1, b1, w2, add=res.cuda())
    assert rel_l2(got, ref + res.double()) < TOL_OP
    # a view of a wider buffer as input (the decoder reads [block output | inputs] in place) and as output
    wide = torch.randn(B, Cin + 3, H, W, generator=g).cuda()
    wide[:, :Cin] = x.cuda()
    out_wide = torch.zeros(B, Cout + 2, H, W, device="cuda")
    a = sdy._lib.SdyPairArgs()
    prep = sdy.ops.pack_pair_h3(w1, w2, "cuda")
    bb = b1.cuda()
    a.x, a.x_bstride = wide.data_ptr(), (Cin + 3) * H * W
    a.w, a.w1_scale, a.w2_scale = prep[0].data_ptr(), prep[1], prep[2]
    a.b1 = bb.data_ptr()
    a.out, a.out_bstride = out_wide.data_ptr(), (Cout + 2) * H * W
    a.B, a.Cin, a.hidden, a.Cout, a.HW = B, Cin, Hd, Cout, H * W
    import ctypes
    assert sdy.lib.sdy_pair_h3(ctypes.byref(a), sdy._lib.current_stream()) == 0
    assert rel_l2(out_wide[:, :Cout], ref) < TOL_OP and float(out_wide[:, Cout:].abs().max()) == 0.0
    if Cout == 256:   # statistics of the stored output from the epilogue
        st = torch.zeros(B, Cout, 2, dtype=torch.float64, device="cuda")
        got_s = sdy.ops.conv_pair(x.cuda(), w1, b1, w2, add=pos.cuda(), stats=st)
        gd = got_s.double().cpu()
        want = torch.stack([gd.sum((2, 3)), (gd * gd).sum((2, 3))], -1)
        assert torch.allclose(st.cpu(), want, rtol=1e-5, atol=1e-6 * H * W)
    else:
        with pytest.raises(sdy.SdyError):
            sdy.ops.conv_pair(x.cuda(), w1, b1, w2, stats=torch.zeros(B, Cout, 2, dtype=torch.float64, device="cuda"))


def test_conv_pair_rejects_other_shapes(sdy):
    for cin, hid, cout in [(145, 256, 256), (65, 128, 256), (417, 256, 63), (65, 256, 128)]:
        assert sdy.lib.sdy_pair_h3_supported(cin, hid, cout) == 0
        with pytest.raises(NotImplementedError):
            sdy.ops.pack_pair_h3(torch.zeros(hid, cin), torch.zeros(cout, hid), "cuda")
    for cin, hid, cout in [(65, 256, 256), (144, 256, 256), (292, 256, 34), (416, 256, 64), (1, 256, 1)]:
        assert sdy.lib.sdy_pair_h3_supported(cin, hid, cout) == 1


def test_mlp_fused_rejects_other_shapes(sdy):
    x = torch.zeros(1, 64, 8, 16).cuda()
    with pytest.raises(NotImplementedError):
        sdy.ops.mlp_fused(x, torch.zeros(128, 64), torch.zeros(128), torch.zeros(64, 128), torch.zeros(64))


@pytest.mark.parametrize("B,H,W", [(2, 20, 36), (1, 45, 64), (3, 6, 36), (3, 87, 96)])
def test_conv256_persistent_kernel_and_statistics(sdy, B, H, W):
    """(the last shape has 393 tiles, ragged image edges included: several tiles per persistent workgroup)
    conv_h3_kernel (256 -> 256, persistent, weight as MFMA fragment stream) == the reference conv with the block's
    inner-skip epilogue GELU(conv + bias + add), the encoder's post-add, and the InstanceNorm statistics of its output."""
    g = _gen(31)
    F = torch.nn.functional
    E = 256
    x = torch.randn(B, E, H, W, generator=g) * 1.4 + 0.1
    w = torch.randn(E, E, 1, 1, generator=g) / np.sqrt(E)
    b = 0.1 * torch.randn(E, generator=g)
    add = torch.randn(B, E, H, W, generator=g)
    pa, pd = 1 + 0.2 * torch.randn(B, E, generator=g), 0.2 * torch.randn(B, E, generator=g)
    frag = sdy.ops.pack_conv256(w, "cuda")
    st = torch.zeros(B, E, 2, dtype=torch.float64, device="cuda")
    out = sdy.ops.conv1x1(x.cuda(), w, b, pre_affine=(pa.cuda(), pd.cuda()), add=add.cuda(), add_mode=1, gelu=True,
                          frag_prepared=frag, stats=st)
    xd = x.double() * pa.double()[:, :, None, None] + pd.double()[:, :, None, None]
    ref = F.gelu(F.conv2d(xd, w.double(), b.double()) + add.double())
    assert rel_l2(out, ref) < TOL_OP
    od = out.double().cpu()
    want = torch.stack([od.sum((2, 3)), (od * od).sum((2, 3))], -1)
    assert torch.allclose(st.cpu(), want, rtol=1e-5, atol=1e-6 * H * W)
    gamma, beta = 1 + 0.1 * torch.randn(E, generator=g), 0.1 * torch.randn(E, generator=g)
    a_s, d_s = sdy.ops.instnorm_from_stats(st, H * W, gamma, beta)
    a_r, d_r = sdy.ops.instnorm_coeffs(out, gamma, beta)
    assert rel_l2(a_s, a_r) < 1e-5 and rel_l2(d_s, d_r) < 1e-5
    # encoder form: no bias, broadcast post-add, no activation
    pe = torch.randn(1, E, H, W, generator=g)
    out2 = sdy.ops.conv1x1(x.cuda(), w, None, add=pe.cuda(), add_mode=2, frag_prepared=frag)
    assert rel_l2(out2, F.conv2d(x.double(), w.double()) + pe.double()) < TOL_OP
    # same result as the tile GEMM
    out3 = sdy.ops.conv1x1(x.cuda(), w, None, add=pe.cuda(), add_mode=2, h3=True)
    assert rel_l2(out2, out3) < 5e-6


@pytest.mark.parametrize("Cin", [65, 128, 130, 200, 4, 321, 384, 260])
def test_conv_cin_to_256_persistent_kernel(sdy, Cin):
    """The persistent kernel with other than 256 input channels (the encoders' first layers: 65 and 128 channels in the
    benchmark configuration; the decoder's first layer reads [block output | inputs]: 321 and 384 channels, the 8-wave
    variant with a 96 KB tile): bias + GELU epilogue, no addend, ragged image edge (3 x 20 x 36 = 12 tiles of 64 +
    ragged)."""
    g = _gen(47)
    F = torch.nn.functional
    B, H, W = 3, 20, 37 * 4
    x = torch.randn(B, Cin, H, W, generator=g) * 1.3 - 0.2
    w = torch.randn(256, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    b = 0.1 * torch.randn(256, generator=g)
    frag = sdy.ops.pack_conv256(w, "cuda")
    out = sdy.ops.conv1x1(x.cuda(), w, b, gelu=True, frag_prepared=frag)
    ref = F.gelu(F.conv2d(x.double(), w.double(), b.double()))
    assert rel_l2(out, ref) < TOL_OP
    st = torch.zeros(B, 256, 2, dtype=torch.float64, device="cuda")
    out2 = sdy.ops.conv1x1(x.cuda(), w, None, frag_prepared=frag, stats=st)
    ref2 = F.conv2d(x.double(), w.double())
    assert rel_l2(out2, ref2) < TOL_OP
    want = torch.stack([ref2.sum((2, 3)), (ref2 * ref2).sum((2, 3))], -1)
    assert torch.allclose(st.cpu(), want, rtol=1e-4, atol=1e-5 * H * W)


@pytest.mark.parametrize("grid", ["equiangular", "legendre-gauss"])
@pytest.mark.parametrize("nlat,nlon,L,M,C,B", [
    (180, 360, 180, 181, 16, 2),    # nlon = 360 FFT specialisation (C % 16 == 0) + folded Legendre kernel
    (180, 360, 120, 100, 32, 1),    # truncated lmax / mmax
    (90, 360, 90, 181, 16, 1),      # fewer rings, mmax > lmax
    (33, 64, 33, 33, 4, 2),         # odd nlat: the unfolded Legendre kernel, generic FFT
    (32, 64, 20, 17, 12, 2),        # truncated small grid
])
def test_sht_shapes_both_directions(sdy, nlat, nlon, L, M, C, B, grid):
    """Shape coverage of the kernel selection inside RealSHT / InverseRealSHT (split-fp16 mode): nlon = 360 with 16-channel
    blocks takes fft360.hip, even nlat takes leg_par.hip (equatorial symmetry folded), everything else the generic kernels."""
    from oracle.sht import InverseRealSHT as OInv, RealSHT as OFwd

    x = torch.randn(B, C, nlat, nlon, generator=_gen(61))
    ref = OFwd(nlat, nlon, lmax=L, mmax=M, grid=grid).float()(x)
    got = sdy.RealSHT(nlat, nlon, lmax=L, mmax=M, grid=grid, gemm_mode="h3").float()(x.cuda())
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < TOL_OP, f"RealSHT {nlat}x{nlon} L={L} M={M} {grid}"
    c = torch.randn(B, C, L, M, dtype=torch.complex64, generator=_gen(62))
    refi = OInv(nlat, nlon, lmax=L, mmax=M, grid=grid).float()(c)
    goti = sdy.InverseRealSHT(nlat, nlon, lmax=L, mmax=M, grid=grid, gemm_mode="h3").float()(c.cuda())
    assert rel_l2(goti, refi) < TOL_OP, f"InverseRealSHT {nlat}x{nlon} L={L} M={M} {grid}"


def test_fp16_range_guard_sets_the_sticky_flag(sdy):
    """The split-precision kernels scale activations by 16 into fp16: |x| >= 4094 overflows to inf.  That must not be silent:
    the kernel sets SDY_FLAG_F16_RANGE in the device's sticky status word; the fp32-MFMA mode has no such limit."""
    ops = sdy.ops
    g = _gen(21)
    F = torch.nn.functional
    ops.status_flags(reset=True)
    x = torch.randn(1, 256, 32, 64, generator=g)
    w = torch.randn(256, 256, 1, 1, generator=g) / 16.0
    got = ops.conv1x1(x.cuda(), w, None, h3=True)                       # conv_h3 (256 -> 256 fragment-stream kernel)
    assert ops.status_flags(reset=True) == 0 and torch.isfinite(got).all()
    big = x.clone()
    big[0, 17, 3, 5] = 1.0e4                                              # 1e4 * 16 > 65504
    got = ops.conv1x1(big.cuda(), w, None, h3=True)
    assert ops.status_flags(reset=False) & ops.FLAG_F16_RANGE
    assert not torch.isfinite(got).all()                                  # ... which is what the flag warns about
    assert ops.status_flags(reset=True) & ops.FLAG_F16_RANGE              # sticky until reset
    assert ops.status_flags(reset=True) == 0
    ref = F.conv2d(big.double(), w.double())
    got32 = ops.conv1x1(big.cuda(), w, None)                              # fp32-MFMA path: exact range
    assert ops.status_flags(reset=True) == 0 and rel_l2(got32, ref) < TOL_OP
    # tile GEMM (other channel counts: encoder / decoder layers)
    xs = torch.randn(1, 36, 32, 64, generator=g)
    ws = torch.randn(130, 36, 1, 1, generator=g) / 6.0
    ops.conv1x1(xs.cuda(), ws, None, h3=True)
    assert ops.status_flags(reset=True) == 0
    xs[0, 5, 1, 1] = -7.0e3
    ops.conv1x1(xs.cuda(), ws, None, h3=True)
    assert ops.status_flags(reset=True) & ops.FLAG_F16_RANGE
    # just inside the range is fine: 4000 * 16 = 64000 < 65504
    xs[0, 5, 1, 1] = 4000.0
    got = ops.conv1x1(xs.cuda(), ws, None, h3=True)
    assert ops.status_flags(reset=True) == 0
    assert rel_l2(got, F.conv2d(xs.double(), ws.double())) < 5e-6


def test_legendre_inputs_are_guarded_at_their_producers(sdy):
    """The folded Legendre kernel (leg_par) stages (x[k] +- x[mirror]) * 16 as fp16 and has no register left for a range guard, so
    its inputs are guarded where they are PRODUCED.  Analysis: rfft360 -- a field that is harmless everywhere else (|x| = 400:
    x 16 = 6400) has the zonal-mean coefficient Xf[m = 0] = 2 pi x 400 = 2513, and 2 x 16 x 2513 passes 65504: the flag must
    come from the FFT's stores, not one InstanceNorm later as "non-finite".  Synthesis: dh_h3's stores -- small coefficients,
    large filter weights.  The fp32 mode has no limit and raises nothing."""
    ops = sdy.ops
    nlat, nlon = 180, 360
    ops.status_flags(reset=True)
    x = torch.zeros(1, 16, nlat, nlon)
    x[:, 3] = 300.0                                  # 2 pi 300 x 32 = 60.3e3: inside
    sht = sdy.RealSHT(nlat, nlon, grid="legendre-gauss", gemm_mode="h3")
    c = sht(x.cuda())
    assert ops.status_flags(reset=True) == 0 and torch.isfinite(torch.view_as_real(c)).all()
    x[:, 3] = 400.0                                  # 2 pi 400 x 32 = 80.4e3: outside the (conservative) bound
    sht(x.cuda())
    assert ops.status_flags(reset=True) & ops.FLAG_F16_RANGE
    sdy.RealSHT(nlat, nlon, grid="legendre-gauss", gemm_mode="f32")(x.cuda())
    assert ops.status_flags(reset=True) == 0
    # dh_h3: |x| ~ 1, weights ~ 300: outputs of ~ 300 sqrt(512) = 6800 > 4094 = 65504 / 16
    g = _gen(8)
    L, M = 20, 11
    xc = torch.randn(2, 256, L, M, dtype=torch.complex64, generator=g) * (torch.arange(M)[None, :] <= torch.arange(L)[:, None])
    w = torch.randn(256, 256, L, 2, generator=g)
    got = ops.contract_dhconv(xc.cuda(), (0.02 * w).cuda(), gemm_mode="h3")
    assert ops.status_flags(reset=True) == 0
    got = ops.contract_dhconv(xc.cuda(), (300.0 * w).cuda(), gemm_mode="h3")
    assert float(torch.view_as_real(got).abs().max()) > 4094.0 and torch.isfinite(torch.view_as_real(got)).all()
    assert ops.status_flags(reset=True) & ops.FLAG_F16_RANGE          # the coefficients themselves are fine: their CONSUMER is not


def test_range_headroom_reports_the_distance_to_the_cliff(sdy):
    """`sdy_range_headroom` (ops.range_headroom): the largest magnitude each consumer class stages as fp16, pre-scale included,
    while the debug read-back is on -- "x N below the cliff" instead of pass / fail.  Exact for a single convolution (max |x| = 3
    -> 16 x 3 = 48); all five classes report for a forward of a production-width network, none of them near 65504 for a
    trained-like network on standardised inputs; nothing is recorded while the switch is off."""
    ops = sdy.ops
    g = _gen(9)
    x = torch.randn(1, 256, 32, 64, generator=g).clamp(-2.5, 2.5)
    x[0, 200, 31, 63] = -3.0
    w = torch.randn(256, 256, 1, 1, generator=g) / 16.0
    frag = ops.pack_conv256(w, torch.device("cuda"))            # the persistent kernel conv_h3 (the tile GEMMs are not tracked)
    with ops.range_headroom() as h:
        ops.conv1x1(x.cuda(), w, None, frag_prepared=frag)
    assert h.max_staged["conv_h3 x tile"] == 48.0 and abs(h.factor["conv_h3 x tile"] - 65504.0 / 48.0) < 1e-9
    assert h.max_staged["mlp_h3 x tile"] == 0.0 and h.factor["mlp_h3 x tile"] == float("inf")
    with ops.range_headroom() as h2:
        pass
    ops.conv1x1(x.cuda(), w, None, frag_prepared=frag)           # switch off: no bookkeeping
    with ops.range_headroom() as h3:
        pass
    assert all(v == 0.0 for v in h2.max_staged.values()) and all(v == 0.0 for v in h3.max_staged.values())
    from sdy_amd import synthetic
    # (three blocks: the inner skip of the first and the last one is folded into their dhconv weights, the middle one runs conv_h3)
    net = synthetic.build_network(8, 6, 2, nlat=180, nlon=360, embed=256, layers=3, dropout_mlp=0.1, drop_path_rate=0.1,
                                  time_range=(0.0, 5.0))
    xin = torch.randn(2, 8, 180, 360, generator=g).cuda()
    cond = torch.randn(2, 2, 180, 360, generator=g).cuda()
    ops.status_flags(reset=True)
    with ops.range_headroom() as hn:
        net(xin, time=torch.tensor([1.0, 4.0]).cuda(), condition=cond)
    assert ops.status_flags(reset=True) == 0
    for name, v in hn.max_staged.items():
        assert 1.0 < v < 65504.0 / 4.0, f"{name}: staged maximum {v} (factor {hn.factor[name]:.1f})"


def test_decoder_pair_flags_nonfinite_inputs_and_takes_any_magnitude(sdy):
    """The decoder is the only consumer of the last block's output (no InstanceNorm in between, whose statistics flag
    non-finite tensors everywhere else) and a max-based guard ignores NaNs: sdy_pair_h3's decoder shapes set
    SDY_FLAG_NONFINITE for a NaN / inf input.  Since round 4 the pair kernels stage their x tile with a scale of its own
    maximum (and the hidden tile with one from a bound of it), so an out-of-range input is no longer an error: a 5e3 / 1e6
    outlier gives the fp64 result and no flag."""
    ops = sdy.ops
    g = _gen(5)
    F = torch.nn.functional

    def ref(x, w1, b1, w2):
        return F.conv2d(F.gelu(F.conv2d(x.double(), w1.double()[:, :, None, None], b1.double())), w2.double()[:, :, None, None])

    x = torch.randn(1, 321, 8, 40, generator=g)
    w1 = torch.randn(256, 321, generator=g) / 18.0
    b1 = 0.1 * torch.randn(256, generator=g)
    w2 = torch.randn(63, 256, generator=g) / 16.0
    ops.status_flags(reset=True)
    assert torch.isfinite(ops.conv_pair(x.cuda(), w1, b1, w2)).all() and ops.status_flags(reset=True) == 0
    for bad in (float("nan"), float("inf")):
        xb = x.clone()
        xb[0, 300, 7, 39] = bad
        out = ops.conv_pair(xb.cuda(), w1, b1, w2)
        fl = ops.status_flags(reset=True)
        assert fl & ops.FLAG_NONFINITE and not torch.isfinite(out).all()
    for big in (5.0e3, 1.0e6):
        xb = x.clone()
        xb[0, 3, 0, 0] = big                # second input part of the decoder shape: channel 300
        xb[0, 300, 5, 17] = -big
        out = ops.conv_pair(xb.cuda(), w1, b1, w2)
        assert ops.status_flags(reset=True) == 0
        assert rel_l2(out, ref(xb, w1, b1, w2)) < TOL_OP, big
    xe = torch.randn(1, 65, 8, 40, generator=g)
    we1, we2 = torch.randn(256, 65, generator=g) / 8.0, torch.randn(256, 256, generator=g) / 16.0
    xe[0, 64, 2, 2] = -5.0e3
    out = ops.conv_pair(xe.cuda(), we1, b1, we2)
    assert ops.status_flags(reset=True) == 0
    assert rel_l2(out, ref(xe, we1, b1, we2)) < TOL_OP


@pytest.mark.parametrize("scale", [1.0e-4, 1.0, 1.0e4, 1.0e7])
@pytest.mark.parametrize("Cin,Cout", [(65, 256), (130, 256), (321, 63), (386, 34)])
def test_conv_pair_dynamic_scale_holds_precision_at_any_input_magnitude(sdy, scale, Cin, Cout):
    """Inputs scaled by 1e-4 ... 1e7 (the bias scaled along, so the GELU sees both its linear and its curved range):
    sdy_pair_h3 against fp64 at the per-operator tolerance, no status flag -- the fixed x16 pre-scale this replaces
    overflowed at 4094 and lost the `lo` parts below 0.01."""
    ops = sdy.ops
    g = _gen(31)
    F = torch.nn.functional
    B, H, W = 2, 9, 40
    x = torch.randn(B, Cin, H, W, generator=g) * scale
    w1 = torch.randn(256, Cin, generator=g) / np.sqrt(Cin)
    b1 = 0.3 * scale * torch.randn(256, generator=g)
    w2 = torch.randn(Cout, 256, generator=g) / 16.0
    ops.status_flags(reset=True)
    out = ops.conv_pair(x.cuda(), w1, b1, w2)
    assert ops.status_flags(reset=True) == 0
    ref = F.conv2d(F.gelu(F.conv2d(x.double(), w1.double()[:, :, None, None], b1.double())), w2.double()[:, :, None, None])
    err = rel_l2(out, ref)
    assert err < TOL_OP, f"scale {scale}: rel L2 {err:.3e}"
