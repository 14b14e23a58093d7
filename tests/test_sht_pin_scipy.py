"""Production-size (180 x 360) pins of the SHT tables against implementations that share no code with `oracle/sht.py`
or `csrc/tables.cpp` (SURVEY.md Appendix A.4 lists such pins at small n only; the reference holds no fixture at this
boundary because `torch_harmonics` is third-party and un-vendored):

* every (m, l, k) of `pct` on both grids vs `scipy.special.sph_harm_y` (scipy's own stable recursion; orthonormal,
  Condon-Shortley phase - the conventions torch-harmonics documents for norm="ortho", csphase=True);
* Clenshaw-Curtis weights vs a moment solve in the Chebyshev basis at n = 180, Gauss-Legendre nodes/weights vs
  `scipy.special.roots_legendre`;
* `pct * w` (the analysis table) vs scipy's values times those independent weights;
* exact properties at n = 180: full-band orthonormality on the Gauss grid, the addition theorem
  sum_m |Y_lm|^2 = (2l+1)/4pi at every latitude of both grids.
The same checks are applied to the table the product uploads (`sdy_sht_tables_host`, csrc/tables.cpp).
"""
import ctypes as C

import numpy as np
import pytest
from scipy import special

from oracle.sht import quadrature, sht_tables

NLAT, NLON = 180, 360
L, M = NLAT, NLON // 2 + 1
GRIDS = {"equiangular": 0, "legendre-gauss": 1}


def independent_quadrature(grid):
    """(colatitudes ascending, weights) without touching oracle/sht.py."""
    if grid == "legendre-gauss":
        x, w = special.roots_legendre(NLAT)                 # ascending x = cos(theta) -> theta descending
        return np.arccos(x)[::-1].copy(), w[::-1].copy()
    theta = np.linspace(0.0, np.pi, NLAT)                   # both poles included
    k = np.arange(NLAT)[:, None]
    A = np.cos(k * theta[None, :])                          # T_k(cos theta_j)
    b = np.zeros(NLAT)
    b[0::2] = 2.0 / (1.0 - np.arange(0, NLAT, 2) ** 2)       # int_{-1}^{1} T_k = 2 / (1 - k^2), k even; 0, k odd
    return theta, np.linalg.solve(A, b)


@pytest.fixture(scope="module", params=list(GRIDS))
def scipy_tables(request):
    grid = request.param
    theta, w = independent_quadrature(grid)
    m = np.arange(M)[:, None, None]
    l = np.arange(L)[None, :, None]
    Y = special.sph_harm_y(l, m, theta[None, None, :], 0.0)
    assert np.abs(Y.imag).max() == 0.0
    P = np.where(m <= l, Y.real, 0.0)                      # scipy returns NaN/0 for m > l depending on version
    return grid, theta, w, P


def test_quadrature_vs_independent(scipy_tables):
    grid, theta, w, _ = scipy_tables
    th_o, w_o = quadrature(NLAT, grid)
    assert np.abs(th_o - theta).max() < 1e-13
    assert np.abs(w_o - w).max() < 2e-13, np.abs(w_o - w).max()
    assert abs(w.sum() - 2.0) < 1e-13


def test_oracle_tables_vs_scipy_every_entry(scipy_tables):
    grid, theta, w, P = scipy_tables
    pct, wq, l_, m_ = sht_tables(NLAT, NLON, L, M, grid)
    assert (l_, m_) == (L, M) and pct.shape == P.shape == (M, L, NLAT)
    assert np.abs(pct - P).max() < 2e-12, np.abs(pct - P).max()               # measured 4e-13 (values up to 5.3)
    assert np.abs(wq - P * w[None, None, :]).max() < 1e-13
    # what the networks see: the fp32 casts agree except where the fp64 values straddle a rounding boundary
    a, b = pct.astype(np.float32), P.astype(np.float32)
    assert (a != b).mean() < 1e-4 and np.abs(a - b).max() <= np.spacing(np.float32(5.4))
    assert (pct[np.arange(M)[:, None] > np.arange(L)[None, :]] == 0).all()      # m > l and the m = 180 column


def test_exact_properties_at_180(scipy_tables):
    grid, theta, w, _ = scipy_tables
    pct, wq, _, _ = sht_tables(NLAT, NLON, L, M, grid)
    # addition theorem on every latitude: sum_{m=-l}^{l} |Y_lm|^2 = (2l+1)/(4 pi)
    s = pct[0] ** 2 + 2.0 * (pct[1:L] ** 2).sum(axis=0)                       # (L, K)
    want = (2 * np.arange(L) + 1) / (4 * np.pi)
    assert np.abs(s / want[:, None] - 1.0).max() < 1e-11
    if grid == "legendre-gauss":      # Gauss quadrature with 180 nodes is exact for degree <= 359: full band
        for m in (0, 1, 2, 45, 90, 178, 179):
            G = 2 * np.pi * wq[m, m:] @ pct[m, m:].T
            assert np.abs(G - np.eye(L - m)).max() < 2e-12, (m, np.abs(G - np.eye(L - m)).max())
    else:                             # Clenshaw-Curtis with 180 nodes is exact for degree <= 179: l + l' <= 179
        for m in (0, 1, 2, 45, 88):
            hi = (L - 1) // 2 + 1
            G = 2 * np.pi * wq[m, m:hi] @ pct[m, m:hi].T
            assert np.abs(G - np.eye(hi - m)).max() < 2e-12, m


def test_product_host_tables_vs_scipy(scipy_tables):
    """csrc/tables.cpp (what sdy_sht_plan_create uploads) against scipy directly, not through the oracle."""
    import sdy_amd

    grid, theta, w, P = scipy_tables
    pct = np.zeros((M, L, NLAT))
    wv, th = np.zeros(NLAT), np.zeros(NLAT)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert sdy_amd.lib.sdy_sht_tables_host(NLAT, NLON, L, M, GRIDS[grid], vp(pct), vp(wv), vp(th)) == 0
    assert np.abs(th - theta).max() < 1e-13 and np.abs(wv - w).max() < 2e-13
    assert np.abs(pct - P).max() < 2e-12, np.abs(pct - P).max()
    assert (pct.astype(np.float32) != P.astype(np.float32)).mean() < 1e-4
    assert np.abs((pct * wv).astype(np.float32) - (P * w).astype(np.float32)).max() < 1e-8


@pytest.mark.parametrize("grid", list(GRIDS))
def test_transform_known_answers_at_180x360(grid):
    """The transforms themselves (rfft scaling, 2*pi factor, complex conventions) at production size: a field built from
    scipy spherical harmonics has the known coefficients, and synthesis of single coefficients reproduces scipy's fields."""
    import torch

    from oracle.sht import InverseRealSHT, RealSHT

    theta, _ = independent_quadrature(grid)
    phi = 2 * np.pi * np.arange(NLON) / NLON
    modes = [(0, 0, 1.5, 0.0), (5, 2, 0.5, -0.25), (40, 40, -1.0, 2.0), (89, 17, 0.75, 0.5), (60, 0, 2.0, 0.0)]
    if grid == "legendre-gauss":
        modes += [(179, 179, 1.0, -1.0), (179, 1, -0.5, 0.25), (150, 97, 0.3, 0.6)]
    # real field f = sum c Y_lm + conj (m > 0): f = sum 2 Re(c Y_lm);  m = 0: c real, f = c Y_l0
    f = np.zeros((NLAT, NLON))
    want = np.zeros((L, M), dtype=np.complex128)
    for l, m, cr, ci in modes:
        Y = special.sph_harm_y(l, m, theta[:, None], phi[None, :])
        c = complex(cr, ci)
        f += (c * Y).real * (2.0 if m > 0 else 1.0)
        want[l, m] = c
    got = RealSHT(NLAT, NLON, grid=grid)(torch.from_numpy(f)).numpy()
    if grid == "legendre-gauss":
        assert np.abs(got - want).max() < 5e-12, np.abs(got - want).max()
    else:   # Clenshaw-Curtis: exact where l + l' <= 179; compare there
        lmax_in = max(l for l, *_ in modes)
        ok = np.arange(L) <= (NLAT - 1) - lmax_in
        assert np.abs(got[ok] - want[ok]).max() < 5e-12, np.abs(got[ok] - want[ok]).max()
    back = InverseRealSHT(NLAT, NLON, grid=grid)(torch.from_numpy(want)).numpy()
    assert np.abs(back - f).max() < 5e-12 * max(1.0, np.abs(f).max())
