"""Analytic pins of the SHT restatement (the reference holds no test or fixture at this boundary: 'parity unpinned';
these known-answer tests are what pins it instead -- SURVEY.md Appendix A.4)."""
import numpy as np
import pytest
import torch

from oracle.sht import InverseRealSHT, RealSHT, clenshaw_curtiss_weights, legendre_gauss_weights, sht_tables


@pytest.mark.parametrize("n", [8, 33, 180])
def test_clenshaw_curtis_moments(n):
    x, w = clenshaw_curtiss_weights(n)
    assert abs(w.sum() - 2.0) < 1e-14
    assert abs((w * x**2).sum() - 2.0 / 3.0) < 1e-14
    if n > 11:
        assert abs((w * x**10).sum() - 2.0 / 11.0) < 1e-13
    assert np.allclose(w, w[::-1], atol=1e-15) and x[0] == -1.0 and x[-1] == 1.0


def test_gauss_legendre_exactness():
    x, w = legendre_gauss_weights(16)
    for p in range(0, 31, 2):
        assert abs((w * x**p).sum() - 2.0 / (p + 1)) < 1e-13


@pytest.mark.parametrize("grid,full_band", [("legendre-gauss", True), ("equiangular", False)])
def test_orthonormality(grid, full_band):
    nlat, nlon = 64, 128
    pct, wq, L, M = sht_tables(nlat, nlon, nlat, nlon // 2 + 1, grid)
    lim = L if full_band else L // 2
    for m in (0, 1, 2, 7, 30):
        G = 2 * np.pi * np.einsum("lk,jk->lj", wq[m, m:lim], pct[m, m:lim])
        assert np.abs(G - np.eye(lim - m)).max() < 5e-13, (grid, m)
    # structural zeros
    l = np.arange(L)[None, :, None]
    mm = np.arange(M)[:, None, None]
    assert (pct[(mm > l).repeat(nlat, axis=2)] == 0).all()


def test_equiangular_full_band_is_not_exact():
    """Clenshaw-Curtis at lmax = nlat is not a full-band exact quadrature; the tables must be replicated, not 'fixed'."""
    pct, wq, L, M = sht_tables(32, 64, 32, 33, "equiangular")
    G = 2 * np.pi * np.einsum("lk,jk->lj", wq[0], pct[0])
    assert np.abs(G - np.eye(L)).max() > 1e-3


def test_known_answer_scipy_sph_harm():
    """RealSHT of Re(Y_5^2) is a single coefficient 0.5 at (l=5, m=2): phase/normalisation convention = scipy's."""
    from scipy.special import sph_harm

    nlat, nlon = 32, 64
    from oracle.sht import quadrature

    theta, _ = quadrature(nlat, "legendre-gauss")
    phi = 2 * np.pi * np.arange(nlon) / nlon
    Y = sph_harm(2, 5, phi[None, :], theta[:, None])   # scipy: (m, n, azimuth, polar)
    x = torch.from_numpy(Y.real.copy())
    c = RealSHT(nlat, nlon, grid="legendre-gauss")(x)
    ref = torch.zeros_like(c)
    ref[5, 2] = 0.5
    assert (c - ref).abs().max() < 1e-12


@pytest.mark.parametrize("grid", ["legendre-gauss", "equiangular"])
def test_roundtrip_bandlimited(grid):
    nlat, nlon = 32, 64
    L = nlat if grid == "legendre-gauss" else nlat // 2
    sht = RealSHT(nlat, nlon, lmax=L, mmax=L, grid=grid)
    isht = InverseRealSHT(nlat, nlon, lmax=L, mmax=L, grid=grid)
    g = torch.Generator().manual_seed(0)
    c = torch.randn(3, L, L, dtype=torch.complex128, generator=g)
    l = torch.arange(L)[:, None]
    m = torch.arange(L)[None, :]
    c = c * (m <= l)
    c[..., 0] = c[..., 0].real + 0j
    c2 = sht(isht(c))
    assert (c2 - c).abs().max() < 1e-11


def test_irfft_ignores_imag_of_dc():
    """Complex dhconv weights make the m=0 coefficients complex; irfft must drop that imaginary part."""
    isht = InverseRealSHT(16, 32, grid="legendre-gauss")
    c = torch.randn(16, 17, dtype=torch.complex128)
    c2 = c.clone()
    c2[:, 0] = c2[:, 0].real + 0j
    assert torch.equal(isht(c), isht(c2))
