"""Oracle: restatement of torch_harmonics.RealSHT / InverseRealSHT (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Third-party dependency being restated: PyPI `torch-harmonics`, un-pinned in the
reference (`setup.py:98`, `environment/install_dependencies.sh:13`); absent from
/root/reference and from this image.  Reference call sites that fix the API:
`src/models/sfno/sfnonet.py:551-554` (positional nlat, nlon; kwargs lmax, mmax,
grid; then `.float()`), `src/models/sfno/s2convolutions.py:73-83,165-186`
(attributes nlat/nlon/lmax/mmax/grid; forward on (..., nlat, nlon) real ->
(..., lmax, mmax) complex64 and back).

Published algorithm (torch-harmonics 0.6.x, `norm="ortho"`, `csphase=True`):
  nodes/weights  : legendre-gauss = numpy leggauss; equiangular = Clenshaw-Curtis
                   including both poles (Waldvogel's FFT construction)
  table          : fully-normalised associated Legendre functions by the
                   standard 3-term recursion in fp64, Condon-Shortley phase
  forward        : X = 2*pi*rfft(x, norm="forward"); out[l,m] = sum_k X[k,m] P[m,l,k] w_k
  inverse        : Y[k,m] = sum_l c[l,m] P[m,l,k]; x = irfft(Y, n=nlon, norm="forward")

PARITY UNPINNED by the reference (it holds no test or fixture at this boundary).
Pinned instead against independent implementations at PRODUCTION size
(tests/test_sht_pin_scipy.py: every (m, l, k) of both 180 x 360 tables vs
scipy.special.sph_harm_y, quadrature vs a Chebyshev moment solve /
scipy.special.roots_legendre, transform known answers from scipy fields,
addition theorem, full-band Gauss orthonormality) and analytically at small
sizes (tests/test_oracle_sht.py).
"""
from __future__ import annotations

import numpy as np
import torch


def legendre_gauss_weights(n: int):
    x, w = np.polynomial.legendre.leggauss(n)
    return x, w


def clenshaw_curtiss_weights(n: int):
    """Clenshaw-Curtis nodes (both poles included) and weights on [-1, 1]."""
    assert n > 1
    tcc = np.cos(np.linspace(np.pi, 0, n))
    if n == 2:
        wcc = np.array([1.0, 1.0])
    else:
        n1 = n - 1
        N = np.arange(1, n1, 2)
        ln = len(N)
        m = n1 - ln
        v = np.concatenate([2 / N / (N - 2), 1 / N[-1:], np.zeros(m)])
        v = 0 - v[:-1] - v[-1:0:-1]
        g0 = -np.ones(n1)
        g0[ln] = g0[ln] + n1
        g0[m] = g0[m] + n1
        g = g0 / (n1**2 - 1 + (n1 % 2))
        wcc = np.fft.ifft(v + g).real
        wcc = np.concatenate((wcc, wcc[:1]))
    return tcc, wcc


def quadrature(nlat: int, grid: str):
    """Return (colatitudes ascending 0..pi, weights) for the grid."""
    if grid == "legendre-gauss":
        cost, w = legendre_gauss_weights(nlat)
    elif grid == "equiangular":
        cost, w = clenshaw_curtiss_weights(nlat)
    else:
        raise ValueError(f"unsupported grid {grid!r}")
    theta = np.flip(np.arccos(cost)).copy()
    return theta, w


def legpoly(mmax: int, lmax: int, x: np.ndarray, csphase: bool = True) -> np.ndarray:
    """Ortho-normalised associated Legendre table P[m, l, k] in fp64."""
    nmax = max(mmax, lmax)
    vdm = np.zeros((nmax, nmax, len(x)), dtype=np.float64)
    vdm[0, 0, :] = 1.0 / np.sqrt(4 * np.pi)
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * x * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = np.sqrt((2 * l + 1) * (1 + x) * (1 - x) / 2 / l) * vdm[l - 1, l - 1, :]
    for l in range(2, nmax):
        for m in range(0, l - 1):
            vdm[m, l, :] = (
                x * np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m)) * vdm[m, l - 1, :]
                - np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m))
                * vdm[m, l - 2, :]
            )
    vdm = vdm[:mmax, :lmax]
    if csphase:
        for m in range(1, mmax, 2):
            vdm[m] *= -1
    return vdm


def sht_tables(nlat: int, nlon: int, lmax: int | None, mmax: int | None, grid: str):
    """fp64 tables: (pct[m,l,k], weights[m,l,k] = pct*w_k, lmax, mmax)."""
    lmax = lmax or nlat
    mmax = mmax or nlon // 2 + 1
    theta, w = quadrature(nlat, grid)
    pct = legpoly(mmax, lmax, np.cos(theta))
    weights = pct * w[None, None, :]
    return pct, weights, lmax, mmax


class RealSHT(torch.nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular"):
        super().__init__()
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        _, weights, self.lmax, self.mmax = sht_tables(nlat, nlon, lmax, mmax, grid)
        self.register_buffer("weights", torch.from_numpy(weights), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.nlat and x.shape[-1] == self.nlon
        x = 2.0 * torch.pi * torch.fft.rfft(x, dim=-1, norm="forward")
        x = torch.view_as_real(x)
        out_shape = list(x.size())
        out_shape[-3] = self.lmax
        out_shape[-2] = self.mmax
        xout = torch.zeros(out_shape, dtype=x.dtype, device=x.device)
        w = self.weights.to(x.dtype)
        xout[..., 0] = torch.einsum("...km,mlk->...lm", x[..., : self.mmax, 0], w)
        xout[..., 1] = torch.einsum("...km,mlk->...lm", x[..., : self.mmax, 1], w)
        return torch.view_as_complex(xout)


class InverseRealSHT(torch.nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular"):
        super().__init__()
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        pct, _, self.lmax, self.mmax = sht_tables(nlat, nlon, lmax, mmax, grid)
        self.register_buffer("pct", torch.from_numpy(pct), persistent=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-2] == self.lmax and x.shape[-1] == self.mmax
        x = torch.view_as_real(x)
        p = self.pct.to(x.dtype)
        rl = torch.einsum("...lm,mlk->...km", x[..., 0], p)
        im = torch.einsum("...lm,mlk->...km", x[..., 1], p)
        x = torch.view_as_complex(torch.stack((rl, im), -1))
        return torch.fft.irfft(x, n=self.nlon, dim=-1, norm="forward")
