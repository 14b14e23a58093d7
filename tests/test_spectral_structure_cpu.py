"""CPU checks of the two structural facts the MI355X Legendre / FFT kernels exploit (leg_par.hip, fft360.hip, DESIGN.md
section 3), on the oracle's own tables (torch_harmonics conventions):

* equatorial symmetry: P_l^m(-x) = (-1)^(l+m) P_l^m(x) and symmetric quadrature weights, so the transforms can be folded
  onto one hemisphere;
* polar decay: for each zonal order the rows near the poles carry negligible table entries, so those (order, latitude)
  pairs can be skipped.
"""
import numpy as np
import pytest

from oracle.sht import sht_tables

GRIDS = ["legendre-gauss", "equiangular"]


@pytest.mark.parametrize("grid", GRIDS)
def test_legendre_tables_are_equatorially_symmetric(grid):
    nlat, nlon = 180, 360
    wq, pct, *_ = sht_tables(nlat, nlon, nlat, nlon // 2 + 1, grid)   # [m][l][k] analysis / synthesis tables
    wq, pct = np.asarray(wq, dtype=np.float64)[:180], np.asarray(pct, dtype=np.float64)[:180]
    m = np.arange(180)[:, None, None]
    l = np.arange(180)[None, :, None]
    sign = np.where((l + m) % 2 == 0, 1.0, -1.0)
    for T in (wq, pct):
        mirror = T[:, :, ::-1]
        assert np.abs(T - sign * mirror).max() <= 2e-6 * np.abs(T).max()   # tables are fp32: symmetric to rounding
    # folded analysis / synthesis == full products
    rng = np.random.default_rng(0)
    x = rng.standard_normal((180, nlat))                       # one column per order: x[m][k]
    full = np.einsum("mlk,mk->ml", wq, x)
    xe, xo = x[:, :90] + x[:, :89:-1], x[:, :90] - x[:, :89:-1]
    even = ((np.arange(180)[None, :] + np.arange(180)[:, None]) % 2 == 0)   # [m][l]
    fold = np.where(even, np.einsum("mlk,mk->ml", wq[:, :, :90], xe), np.einsum("mlk,mk->ml", wq[:, :, :90], xo))
    assert np.abs(fold - full).max() <= 1e-5 * np.abs(full).max()
    c = rng.standard_normal((180, 180))                        # c[m][l]
    c = c * (np.arange(180)[None, :] >= np.arange(180)[:, None])
    fulls = np.einsum("mlk,ml->mk", pct, c)
    E = np.einsum("mlk,ml->mk", pct[:, :, :90], c * even)
    O = np.einsum("mlk,ml->mk", pct[:, :, :90], c * ~even)
    folds = np.concatenate([E + O, (E - O)[:, ::-1]], axis=1)
    assert np.abs(folds - fulls).max() <= 1e-5 * np.abs(fulls).max()


@pytest.mark.parametrize("grid", GRIDS)
def test_polar_cutoff_is_negligible(grid):
    nlat, nlon, eps = 180, 360, 1e-12
    wq, pct, *_ = sht_tables(nlat, nlon, nlat, nlon // 2 + 1, grid)
    wq, pct = np.abs(np.asarray(wq, dtype=np.float64)[:180]), np.abs(np.asarray(pct, dtype=np.float64)[:180])
    dead = ((wq.max(axis=1) < eps * wq.max(axis=(1, 2), keepdims=True)[:, 0])
            & (pct.max(axis=1) < eps * pct.max(axis=(1, 2), keepdims=True)[:, 0]))          # [m][k]
    frac = dead.mean()
    assert 0.15 < frac < 0.35, f"dead (order, latitude) fraction {frac:.3f}"
    # the dead rows of an order form a polar cap that grows with the order (what the per-ring order cut-off relies on)
    kdead = np.array([np.argmin(dead[m, :90]) if not dead[m, :90].all() else 90 for m in range(180)])
    assert (dead[np.arange(180)[:, None], np.arange(90)[None, :]] == (np.arange(90)[None, :] < kdead[:, None])).all()
    assert (np.diff(kdead) >= 0).all() and kdead[0] == 0 and kdead[-1] > 50
    # worst-case contribution of everything that is skipped, relative to a typical coefficient
    rng = np.random.default_rng(1)
    x = rng.standard_normal((180, nlat))
    skipped = np.einsum("mlk,mk->ml", wq * dead[:, None, :], np.abs(x))
    kept = np.abs(np.einsum("mlk,mk->ml", wq, x))
    assert skipped.max() < 1e-9 * kept.max()
