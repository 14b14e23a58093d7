"""On-device ensemble diagnostics: host mirror of the reductions in `src/ace_inference/core/metrics.py`.

`ensemble_metrics(truth, predicted, weights)` returns, per (sample, time) plane, what the reference computes with
`root_mean_squared_error(truth, predicted.mean(0), weights, dim=(-2, -1))` (`metrics.py:107-132`),
`ensemble_spread(predicted, weights, dim=(-2, -1))` (`:135-144`), `spread_skill_ratio` (`:146-155`),
`weighted_crps(truth, predicted, weights, dim=(-2, -1))` (`:158-208`, fair form) and `weighted_mean_bias` (`:84-104`) of
the ensemble mean -- in ONE pass over the ensemble on the device (`sdy_ensemble_metrics`): the reference materialises the
(E, E, ...) pairwise-difference tensor for the CRPS.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from ._lib import check, current_stream, lib, ptr


def spherical_area_weights(lats, num_lon: int) -> torch.Tensor:
    """`metrics.py:14-29`: cos(latitude) weights of a regular lat-lon grid, normalised to sum 1; shape (num_lat, num_lon)."""
    lats = torch.as_tensor(lats)
    w = torch.cos(torch.deg2rad(lats)).repeat(num_lon, 1).t()
    return w / w.sum()


def ensemble_metrics(truth: torch.Tensor, predicted: torch.Tensor, weights: torch.Tensor) -> Dict[str, torch.Tensor]:
    """truth (..., H, W), predicted (E, ..., H, W), weights (H, W) -> dict of (...) fp64 tensors:
    rmse (of the ensemble mean), spread (with the (E+1)/E correction), spread_skill_ratio, crps (fair), bias."""
    if not predicted.is_cuda:
        raise RuntimeError("sdy_amd metrics run on the GPU only (no CPU fallback)")
    assert predicted.shape[1:] == truth.shape, f"truth {tuple(truth.shape)} vs predicted {tuple(predicted.shape)}"
    dev = predicted.device
    p = predicted.to(torch.float32).contiguous()
    t = truth.to(dev, torch.float32).contiguous()
    w = weights.to(dev, torch.float32).contiguous()
    E = p.shape[0]
    H, W = p.shape[-2:]
    assert tuple(w.shape) == (H, W)
    lead = tuple(t.shape[:-2])
    n = 1
    for s in lead:
        n *= s
    out = torch.zeros(max(n, 1), 4, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.sdy_ensemble_metrics(ptr(p), ptr(t), ptr(w), E, max(n, 1) * H * W, max(n, 1), H * W, ptr(out),
                                       current_stream()), "sdy_ensemble_metrics")
    out = out / w.double().sum()
    rmse = out[:, 0].sqrt().reshape(lead)
    spread = out[:, 1].sqrt().reshape(lead) * ((E + 1) / E) ** 0.5
    return {"rmse": rmse, "spread": spread, "spread_skill_ratio": spread / rmse, "crps": out[:, 2].reshape(lead),
            "bias": out[:, 3].reshape(lead)}
