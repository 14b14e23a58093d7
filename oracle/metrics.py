"""Oracle: restatement of the reference's ensemble diagnostics (CPU, torch, fp64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `src/ace_inference/core/metrics.py`: `weighted_mean` (:32-54), `root_mean_squared_error` (:107-132),
`ensemble_spread` (:135-144), `spread_skill_ratio` (:146-155), `weighted_crps` (:158-208, fair form),
`weighted_mean_bias` (:84-104).  Pinned by `tests/golden/fx_metrics.npz` (produced by the reference's own functions).
"""
import torch


def weighted_mean(x, w, dim=(-2, -1)):
    return (x * w).sum(dim=dim) / w.expand(x.shape).sum(dim=dim)


def ensemble_metrics(truth: torch.Tensor, predicted: torch.Tensor, weights: torch.Tensor):
    truth, predicted, weights = truth.double(), predicted.double(), weights.double()
    E = predicted.shape[0]
    mean = predicted.mean(0)
    rmse = weighted_mean((mean - truth) ** 2, weights).sqrt()
    spread = weighted_mean(predicted.var(dim=0), weights).sqrt() * ((E + 1) / E) ** 0.5
    skill = (predicted - truth).abs().mean(0)
    diff = (predicted.unsqueeze(0) - predicted.unsqueeze(1)).abs().sum(dim=(0, 1)) / (E * (E - 1))
    crps = weighted_mean(skill - 0.5 * diff, weights)
    bias = weighted_mean(mean - truth, weights)
    return {"rmse": rmse, "spread": spread, "spread_skill_ratio": spread / rmse, "crps": crps, "bias": bias}
