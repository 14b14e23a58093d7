"""Oracle: restatement of the reference's ensemble diagnostics (CPU, torch, fp64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `src/ace_inference/core/metrics.py`: `weighted_mean` (:32-54), `root_mean_squared_error` (:107-132),
`ensemble_spread` (:135-144), `spread_skill_ratio` (:146-155), `weighted_crps` (:158-208, fair form),
`weighted_mean_bias` (:84-104), `weighted_std` (:57-82).  Pinned by `tests/golden/fx_metrics.npz` (produced by the
reference's own functions); the aggregators by `fx_time_mean.npz` / `fx_mean_series.npz` (the reference's own classes).
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


def time_mean_maps(windows, is_ensemble: bool):
    """Restatement of `TimeMeanAggregator.record_batch` + `_get_target_gen_pairs`
    (`src/ace_inference/core/aggregator/inference/time_mean.py:97-160`) for one process: `windows` = [(i_time_start,
    target {name: (S, T, H, W)}, gen {name: (S, T, H, W) or (E, S, T, H, W)})].  Returns ({name: gen map}, {name: target
    map}).  Pinned by `tests/golden/fx_time_mean.npz` (produced by the reference's own aggregator)."""
    tgt_acc, gen_acc, n = {}, {}, 0
    for i_time_start, tgt, gen in windows:
        sl = slice(1, None) if i_time_start == 0 else slice(0, None)
        if is_ensemble:
            gen = {k: v.mean(dim=0) for k, v in gen.items()}
        for acc, d in ((tgt_acc, tgt), (gen_acc, gen)):
            for k, v in d.items():
                m = v[:, sl].mean(dim=1).mean(dim=0)
                acc[k] = acc[k] + m if k in acc else m
        n += 1
    return {k: v / n for k, v in gen_acc.items()}, {k: v / n for k, v in tgt_acc.items()}


def mean_series(windows, is_ensemble: bool, n_timesteps: int):
    """Restatement of `MeanAggregator.record_batch` + `AreaWeightedReducedMetric` + `_get_series_data`
    (`src/ace_inference/core/aggregator/inference/reduced.py:105-250`) for one process, without the gradient-magnitude metric:
    `windows` = [(i_time_start, target {name: (S, T, H, W)}, gen {name: (S, T, H, W) or (E, S, T, H, W)}, weights (H, W))].
    Returns {"<metric>/<name>": (n_timesteps,) fp64} (NaN where no window touched a time index, as the reference's 0 / 0).
    Pinned by `tests/golden/fx_mean_series.npz` (produced by the reference's own aggregator)."""
    total, count = {}, torch.zeros(n_timesteps, dtype=torch.float64)
    for i_time_start, tgt, gen, w in windows:
        w = w.double()
        nt = None
        for name, g in gen.items():
            t, g = tgt[name].double(), g.double()
            mean = g.mean(dim=0) if is_ensemble else g
            vals = {"weighted_rmse": weighted_mean((mean - t) ** 2, w).sqrt(), "weighted_bias": weighted_mean(mean - t, w),
                    "weighted_mean_gen": weighted_mean(mean, w), "weighted_mean_target": weighted_mean(t, w)}
            for key, x in (("weighted_std_gen", mean), ("weighted_std_target", t)):
                m = weighted_mean(x, w)[..., None, None]
                vals[key] = weighted_mean((x - m) ** 2, w).sqrt()
            if is_ensemble:
                em = ensemble_metrics(t, g, w)
                vals["weighted_crps"], vals["weighted_ssr"] = em["crps"], em["spread_skill_ratio"]
            nt = t.shape[1]
            for metric, v in vals.items():
                acc = total.setdefault(f"{metric}/{name}", torch.zeros(n_timesteps, dtype=torch.float64))
                acc[i_time_start:i_time_start + nt] += v.mean(dim=0)
        count[i_time_start:i_time_start + nt] += 1
    return {k: v / count for k, v in total.items()}
