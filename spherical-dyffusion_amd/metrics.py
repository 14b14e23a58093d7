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


class TorchDistributed:
    """`reduce_mean` of the reference's `Distributed` singleton (`src/ace_inference/core/distributed.py`): mean over ranks
    with `torch.distributed.all_reduce` (RCCL over xGMI on the GPU box; identity without a process group).  The reduce is
    issued on a side stream: it belongs to `get_logs`, not to the sampling path, and never blocks the compute stream."""

    def __init__(self):
        import torch.distributed as dist

        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self._stream = None

    @property
    def world_size(self) -> int:
        return self._dist.get_world_size() if self._dist is not None else 1

    def reduce_mean(self, tensor: torch.Tensor) -> torch.Tensor:
        if self._dist is None:
            return tensor
        if not tensor.is_cuda:
            out = tensor.clone()
            self._dist.all_reduce(out)
            return out / self.world_size
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=tensor.device)
        self._stream.wait_stream(torch.cuda.current_stream(tensor.device))
        with torch.cuda.stream(self._stream):
            out = tensor.clone()
            self._dist.all_reduce(out)
            out /= self.world_size
        torch.cuda.current_stream(tensor.device).wait_stream(self._stream)
        return out


class TimeMeanAggregator:
    """Statistics on the time-mean state: host mirror of `TimeMeanAggregator`
    (`src/ace_inference/core/aggregator/inference/time_mean.py:45-173`) minus its plots.

    Same constructor keywords that matter (`area_weights`, `dist`, `target`, `is_ensemble`), same `record_batch(loss,
    target_data, gen_data, target_data_norm, gen_data_norm, i_time_start)` (what `run_inference` calls once per window) and
    the same numbers from `get_logs(label)`: `rmse/<name>`, `bias/<name>`, `rmse/channel_mean` of the time-mean maps.  The
    maps stay on the device (`time_mean_maps()`), one HIP launch per variable and window adds a window's mean over members,
    samples and time (`sdy_time_mean_accumulate`, strided views welcome); `get_logs` reduces the maps over ranks
    (`dist.reduce_mean`) and takes RMSE / bias with `sdy_ensemble_metrics`.  The matplotlib / wandb images of the reference
    are out of scope."""

    def __init__(self, area_weights: torch.Tensor, dist=None, target: str = "denorm", metadata=None,
                 log_individual_channels: bool = True, is_ensemble: bool = False):
        if target not in ("norm", "denorm"):
            raise ValueError(f"target must be 'norm' or 'denorm', got {target!r}")
        self._area_weights = area_weights
        self._is_ensemble = is_ensemble
        self._target = target
        self._log_individual_channels = log_individual_channels
        self._dist = TorchDistributed() if dist is None else dist
        self._target_data: Dict[str, torch.Tensor] = {}
        self._gen_data: Dict[str, torch.Tensor] = {}
        self._n_batches = 0

    @staticmethod
    def _accumulate(maps: Dict[str, torch.Tensor], data, t0: int, ensemble: bool) -> None:
        for name, v in data.items():
            if not v.is_cuda:
                raise RuntimeError("sdy_amd aggregators run on the GPU only (no CPU fallback)")
            v = v.to(torch.float32)
            if ensemble:
                assert v.dim() == 5, "ensemble data are (members, samples, time, lat, lon)"
                n0, n1, T, H, W = v.shape
            else:
                assert v.dim() == 4, "data are (samples, time, lat, lon)"
                (n1, T, H, W), n0 = v.shape, 1
            if v.stride(-1) != 1 or v.stride(-2) != W or v.stride(-3) != H * W:
                v = v.contiguous()
            s0, s1 = (v.stride(0), v.stride(1)) if ensemble else (0, v.stride(0))
            if name not in maps:
                maps[name] = torch.zeros(H, W, dtype=torch.float32, device=v.device)
            with torch.cuda.device(v.device):
                check(lib.sdy_time_mean_accumulate(ptr(v), n0, s0, n1, s1, t0, T, H * W, 1.0 / (n0 * n1 * (T - t0)),
                                                   ptr(maps[name]), current_stream()), "sdy_time_mean_accumulate")

    @torch.no_grad()
    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start: int = 0):
        if self._target == "norm":
            target_data, gen_data = target_data_norm, gen_data_norm
        t0 = 1 if i_time_start == 0 else 0          # the very first time of a run is the initial condition
        self._accumulate(self._target_data, target_data, t0, ensemble=False)
        self._accumulate(self._gen_data, gen_data, t0, ensemble=self._is_ensemble)
        self._n_batches += 1

    def time_mean_maps(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """{"gen": {name: (H, W)}, "target": {...}}: time means so far, reduced over ranks, on the device."""
        if self._n_batches == 0:
            raise ValueError("No data recorded.")
        red = lambda d: {k: self._dist.reduce_mean(v / self._n_batches) for k, v in d.items()}  # noqa: E731
        return {"gen": red(self._gen_data), "target": red(self._target_data)}

    @torch.no_grad()
    def get_logs(self, label: str) -> Dict[str, float]:
        maps = self.time_mean_maps()
        logs, rmse_all = {}, {}
        for name, gen in maps["gen"].items():
            m = ensemble_metrics(maps["target"][name], gen[None], self._area_weights)     # one "member": RMSE and bias of a map
            rmse_all[name] = float(m["rmse"])
            if self._log_individual_channels:
                logs[f"rmse/{name}"] = rmse_all[name]
                logs[f"bias/{name}"] = float(m["bias"])
        logs["rmse/channel_mean"] = sum(rmse_all.values()) / len(rmse_all)
        return {f"{label}/{k}": v for k, v in logs.items()} if len(label) != 0 else logs
