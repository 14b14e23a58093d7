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
from typing import Dict, List, Mapping, Optional, Sequence

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
    """`reduce_mean` / `reduce_sum` of the reference's `Distributed` singleton (`src/ace_inference/core/distributed.py:70-94`):
    `torch.distributed.all_reduce` over ranks (RCCL over xGMI on the GPU box; identity without a process group).  The reduce
    is issued on a side stream: it belongs to `get_logs`, not to the sampling path, and never blocks the compute stream."""

    def __init__(self):
        import torch.distributed as dist

        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self._stream = None

    @property
    def world_size(self) -> int:
        return self._dist.get_world_size() if self._dist is not None else 1

    def reduce_sum(self, tensor: torch.Tensor) -> torch.Tensor:
        if self._dist is None:
            return tensor
        if not tensor.is_cuda:
            out = tensor.clone()
            self._dist.all_reduce(out)
            return out
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=tensor.device)
        self._stream.wait_stream(torch.cuda.current_stream(tensor.device))
        with torch.cuda.stream(self._stream):
            out = tensor.clone()
            self._dist.all_reduce(out)
        torch.cuda.current_stream(tensor.device).wait_stream(self._stream)
        return out

    def reduce_mean(self, tensor: torch.Tensor) -> torch.Tensor:
        if self._dist is None:
            return tensor
        return self.reduce_sum(tensor) / self.world_size


class TimeMeanAggregator:
    """Statistics on the time-mean state: host mirror of `TimeMeanAggregator`
    (`src/ace_inference/core/aggregator/inference/time_mean.py:45-173`) minus its plots.

    Same constructor keywords that matter (`area_weights`, `dist`, `target`, `is_ensemble`), same `record_batch(loss,
    target_data, gen_data, target_data_norm, gen_data_norm, i_time_start)` (what `run_inference` calls once per window) and
    the same numbers from `get_logs(label)`: `rmse/<name>`, `bias/<name>`, `rmse/channel_mean` of the time-mean maps.  The
    maps stay on the device (`time_mean_maps()`), one HIP launch per variable and window adds a window's time means
    (`sdy_time_mean_accumulate`, strided views welcome); `get_logs` combines ranks and takes RMSE / bias with
    `sdy_ensemble_metrics`.  The matplotlib / wandb images of the reference are out of scope.

    Ranks.  The reference shards whole initial conditions over ranks and averages the ranks' maps with equal weight
    (`time_mean.py:147-148`, `Distributed.reduce_mean`).  Here a rank's share is any contiguous range of trajectories
    (`ensemble.shard`: 25 members over 8 GPUs are 4, 3, 3, ... rows), so the maps are kept as SUMS over rows of per-row time
    means next to the row count, both are summed over ranks (`dist.reduce_sum`) and divided at the end: every trajectory
    weighs the same whatever the sharding, and one process gets exactly the reference's numbers.  Generated data may be the
    member-stacked `(members, samples, time, lat, lon)` or, for a ragged share, flat `(rows, time, lat, lon)`; for a ragged
    share `run_inference` also passes `sample_weights` (the fraction of each touched initial condition's members that ran on
    this rank), which weigh the target maps the same way."""

    accepts_sample_weights = True

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
        self._target_rows = 0.0       # sum over windows of the (weighted) sample count behind the target maps
        self._gen_rows = 0.0          # ... of the trajectory count behind the generated maps
        self._n_batches = 0

    @staticmethod
    def _accumulate(maps: Dict[str, torch.Tensor], data, t0: int, ensemble: bool,
                    sample_weights: Optional[Sequence[float]] = None) -> float:
        """maps[name] += sum over rows of the row's mean over times t0..T-1 (row weights optional); returns the row count."""
        rows = 0.0
        for name, v in data.items():
            if not v.is_cuda:
                raise RuntimeError("sdy_amd aggregators run on the GPU only (no CPU fallback)")
            v = v.to(torch.float32)
            if ensemble and v.dim() == 5:
                n0, n1, T, H, W = v.shape
            else:
                assert v.dim() == 4, "data are (samples, time, lat, lon) [or (members, samples, time, lat, lon)]"
                (n1, T, H, W), n0 = v.shape, 1
            if v.stride(-1) != 1 or v.stride(-2) != W or v.stride(-3) != H * W:
                v = v.contiguous()
            s0, s1 = (v.stride(0), v.stride(1)) if v.dim() == 5 else (0, v.stride(0))
            if name not in maps:
                maps[name] = torch.zeros(H, W, dtype=torch.float32, device=v.device)
            with torch.cuda.device(v.device):
                if sample_weights is None:
                    check(lib.sdy_time_mean_accumulate(ptr(v), n0, s0, n1, s1, t0, T, H * W, 1.0 / (T - t0), ptr(maps[name]),
                                                       current_stream()), "sdy_time_mean_accumulate")
                    rows = float(n0 * n1)
                else:
                    assert n0 == 1 and len(sample_weights) == n1, "one weight per sample"
                    for j, wj in enumerate(sample_weights):
                        check(lib.sdy_time_mean_accumulate(ptr(v[j]), 1, 0, 1, 0, t0, T, H * W, float(wj) / (T - t0),
                                                           ptr(maps[name]), current_stream()), "sdy_time_mean_accumulate")
                    rows = float(sum(sample_weights))
        return rows

    @torch.no_grad()
    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start: int = 0,
                     sample_weights: Optional[Sequence[float]] = None):
        if self._target == "norm":
            target_data, gen_data = target_data_norm, gen_data_norm
        t0 = 1 if i_time_start == 0 else 0          # the very first time of a run is the initial condition
        self._target_rows += self._accumulate(self._target_data, target_data, t0, ensemble=False,
                                              sample_weights=sample_weights)
        self._gen_rows += self._accumulate(self._gen_data, gen_data, t0, ensemble=self._is_ensemble)
        self._n_batches += 1

    def time_mean_maps(self) -> Dict[str, Dict[str, torch.Tensor]]:
        """{"gen": {name: (H, W)}, "target": {...}}: time means so far over every rank's trajectories, on the device."""
        if self._n_batches == 0:
            # (raised BEFORE any collective: every rank of a job must have recorded at least one window -- a rank without a
            # share would leave the others waiting in reduce_sum; shard with ensemble.partition, which gives every rank of
            # world <= trajectories a non-empty share)
            raise ValueError("No data recorded.")

        def red(d, rows):
            if not d:
                return {}
            dev = next(iter(d.values())).device
            n = float(self._dist.reduce_sum(torch.tensor([rows], dtype=torch.float64, device=dev))[0])
            return {k: self._dist.reduce_sum(v) / n for k, v in d.items()}

        return {"gen": red(self._gen_data, self._gen_rows), "target": red(self._target_data, self._target_rows)}

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


SERIES_METRICS = ("weighted_rmse", "weighted_bias", "weighted_mean_gen", "weighted_mean_target", "weighted_std_gen",
                  "weighted_std_target")
SERIES_METRICS_ENSEMBLE = ("weighted_crps", "weighted_ssr")


def ensemble_series(truth: torch.Tensor, predicted: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """truth (n_sample, T, H, W), predicted (members, n_sample, T, H, W) -- any strides on the two leading axes, so the window
    driver's member-stacked VIEW is read in place -- weights (H, W)  ->  (n_sample, T, 8) fp64: the area-weighted means of
    (ens. mean - truth)^2 | member variance (unbiased) | fair CRPS | ens. mean - truth | ens. mean | (ens. mean)^2 | truth |
    truth^2 per (sample, time) plane (`sdy_ensemble_series`: one pass, members in registers)."""
    if not predicted.is_cuda:
        raise RuntimeError("sdy_amd metrics run on the GPU only (no CPU fallback)")
    assert predicted.dim() == 5 and predicted.shape[1:] == truth.shape, \
        f"truth {tuple(truth.shape)} vs predicted {tuple(predicted.shape)}"
    dev = predicted.device
    M, n_sample, T, H, W = predicted.shape
    p = predicted.to(torch.float32)
    if p.stride(-1) != 1 or p.stride(-2) != W or p.stride(-3) != H * W:
        p = p.contiguous()
    t = truth.to(dev, torch.float32)
    if t.stride(-1) != 1 or t.stride(-2) != W or t.stride(-3) != H * W:
        t = t.contiguous()
    w = weights.to(dev, torch.float32).contiguous()
    out = torch.zeros(n_sample, T, 8, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.sdy_ensemble_series(ptr(p), M, p.stride(0), p.stride(1), ptr(t), t.stride(0), ptr(w), n_sample, T, H * W,
                                      ptr(out), current_stream()), "sdy_ensemble_series")
    return out / w.double().sum()


class MeanAggregator:
    """Per-timestep series of area-weighted metrics: host mirror of `MeanAggregator` / `AreaWeightedReducedMetric`
    (`src/ace_inference/core/aggregator/inference/reduced.py:105-266`) minus the wandb table / xarray packaging.

    Same constructor keywords that matter (`area_weights`, `target`, `n_timesteps`, `is_ensemble`, `dist`), same
    `record_batch(loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start)`; `get_series()` returns
    `{"<metric>/<variable>": (n_timesteps,) fp64 tensor}` -- the arrays the reference puts into its table: mean over the
    windows' samples, accumulated at `i_time_start ...` and divided by the number of windows that touched a time index, then
    averaged over ranks (`dist.reduce_mean`, `reduced.py:247`).  Metrics: `weighted_rmse`, `weighted_bias` (of the ensemble
    mean), `weighted_mean_gen / _target`, `weighted_std_gen / _target`, and for ensembles `weighted_crps` (fair) and
    `weighted_ssr`; the reference's `weighted_grad_mag_percent_diff` is not computed (out of scope).
    One `sdy_ensemble_series` launch per variable and window reads the member-stacked view in place; the accumulators are
    (n_timesteps,) fp64 tensors on the device.

    Ensemble metrics need every member of an initial condition on one rank (the reference's IC sharding): a ragged share
    (`run_inference(unit_range=...)` cutting through an IC's members) hands over flat rows and is refused."""

    def __init__(self, area_weights: torch.Tensor, target: str = "denorm", n_timesteps: int = 1, is_ensemble: bool = False,
                 dist=None, device=None, metadata=None):
        if target not in ("norm", "denorm"):
            raise ValueError(f"target must be 'norm' or 'denorm', got {target!r}")
        self._area_weights = area_weights
        self._target = target
        self._n_timesteps = int(n_timesteps)
        self.is_ensemble = is_ensemble
        self._dist = TorchDistributed() if dist is None else dist
        self._total: Dict[str, Dict[str, torch.Tensor]] = {}      # metric -> variable -> (n_timesteps,) fp64
        self._n_batches: Optional[torch.Tensor] = None            # (n_timesteps,) int32, as AreaWeightedReducedMetric

    @property
    def metric_names(self) -> List[str]:
        return list(SERIES_METRICS + (SERIES_METRICS_ENSEMBLE if self.is_ensemble else ()))

    @torch.no_grad()
    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start: int = 0):
        if self._target == "norm":
            target_data, gen_data = target_data_norm, gen_data_norm
        n_time = None
        for name, gen in gen_data.items():
            if self.is_ensemble:
                if gen.dim() != 5:
                    raise ValueError("MeanAggregator(is_ensemble=True) needs member-stacked (members, samples, time, lat, lon) "
                                     "predictions: ensemble-mean RMSE / CRPS / spread of an initial condition need all of its "
                                     "members on one rank (shard whole initial conditions, or use TimeMeanAggregator)")
                pred = gen
            else:
                pred = gen[None]
            s = ensemble_series(target_data[name], pred, self._area_weights)       # (n_sample, T, 8)
            E = pred.shape[0]
            mse, var, crps, bias, mg, mg2, mt, mt2 = s.unbind(dim=-1)
            rmse = mse.sqrt()
            vals = {"weighted_rmse": rmse, "weighted_bias": bias, "weighted_mean_gen": mg, "weighted_mean_target": mt,
                    "weighted_std_gen": (mg2 - mg * mg).clamp_min(0.0).sqrt(),
                    "weighted_std_target": (mt2 - mt * mt).clamp_min(0.0).sqrt()}
            if self.is_ensemble:
                vals["weighted_crps"] = crps
                vals["weighted_ssr"] = var.sqrt() * ((E + 1) / E) ** 0.5 / rmse
            n_time = s.shape[1]
            sl = slice(i_time_start, i_time_start + n_time)
            for metric, v in vals.items():
                tot = self._total.setdefault(metric, {})
                if name not in tot:
                    tot[name] = torch.zeros(self._n_timesteps, dtype=torch.float64, device=s.device)
                tot[name][sl] += v.mean(dim=0)                                        # mean over the window's samples
            if self._n_batches is None:
                self._n_batches = torch.zeros(self._n_timesteps, dtype=torch.int32, device=s.device)
        if n_time is not None:
            self._n_batches[i_time_start:i_time_start + n_time] += 1

    @torch.no_grad()
    def get_series(self) -> Dict[str, torch.Tensor]:
        if not self._total:
            raise ValueError("No batches have been recorded.")
        return {f"{metric}/{name}": self._dist.reduce_mean(tot / self._n_batches)
                for metric, per_var in self._total.items() for name, tot in per_var.items()}

    @torch.no_grad()
    def get_logs(self, label: str):
        """`reduced.py:252-266` puts the series into one wandb table under `<label>/series`; here: the arrays themselves."""
        return {f"{label}/series": {k: v.cpu().numpy() for k, v in self.get_series().items()}}


# ---- the reference's composite (`src/ace_inference/core/aggregator/inference/main.py`) ----------------------------------
class Table:
    """The two members of `wandb.Table` the reference's log plumbing uses (`columns`, `data`, `add_data`): wandb itself is not
    a dependency of this package."""

    def __init__(self, columns: Sequence[str]):
        self.columns = list(columns)
        self.data: List[list] = []

    def add_data(self, *row):
        if len(row) != len(self.columns):
            raise ValueError(f"expected {len(self.columns)} values, got {len(row)}")
        self.data.append(list(row))


def data_to_table(data: Mapping[str, Sequence[float]]) -> Table:
    """`reduced.py:282-293`: 1-D series -> one table with a `forecast_step` column and the keys in sorted order."""
    keys = sorted(data.keys())
    table = Table(["forecast_step"] + keys)
    for i in range(len(data[keys[0]])):
        table.add_data(i, *[data[k][i] for k in keys])
    return table


def to_inference_logs(log: Mapping[str, object]) -> List[Dict[str, float]]:
    """`main.py:189-211`: a dict holding tables and scalars -> one dict per table row (the wandb step is the forecast step),
    columns renamed `<key without its last component>/<column>`; scalars go into the last row's dict."""
    n_rows = max([len(v.data) for v in log.values() if isinstance(v, Table)], default=0)
    logs: List[Dict[str, float]] = [{} for _ in range(n_rows)]
    for key, val in log.items():
        if isinstance(val, Table):
            stem = key[: key.rfind("/")]
            for i, row in enumerate(val.data):
                for j, col in enumerate(val.columns):
                    logs[i][f"{stem}/{col}"] = row[j]
        else:
            logs[-1][key] = val
    return logs


class OneStepMeanAggregator:
    """Metrics of ONE forecast step averaged over the windows that contain it (`one_step/reduced.py:35-147`, the reference's
    `mean_step_20`): `weighted_rmse`, `weighted_bias` (of the ensemble mean), `weighted_mean_gen`, for ensembles
    `weighted_crps` and `weighted_ssr`, and the mean of the `loss` values handed to `record_batch`.  The reference's
    `weighted_grad_mag_percent_diff` is not computed (as in `MeanAggregator`)."""

    def __init__(self, area_weights: torch.Tensor, target_time: int = 1, is_ensemble: bool = False, dist=None, device=None):
        self._area_weights = area_weights
        self._target_time = int(target_time)
        self.is_ensemble = is_ensemble
        self._dist = TorchDistributed() if dist is None else dist
        self._loss = 0.0
        self._n_batches = 0
        self._total: Dict[str, Dict[str, torch.Tensor]] = {}

    @torch.no_grad()
    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start: int = 0):
        self._loss = self._loss + loss        # (added for every window, as the reference does: reduced.py:103)
        t = self._target_time - i_time_start
        any_gen = next(iter(gen_data.values()))
        if t < 0 or t >= any_gen.shape[2 if self.is_ensemble else 1]:
            return
        for name, gen in gen_data.items():
            pred = gen if self.is_ensemble else gen[None]
            s = ensemble_series(target_data[name][:, t:t + 1], pred[:, :, t:t + 1], self._area_weights)[:, 0]    # (n_sample, 8)
            E = pred.shape[0]
            mse, var, crps, bias, mg = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4]
            vals = {"weighted_rmse": mse.sqrt(), "weighted_bias": bias, "weighted_mean_gen": mg}
            if self.is_ensemble:
                vals["weighted_crps"] = crps
                vals["weighted_ssr"] = var.sqrt() * ((E + 1) / E) ** 0.5 / mse.sqrt()
            for metric, v in vals.items():
                per_var = self._total.setdefault(metric, {})
                per_var[name] = per_var.get(name, 0.0) + v.mean()
        self._n_batches += 1

    @torch.no_grad()
    def get_logs(self, label: str) -> Dict[str, float]:
        if self._n_batches == 0:
            raise ValueError("No batches have been recorded.")
        dev = self._area_weights.device
        logs = {f"{label}/loss": torch.as_tensor(self._loss, dtype=torch.float64, device=dev) / self._n_batches}
        for metric, per_var in self._total.items():
            for name, tot in per_var.items():
                logs[f"{label}/{metric}/{name}"] = tot.double() / self._n_batches
        return {k: float(self._dist.reduce_mean(logs[k].reshape(1))[0]) for k in sorted(logs)}


class InferenceAggregator:
    """The aggregator `run_inference` is handed by the reference's entry point (`inference/inference.py:247-262`): `mean`
    (per-step series, denormalised), `mean_norm` (normalised), `time_mean`, and `mean_step_20` when asked for, behind ONE
    `record_batch` / `get_logs` / `get_inference_logs` (`aggregator/inference/main.py:42-168`).  Same constructor keywords;
    `sigma_coordinates` and `metadata` are accepted and unused (they feed derived variables and image captions).  The
    image products of the reference -- snapshots, videos, zonal-mean hovmollers, the time-mean maps as pictures -- are out of
    scope: `log_video` / `log_zonal_mean_images` raise, snapshots are not produced;
    `get_time_mean_maps()` returns what `get_datasets(["time_mean"])` would hold, as device tensors."""

    accepts_sample_weights = True

    def __init__(self, area_weights: torch.Tensor, sigma_coordinates=None, n_timesteps: Optional[int] = None,
                 n_ensemble_members: int = 1, record_step_20: bool = False, log_video: bool = False,
                 enable_extended_videos: bool = False, log_zonal_mean_images: bool = False, dist=None, metadata=None,
                 device=None):
        if log_video or enable_extended_videos or log_zonal_mean_images:
            raise NotImplementedError("video / zonal-mean image logging is out of scope of sdy_amd (DESIGN.md section 8)")
        if n_timesteps is None:
            raise ValueError("n_timesteps (forward steps + 1) is needed for the per-step series")
        self._is_ensemble = n_ensemble_members > 1
        if device is not None:
            area_weights = area_weights.to(device)
        kw = dict(area_weights=area_weights, dist=dist, is_ensemble=self._is_ensemble)
        self._aggregators = {
            "mean": MeanAggregator(target="denorm", n_timesteps=n_timesteps, **kw),
            "mean_norm": MeanAggregator(target="norm", n_timesteps=n_timesteps, **kw),
            "time_mean": TimeMeanAggregator(area_weights, dist=dist, is_ensemble=self._is_ensemble),
        }
        if record_step_20:
            self._aggregators["mean_step_20"] = OneStepMeanAggregator(target_time=20, **kw)

    @torch.no_grad()
    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start: int = 0,
                     sample_weights: Optional[Sequence[float]] = None):
        if len(target_data) == 0:
            raise ValueError("No data in target_data")
        if len(gen_data) == 0:
            raise ValueError("No data in gen_data")
        for agg in self._aggregators.values():
            kw = {"sample_weights": sample_weights} if getattr(agg, "accepts_sample_weights", False) else {}
            agg.record_batch(loss=loss, target_data=target_data, gen_data=gen_data, target_data_norm=target_data_norm,
                             gen_data_norm=gen_data_norm, i_time_start=i_time_start, **kw)

    @torch.no_grad()
    def get_logs(self, label: str) -> Dict[str, object]:
        logs: Dict[str, object] = {}
        for name, agg in self._aggregators.items():
            for key, val in agg.get_logs(label=name).items():
                logs[key] = data_to_table(val) if isinstance(val, Mapping) else val     # a series -> the reference's table
        return {f"{label}/{key}": val for key, val in logs.items()}

    @torch.no_grad()
    def get_inference_logs(self, label: str) -> List[Dict[str, float]]:
        return to_inference_logs(self.get_logs(label=label))

    def get_time_mean_maps(self):
        return self._aggregators["time_mean"].time_mean_maps()
