"""Predict surface of the reference's LightningModules, minus Lightning.

Mirrors what `run_inference.py`'s stepper calls on the module
(`src/ace_inference/core/stepper_multistep.py:387-399`):
  * `InterpolationExperiment`  (`src/experiment_types/interpolation.py:13-69,133-141`)
  * `MultiHorizonForecastingDYffusion.get_preds_at_t_for_batch` (`src/experiment_types/forecasting_multi_horizon.py:331-381`)
  * `BaseExperiment.predict / _predict / predict_packed / ema_scope / inference_dropout_scope`
    (`src/experiment_types/_base_experiment.py:386-423,473-572`)
and the Lightning-driven entry (`src/interface.py:302-313` -> `trainer.predict`): `predict_step` / `on_predict_epoch_end` /
`_evaluation_get_preds` (`_base_experiment.py:1083-1102,906-919`) over the autoregressive loop of `_evaluation_step`
(`src/experiment_types/forecasting_multi_horizon.py:139-320`), for batches that are already packed and normalised;
`interface.run_inference` drives it without Lightning.
Same method names, argument meaning, returned keys (`t{h}_preds_normed`), and the same statefulness
(`_current_preds` cache: not re-entrant).  Un-normalisation/unpacking needs the datamodule's statistics, which are
outside the hot-path scope: without a datamodule the reference, too, returns only the
`*_normed` entries (`_base_experiment.py:559-563`).
"""
from __future__ import annotations

from contextlib import contextmanager
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .dyffusion import DYffusion


class _BaseExperiment(torch.nn.Module):
    def __init__(self, model, horizon: int, window: int = 1, enable_inference_dropout: bool = False,
                 num_predictions: int = 1, use_ema: bool = False):
        super().__init__()
        self.model = model
        self.horizon = horizon
        self.window = window
        self.num_predictions = num_predictions
        self.num_predictions_in_mem = num_predictions
        self.num_prediction_loops = 1
        self.use_ema = use_ema          # EMA weights are baked in at load time (one-off), so the scope is a no-op
        self.hparams = SimpleNamespace(enable_inference_dropout=enable_inference_dropout, use_ema=use_ema)
        self.datamodule = None

    @property
    def true_horizon(self) -> int:
        return self.horizon

    @contextmanager
    def ema_scope(self, context=None, force_non_ema: bool = False, condition: bool = None):
        """`_base_experiment.py:386-401`.  The reference clones/swaps all parameters on every call; here EMA weights
        are selected once when the checkpoint is loaded, so entering the scope costs nothing."""
        yield None

    @contextmanager
    def inference_dropout_scope(self, condition: bool = None, context=None):
        """`_base_experiment.py:403-423`."""
        condition = self.hparams.enable_inference_dropout if condition is None else condition
        if condition:
            self.model.enable_inference_dropout()
        try:
            yield None
        finally:
            if condition:
                self.model.disable_inference_dropout()

    def predict_packed(self, *inputs: Tensor, **kwargs) -> Dict[str, Tensor]:
        """`_base_experiment.py:473-485`."""
        kwargs.pop("num_predictions", None)
        results = self.model.predict_forward(*inputs, **kwargs)
        if torch.is_tensor(results):
            results = {"preds": results}
        return results

    def _predict(self, *inputs: Tensor, num_predictions: Optional[int] = None, predictions_mask=None, **kwargs):
        """`_base_experiment.py:487-547` with num_prediction_loops == 1, then `postprocess_predictions` (:549-565)
        without a datamodule: keys containing 'preds' are renamed `<key>_normed`."""
        results = self.predict_packed(*inputs, **kwargs)
        if predictions_mask is not None:
            results = {k: v[..., predictions_mask[0, :]] for k, v in results.items()}
        for k in list(results.keys()):
            if "preds" in k:
                results[f"{k}_normed"] = results.pop(k)
        return results

    def predict(self, inputs, **kwargs) -> Dict[str, Tensor]:
        """`_base_experiment.py:567-572`."""
        if torch.is_tensor(inputs):
            return self._predict(inputs, **kwargs)
        return self._predict(**inputs, **kwargs)


class InterpolationExperiment(_BaseExperiment):
    def __init__(self, model, horizon: int, window: int = 1, enable_inference_dropout: bool = True, **kw):
        super().__init__(model, horizon, window, enable_inference_dropout, **kw)
        assert self.horizon >= 2, "horizon must be >=2 for interpolation experiments"
        if hasattr(self.model, "set_min_max_time"):       # interpolation.py:23-24
            self.model.set_min_max_time(min_time=self.horizon_range[0], max_time=self.horizon_range[-1])

    @property
    def horizon_range(self) -> List[int]:
        return list(np.arange(1, self.horizon))

    def get_dynamical_condition(self, dynamical_condition: Optional[Tensor], target_time) -> Optional[Tensor]:
        """interpolation.py:133-141"""
        if dynamical_condition is None:
            return None
        if isinstance(target_time, (int, np.integer)):
            return dynamical_condition[:, int(target_time), ...]
        return dynamical_condition[torch.arange(dynamical_condition.shape[0]), target_time.long(), ...]


class MultiHorizonForecastingDYffusion(_BaseExperiment):
    """Forecaster experiment whose `model` is the DYffusion sampler wrapping the forecaster network."""

    inputs_data_key = "dynamics"

    def __init__(self, forecaster_net, interpolator: InterpolationExperiment, horizon: int, window: int = 1,
                 enable_inference_dropout: bool = False, diffusion_config: Optional[dict] = None,
                 inputs_noise: float = 0.0, **kw):
        diffusion_config = dict(diffusion_config or {})
        diffusion_config.setdefault("timesteps", horizon)
        sampler = DYffusion(model=forecaster_net, interpolator=interpolator, **diffusion_config)
        super().__init__(sampler, horizon, window, enable_inference_dropout, **kw)
        self._net = forecaster_net
        # forecasting_multi_horizon.py:52-57 (time range of the backbone = the diffusion model's valid range)
        rng = sampler.valid_time_range_for_backbone_model
        forecaster_net.set_min_max_time(min_time=rng[0], max_time=rng[-1])
        self._prediction_timesteps = None
        self._current_preds: Optional[Dict[str, Tensor]] = None
        self._predict_step_outputs: List[Dict[str, np.ndarray]] = []
        self.inputs_noise = float(inputs_noise)      # _base_experiment.py:66: std of the members' input perturbation

    @property
    def horizon_range(self) -> List[int]:
        return list(np.arange(1, self.horizon + 1))

    @property
    def prediction_timesteps(self):
        return self._prediction_timesteps or self.horizon_range

    @contextmanager
    def inference_dropout_scope(self, condition: bool = None, context=None):
        condition = self.hparams.enable_inference_dropout if condition is None else condition
        if condition:
            self._net.enable_inference_dropout()
        try:
            yield None
        finally:
            if condition:
                self._net.disable_inference_dropout()

    def set_batch_offset(self, offset: int) -> None:
        """Global index of the first trajectory of the batches that follow: keys the dropout streams of both networks, so a
        trajectory's draws do not depend on how members / initial conditions are grouped or sharded."""
        self.model.model.batch_offset = int(offset)                      # forecaster SFNO (no dropout today)
        self.model.interpolator.model.batch_offset = int(offset)         # interpolator SFNO

    def dropout_calls(self):
        """Position of both networks in their dropout streams (the call counters that key the Philox stream beside the
        trajectory index).  A driver that runs one window as several device batches restores it before each batch
        (`set_dropout_calls`), so a trajectory's draws do not depend on how the window was cut (`run_inference(max_batch=)`)."""
        return (self.model.model._call, self.model.interpolator.model._call)

    def set_dropout_calls(self, state) -> None:
        self.model.model._call, self.model.interpolator.model._call = int(state[0]), int(state[1])

    # ---- Lightning predict surface (src/interface.py:302-313 -> trainer.predict -> predict_step) -------------------------
    def use_ensemble_predictions(self, split: str = "predict") -> bool:       # _base_experiment.py:845-846
        return self.num_predictions > 1 and split in ("val", "test", "predict")

    def get_ensemble_inputs(self, inputs_raw: Optional[Tensor], split: str = "predict", add_noise: bool = True):
        """`_base_experiment.py:851-890` for tensors: the batch of `num_predictions` copies, member-major `(N B) ...`; with
        `inputs_noise` > 0 every copy gets its own Gaussian perturbation (drawn by torch on the device: plumbing)."""
        if inputs_raw is None or not self.use_ensemble_predictions(split):
            return inputs_raw
        n = self.num_predictions
        if add_noise and self.inputs_noise > 0:
            copies = [inputs_raw + self.inputs_noise * torch.randn_like(inputs_raw) for _ in range(n)]
        else:
            copies = [inputs_raw] * n
        return torch.stack(copies, dim=0).flatten(0, 1)

    def get_inputs_and_extra_kwargs(self, batch: Dict[str, Tensor], split: str = "predict", ensemble: bool = False,
                                    is_autoregressive: bool = False):
        """`forecasting_multi_horizon.py:383-455` for PACKED, NORMALISED tensors (the datamodule's packing and statistics are
        outside the path): the first `window` steps of `dynamics` stacked on the channel axis, replicated
        per member unless the batch already is the members' autoregressive state; `static_condition` replicated;
        `dynamical_condition` passed whole (the sampler picks the times, :494-498) and replicated."""
        dyn = batch[self.inputs_data_key]
        inputs = dyn[:, : self.window].flatten(1, 2)
        if ensemble and not is_autoregressive:
            inputs = self.get_ensemble_inputs(inputs, split)
        extra = {}
        for k, v in batch.items():
            if k == self.inputs_data_key or k == "metadata":
                continue
            if k in ("static_condition", "dynamical_condition"):
                extra[k] = self.get_ensemble_inputs(v, split, add_noise=False) if ensemble else v
            else:
                raise ValueError(f"Unsupported key {k} in batch")
        return inputs, extra

    def predict_step(self, batch: Dict[str, Tensor], batch_idx: int = 0, dataloader_idx: Optional[int] = None,
                     prediction_horizon: Optional[int] = None, return_outputs: str = "all") -> Dict[str, np.ndarray]:
        """`BaseExperiment.predict_step` (`_base_experiment.py:1083-1096`) -> `evaluation_step` (:807-832: EMA and
        inference-dropout scopes) -> `_evaluation_step` (`forecasting_multi_horizon.py:139-320`): the autoregressive loop
        over `prediction_horizon` steps in chunks of the training horizon, `num_predictions` ensemble members batched on the
        device.  `batch`: `dynamics` (B, >= window + prediction_horizon, C, H, W) ALREADY packed and normalised (targets are
        read from it), optional `dynamical_condition` (B, >= prediction_horizon + 1, Cc, H, W) / `static_condition`.
        Returns numpy arrays like the reference: `t{k}_preds_normed` (N, B, C, H, W) for an ensemble, (B, C, H, W)
        otherwise, and `t{k}_targets_normed` (B, C, H, W); `on_predict_epoch_end` concatenates the batches."""
        split = "predict"
        if self.window != 1:
            raise NotImplementedError("predict_step: window > 1 is not on the sampling path (shipped configs: window 1)")
        batch = dict(batch)
        dyn_full = batch[self.inputs_data_key]
        if not dyn_full.is_cuda:
            raise RuntimeError("sdy_amd predict_step runs on the GPU only (no CPU fallback); move the batch to cuda")
        H = self.true_horizon
        if prediction_horizon is None:
            prediction_horizon = getattr(self.hparams, "prediction_horizon", None) or H
        if dyn_full.shape[1] < prediction_horizon:
            raise ValueError(f"Prediction horizon {prediction_horizon} is larger than {tuple(dyn_full.shape)}[1]")
        dynamic_conds = batch.pop("dynamical_condition", None)
        n_outer = -(-int(prediction_horizon) // H)          # num_autoregressive_steps_for_horizon(...) + 1
        batch[self.inputs_data_key] = dyn_full[:, : self.window + H]
        ens = self.use_ensemble_predictions(split)
        out: Dict[str, np.ndarray] = {}
        to_np = lambda t: None if t is None else t.detach().cpu().numpy()   # noqa: E731
        with self.ema_scope(), self.inference_dropout_scope():
            for ar_step in range(n_outer):
                ar_window_steps = []
                for t_step in self.prediction_timesteps:
                    total = ar_step * H + t_step
                    if total > prediction_horizon:
                        break
                    if dynamic_conds is not None:     # ar_step 0 -> slice(0, H + 1), 1 -> slice(H, 2 H + 1), ...
                        batch["dynamical_condition"] = dynamic_conds[:, ar_step * H:(ar_step + 1) * H + 1]
                    res = self.get_preds_at_t_for_batch(batch, t_step, split, is_autoregressive=ar_step > 0, ensemble=True)
                    preds = res.pop(f"t{t_step}_preds_normed")
                    target_time = self.window + int(total) - 1
                    target = dyn_full[:, target_time] if target_time < dyn_full.shape[1] else None
                    if return_outputs in (True, "all"):
                        out[f"t{total}_targets_normed"] = to_np(target)
                    out[f"t{total}_preds_normed"] = to_np(preds)
                    if t_step == self.horizon_range[-1]:
                        ar_init = res.pop("preds_autoregressive_init_normed", preds)
                        if ens:
                            ar_init = ar_init.flatten(0, 1)
                        ar_window_steps.append(ar_init)
                if ar_step < n_outer - 1:
                    batch[self.inputs_data_key] = torch.stack(ar_window_steps, dim=1)
        self._current_preds = None
        self._predict_step_outputs.append(out)
        return out

    def _evaluation_get_preds(self, outputs, split: str = "predict") -> Dict[str, np.ndarray]:
        """`_base_experiment.py:906-919`: the batches' results concatenated along the batch axis (axis 1 for ensemble
        predictions, whose leading axis is the member)."""
        if isinstance(outputs, list) and len(outputs) == 1 and isinstance(outputs[0], list):
            outputs = outputs[0]
        ens = self.use_ensemble_predictions(split)
        res = {}
        for key in outputs[0].keys():
            axis = 1 if (ens and "targets" not in key and "true" not in key) else 0
            vals = [o[key] for o in outputs]
            res[key] = None if any(v is None for v in vals) else np.concatenate(vals, axis=axis)
        return res

    def on_predict_epoch_end(self) -> Dict[str, np.ndarray]:                 # _base_experiment.py:1098-1102
        results = self._evaluation_get_preds(self._predict_step_outputs, split="predict")
        self._predict_step_outputs = []
        return results

    def get_preds_at_t_for_batch(self, batch: Dict[str, Tensor], horizon, split: str = "predict", ensemble: bool = False,
                                 is_autoregressive: bool = False, prepare_inputs: bool = True, **kwargs):
        """forecasting_multi_horizon.py:331-381 (cache_preds branch: DYffusion predicts all horizons at once)."""
        assert 0 < horizon <= self.true_horizon, f"horizon={horizon} must be in [1, {self.true_horizon}]"
        if horizon == self.prediction_timesteps[0]:
            if prepare_inputs:     # packed, normalised tensors (the datamodule's packing is outside the path)
                inputs, extra = self.get_inputs_and_extra_kwargs(batch, split=split, ensemble=ensemble,
                                                                 is_autoregressive=is_autoregressive)
            else:                  # what the stepper does (stepper_multistep.py:389-397)
                extra = dict(batch)
                inputs = extra.pop(self.inputs_data_key)
            kwargs.pop("num_predictions", None)
            with torch.inference_mode():
                self._current_preds = self.predict(inputs, **extra, **kwargs)
                if prepare_inputs and self.use_ensemble_predictions(split):
                    # reshape_predictions (_base_experiment.py:892-904): (N B) ... -> N B ...
                    n = self.num_predictions
                    self._current_preds = {k: v.reshape(n, v.shape[0] // n, *v.shape[1:])
                                           for k, v in self._current_preds.items()}
        assert self._current_preds is not None, "call with horizon == prediction_timesteps[0] first"
        preds_key = f"t{horizon}_preds"
        results = {k: self._current_preds.pop(k) for k in list(self._current_preds.keys()) if preds_key in k}
        if horizon == self.horizon_range[-1]:
            assert all(["preds" not in k or "preds_autoregressive_init" in k for k in self._current_preds.keys()]), (
                f'{preds_key=} must be the only key containing "preds" in last prediction. '
                f"Got: {list(self._current_preds.keys())}")
            results = {**results, **self._current_preds}
            self._current_preds = None
        return results
