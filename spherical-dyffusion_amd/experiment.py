"""Predict surface of the reference's LightningModules, minus Lightning.

Mirrors what `run_inference.py`'s stepper calls on the module
(`src/ace_inference/core/stepper_multistep.py:387-399`):
  * `InterpolationExperiment`  (`src/experiment_types/interpolation.py:13-69,133-141`)
  * `MultiHorizonForecastingDYffusion.get_preds_at_t_for_batch` (`src/experiment_types/forecasting_multi_horizon.py:331-381`)
  * `BaseExperiment.predict / _predict / predict_packed / ema_scope / inference_dropout_scope`
    (`src/experiment_types/_base_experiment.py:386-423,473-572`)
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
                 enable_inference_dropout: bool = False, diffusion_config: Optional[dict] = None, **kw):
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

    def get_preds_at_t_for_batch(self, batch: Dict[str, Tensor], horizon, split: str = "predict", ensemble: bool = False,
                                 is_autoregressive: bool = False, prepare_inputs: bool = True, **kwargs):
        """forecasting_multi_horizon.py:331-381 (cache_preds branch: DYffusion predicts all horizons at once)."""
        assert 0 < horizon <= self.true_horizon, f"horizon={horizon} must be in [1, {self.true_horizon}]"
        if horizon == self.prediction_timesteps[0]:
            if prepare_inputs:
                raise NotImplementedError("prepare_inputs=True needs the datamodule (out of scope); the stepper "
                                          "calls with prepare_inputs=False (stepper_multistep.py:389-397)")
            batch = dict(batch)
            inputs = batch.pop(self.inputs_data_key)
            kwargs.pop("num_predictions", None)
            with torch.inference_mode():
                self._current_preds = self.predict(inputs, **batch, **kwargs)
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
