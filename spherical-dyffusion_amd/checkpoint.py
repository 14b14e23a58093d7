"""Checkpoint ingestion: Lightning checkpoints of the reference's two experiments -> the MI355X module and stepper.

Mirrors `MultiStepStepper.from_state` / `load_state` (`src/ace_inference/core/stepper_multistep.py:195-295`),
`MultiHorizonForecastingDYffusion.load_state_dict` (interpolator keys skipped,
`src/experiment_types/forecasting_multi_horizon.py:510-513`) and the EMA selection of `BaseExperiment.ema_scope` /
`LitEma.copy_to` (`src/experiment_types/_base_experiment.py:386-401,1236-1242`, `src/models/modules/ema.py:20-27,55-68`).

A checkpoint here is the dictionary `torch.load(path)` returns for a Lightning `.ckpt`: `{"hyper_parameters": {...,
"model_config", "datamodule_config", "diffusion_config", "use_ema", ...}, "state_dict": {...}}`.  The forecaster's
checkpoint does not hold the interpolator's weights (`on_save_checkpoint` drops them), so the interpolator's own
checkpoint is a second argument, exactly as the reference loads it from a second run.

Which weights: the forecaster's EMA shadows iff its `use_ema` hyper-parameter is set (the stepper runs every step under
`module.ema_scope()`), the interpolator's iff `diffusion_config.interpolator_use_ema` of the FORECASTER is set
(`src/diffusion/dyffusion.py:236-237`) - the interpolator's own `use_ema` is not consulted.

MI355X-first difference: EMA weights are baked ONCE at load time (the reference swaps EMA weights in and out of the
module on every `ema_scope()`, i.e. every autoregressive step); `ema_scope()` of the resulting module is free.
Dataset statistics (`data_dir_stats/*.nc` in the reference) are passed as plain `{name: float}` dictionaries.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional, Tuple

import torch

from .experiment import InterpolationExperiment, MultiHorizonForecastingDYffusion
from .sfno import SphericalFourierNeuralOperatorNet
from .stepper import MultiStepStepper, Prescriber

_SFNO_TARGET = "SphericalFourierNeuralOperatorNet"


def _plain(cfg) -> Dict[str, Any]:
    """hyper-parameter containers (OmegaConf DictConfig, AttributeDict, dict) -> dict"""
    if cfg is None:
        return {}
    try:
        from omegaconf import OmegaConf   # optional: real checkpoints store DictConfig objects

        if OmegaConf.is_config(cfg):
            return dict(OmegaConf.to_container(cfg, resolve=True))
    except ImportError:
        pass
    return {k: cfg[k] for k in cfg.keys()}


def select_weights(state_dict: Mapping[str, torch.Tensor], net_prefix: str, ema_handle_prefix: str,
                   use_ema: bool) -> Dict[str, torch.Tensor]:
    """Weights of one network from a Lightning `state_dict`.

    `net_prefix`: where the network's parameters live (`"model.model."` for the forecaster inside the DYffusion module,
    `"model."` for the interpolation experiment).  With `use_ema` the value of parameter `p` is the `LitEma` buffer
    `model_ema.<name with the dots removed>`, the name being relative to the EMA handle (`experiment.model`), i.e.
    `ema_handle_prefix + p` (`ema.py:25-27`); parameters without a shadow (frozen ones) keep their raw value."""
    out = {}
    for k, v in state_dict.items():
        if k.startswith(net_prefix) and not k.startswith("model_ema"):
            out[k[len(net_prefix):]] = v
    if not out:
        raise KeyError(f"no parameters under '{net_prefix}' in the checkpoint (keys start with "
                       f"{sorted({k.split('.')[0] for k in state_dict})})")
    if use_ema:
        n_shadow = 0
        for p in list(out):
            s = "model_ema." + (ema_handle_prefix + p).replace(".", "")
            if s in state_dict:
                out[p] = state_dict[s]
                n_shadow += 1
        if n_shadow == 0:
            raise KeyError("use_ema is set but the checkpoint holds no 'model_ema.*' buffers for this network")
    return out


def module_weights(state: Mapping[str, Any], interpolator_state: Mapping[str, Any], use_ema: Optional[bool] = None,
                   interpolator_use_ema: Optional[bool] = None) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(forecaster weights, interpolator weights) as the reference samples with them.

    Forecaster: the stepper wraps every step in `module.ema_scope()` (stepper_multistep.py:387), which swaps the EMA
    shadows in iff the forecaster's `use_ema` hyper-parameter is set (`_base_experiment.py:386-401`).
    Interpolator: the reference enters ITS `ema_scope` only when the FORECASTER's `diffusion_config.interpolator_use_ema`
    is set (`src/diffusion/dyffusion.py:236-237`); the interpolator's own `use_ema` merely decides whether its checkpoint
    carries `model_ema.*` shadows (loaded next to the raw weights, never copied in by the loader, `src/interface.py:158-166`).
    Shipped configs: every module trains with `use_ema: True` (`configs/experiment/fv3gfs.yaml:14`) and
    `configs/diffusion/dyffusion.yaml:41` sets `interpolator_use_ema: False`, i.e. the published interpolator samples
    with its RAW weights."""
    hp = _plain(state["hyper_parameters"])
    dc = _plain(hp.get("diffusion_config"))
    if use_ema is None:
        use_ema = bool(hp.get("use_ema", False))
    if interpolator_use_ema is None:
        interpolator_use_ema = bool(dc.get("interpolator_use_ema", False))
    fw = select_weights(state["state_dict"], "model.model.", "model.", use_ema)
    iw = select_weights(interpolator_state["state_dict"], "model.", "", interpolator_use_ema)
    return fw, iw


def _build_net(model_config: Mapping[str, Any], n_in: int, n_out: int, n_cond: int, spatial_shape: Tuple[int, int],
               weights: Mapping[str, torch.Tensor], **net_kwargs) -> SphericalFourierNeuralOperatorNet:
    mc = dict(model_config)
    target = str(mc.pop("_target_", _SFNO_TARGET))
    if not target.endswith(_SFNO_TARGET):
        raise NotImplementedError(f"model _target_ {target}: only the SFNO backbone is on the hot path (SURVEY.md section 8)")
    for k in ("loss_function", "verbose", "name"):
        mc.pop(k, None)
    net = SphericalFourierNeuralOperatorNet(num_input_channels=n_in, num_output_channels=n_out,
                                            num_conditional_channels=n_cond, spatial_shape_in=tuple(spatial_shape),
                                            **mc, **net_kwargs)
    net.load_state_dict(weights, strict=True)
    return net


def resolve_interpolator_state(state: Mapping[str, Any], interpolator_state: Optional[Mapping[str, Any]] = None):
    """The interpolator's checkpoint as `DYffusion.__init__` finds it (`src/diffusion/dyffusion.py:612-630` ->
    `get_checkpoint_from_path_or_wandb`, `src/interface.py:193-222`): an explicit one wins; otherwise the forecaster's
    `diffusion_config.interpolator_local_checkpoint_path` is loaded from disk.  A wandb run id alone cannot be resolved here
    (no network on this path): download the file and pass its path or its contents."""
    if interpolator_state is not None:
        return interpolator_state
    dc = _plain(_plain(state["hyper_parameters"]).get("diffusion_config"))
    path = dc.get("interpolator_local_checkpoint_path")
    if isinstance(path, str) and path:
        return torch.load(path, map_location="cpu", weights_only=False)
    raise ValueError("no interpolator checkpoint: pass `interpolator_state`, or set "
                     "diffusion_config.interpolator_local_checkpoint_path to the interpolator's .ckpt "
                     f"(interpolator_run_id={dc.get('interpolator_run_id')!r} needs wandb access, which this path does not have)")


def module_from_state(state: Mapping[str, Any], interpolator_state: Optional[Mapping[str, Any]], spatial_shape: Tuple[int, int],
                      use_ema: Optional[bool] = None, interpolator_use_ema: Optional[bool] = None,
                      device="cuda", **net_kwargs) -> MultiHorizonForecastingDYffusion:
    """The sampling module (`module_class(**hyper_parameters)` + `load_state_dict`, stepper_multistep.py:241-245,207-209)."""
    interpolator_state = resolve_interpolator_state(state, interpolator_state)
    hp, ihp = _plain(state["hyper_parameters"]), _plain(interpolator_state["hyper_parameters"])
    dm, dc = _plain(hp["datamodule_config"]), _plain(hp["diffusion_config"])
    in_names, out_names, forcing = list(dm["in_names"]), list(dm["out_names"]), list(dm.get("forcing_names", []))
    in_names = [n for n in in_names if n not in forcing]
    n_in, n_out, n_cond = len(in_names), len(out_names), len(forcing)
    horizon = int(dm.get("horizon", dc.get("timesteps", 6)))
    fw, iw = module_weights(state, interpolator_state, use_ema, interpolator_use_ema)
    # the interpolator sees (x_0, x_h) stacked on the channel axis (interpolation.py: window + 1 snapshot)
    with torch.cuda.device(device):
        fnet = _build_net(_plain(hp["model_config"]), n_in, n_out, n_cond, spatial_shape, fw, **net_kwargs)
        inet = _build_net(_plain(ihp["model_config"]), 2 * n_in, n_out, n_cond, spatial_shape, iw, **net_kwargs)
    ipol = InterpolationExperiment(inet, horizon=horizon,
                                   enable_inference_dropout=bool(ihp.get("enable_inference_dropout", True)))
    diffusion = {k: v for k, v in dc.items() if k not in ("_target_", "interpolator", "interpolator_run_id",
                                                          "interpolator_local_checkpoint_path",
                                                          "interpolator_wandb_ckpt_filename")}
    return MultiHorizonForecastingDYffusion(fnet, ipol, horizon=horizon, diffusion_config=diffusion,
                                            num_predictions=int(hp.get("num_predictions", 1) or 1),
                                            inputs_noise=float(hp.get("inputs_noise", 0.0) or 0.0))


def stepper_from_state(state: Mapping[str, Any], interpolator_state: Optional[Mapping[str, Any]], means: Mapping[str, float],
                       stds: Mapping[str, float], spatial_shape: Tuple[int, int], overrides: Optional[Dict[str, Any]] = None,
                       **kw) -> MultiStepStepper:
    """`MultiStepStepper.from_state` (stepper_multistep.py:228-295): module + packers (names) + prescriber + normaliser."""
    if overrides:
        state = dict(state)
        hp = _plain(state["hyper_parameters"])
        for k, v in overrides.items():       # update_dict_with_other: nested dictionaries are merged
            hp[k] = {**_plain(hp.get(k)), **v} if isinstance(v, Mapping) else v
        state["hyper_parameters"] = hp
    module = module_from_state(state, interpolator_state, spatial_shape, **kw)
    dm = _plain(_plain(state["hyper_parameters"])["datamodule_config"])
    forcing = list(dm.get("forcing_names", []))
    pres = None
    pc = _plain(dm.get("prescriber")) if dm.get("prescriber") is not None else None
    if pc and pc.get("prescribed_name"):
        pres = Prescriber(pc["prescribed_name"], pc["mask_name"], int(pc["mask_value"]), bool(pc.get("interpolate", False)))
    return MultiStepStepper(module, list(dm["in_names"]) + [f for f in forcing if f not in dm["in_names"]],
                            list(dm["out_names"]), forcing, dict(means), dict(stds), pres)
