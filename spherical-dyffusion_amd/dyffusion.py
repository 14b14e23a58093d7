"""DYffusion cold-sampling loop on the MI355X-native SFNO.

Host-side mirror of `BaseDYffusion` / `DYffusion` (`src/diffusion/dyffusion.py:19-662`) for the sampling path:
same method names, argument meaning and error behaviour (`sample`, `sample_loop`, `predict_forward`, `q_sample`,
`_interpolate`, `predict_x_last`, `diffusion_step_to_interpolation_step`, `sampling_schedule`), so the
experiment shim (`experiment.py`) and an unchanged stepper can drive it.  Training (`p_losses`) is out of scope.

What is different on purpose (MI355X-first):
  * no device->host syncs inside the loop: schedule/time validity is checked on host scalars, the time tensors are
    built from host floats (the reference asserts on device tensors at dyffusion.py:144-146,311,651-653);
  * the pointwise update x_s + (x_ip_next - x_ip_s) and the channel concats are single HIP launches
    (`sdy_cold_update`, `sdy_concat_channels`), the networks are one native call each.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from torch import Tensor

from . import ops


class DYffusion(torch.nn.Module):
    def __init__(
        self,
        model,                                   # forecaster network (callable like SFNO.forward / predict_forward)
        interpolator,                            # InterpolationExperiment-like: predict_packed, inference_dropout_scope
        timesteps: int,
        forward_conditioning: str = "none",
        dynamic_cond_from_t: str = "h",
        schedule: str = "before_t1_only",
        additional_interpolation_steps: int = 0,
        additional_interpolation_steps_factor: int = 0,
        interpolate_before_t1: bool = True,
        sampling_type: str = "cold",
        sampling_schedule: Union[List[float], str, None] = None,
        use_cold_sampling_for_intermediate_steps: bool = True,
        use_cold_sampling_for_last_step: bool = True,
        use_cold_sampling_for_init_of_ar_step: Optional[bool] = None,
        time_encoding: str = "dynamics",
        refine_intermediate_predictions: bool = False,
        prediction_timesteps: Optional[Sequence[float]] = None,
        enable_interpolator_dropout: Union[bool, str] = True,
        interpolator_use_ema: bool = False,
        log_every_t: Union[str, int, None] = None,
        hack_for_imprecise_interpolation: bool = False,
        **unused,
    ):
        super().__init__()
        if schedule != "before_t1_only":
            raise NotImplementedError(f"schedule={schedule!r}: only 'before_t1_only' (the shipped config) is in scope")
        if forward_conditioning != "none":
            raise NotImplementedError(f"forward_conditioning={forward_conditioning!r}: only 'none' is in scope")
        if time_encoding not in ("dynamics", "discrete"):
            raise ValueError(f"Invalid time_encoding: {time_encoding}")
        if enable_interpolator_dropout not in [True, False, "always", "except_dynamical_steps"]:
            raise ValueError(f"invalid enable_interpolator_dropout={enable_interpolator_dropout!r}")
        assert additional_interpolation_steps_factor == 0, \
            "additional_interpolation_steps_factor must be 0 when using before_t1_only schedule"
        assert interpolate_before_t1, "interpolate_before_t1 must be True when using before_t1_only schedule"
        assert timesteps > 1, f"horizon must be > 1, but got {timesteps}. Please use datamodule.horizon with > 1"
        self.model = model
        self.interpolator = interpolator
        if use_cold_sampling_for_init_of_ar_step is None:
            use_cold_sampling_for_init_of_ar_step = use_cold_sampling_for_last_step
        self.use_cold_sampling_for_init_of_ar_step = use_cold_sampling_for_init_of_ar_step
        self.hparams = SimpleNamespace(
            timesteps=timesteps, forward_conditioning=forward_conditioning, dynamic_cond_from_t=dynamic_cond_from_t,
            schedule=schedule, additional_interpolation_steps=additional_interpolation_steps,
            sampling_type=sampling_type, time_encoding=time_encoding,
            use_cold_sampling_for_intermediate_steps=use_cold_sampling_for_intermediate_steps,
            use_cold_sampling_for_last_step=use_cold_sampling_for_last_step,
            refine_intermediate_predictions=refine_intermediate_predictions,
            prediction_timesteps=prediction_timesteps, interpolator_use_ema=interpolator_use_ema,
            log_every_t=log_every_t, hack_for_imprecise_interpolation=hack_for_imprecise_interpolation,
        )
        self.additional_diffusion_steps = additional_interpolation_steps
        self.num_timesteps = timesteps + self.additional_diffusion_steps          # dyffusion.py:56,97
        d_to_i = {d: self.diffusion_step_to_interpolation_step(d) for d in range(1, self.num_timesteps)}
        self.dynamical_steps = {d: i for d, i in d_to_i.items() if float(i).is_integer()}
        self.i_to_diffusion_step = {i: d for d, i in d_to_i.items()}
        self.artificial_interpolation_steps = {d: i for d, i in d_to_i.items() if not float(i).is_integer()}
        self.enable_interpolator_dropout = enable_interpolator_dropout
        self.full_sampling_schedule = list(range(0, self.num_timesteps))
        self.sampling_schedule = sampling_schedule or self.full_sampling_schedule
        # DYffusion.__init__ consistency check (dyffusion.py:632-640)
        self.interpolator_window = getattr(interpolator, "window", 1)
        self.interpolator_horizon = getattr(interpolator, "true_horizon", timesteps)
        last = self.diffusion_step_to_interpolation_step(self.num_timesteps - 1)
        if self.interpolator_horizon != last + 1:
            raise ValueError(f"interpolator horizon {self.interpolator_horizon} must be equal to the "
                             f"last interpolation step+1=i_N=i_{self.num_timesteps - 1}={last + 1}")

    # ---- schedule maths (dyffusion.py:134-188, 273-284, 363-455) ---------------------------------------------
    @property
    def diffusion_steps(self) -> List[int]:
        return list(range(0, self.num_timesteps))

    def diffusion_step_to_interpolation_step(self, diffusion_step):
        assert 0 <= diffusion_step <= self.num_timesteps - 1, \
            f"diffusion_step must be in [0, num_timesteps-1]=[0, {self.num_timesteps - 1}], but got {diffusion_step}"
        k = self.additional_diffusion_steps
        if diffusion_step >= k + 1:
            return diffusion_step - k
        return diffusion_step / (k + 1)

    @property
    def valid_time_range_for_backbone_model(self) -> List[float]:
        if self.hparams.time_encoding == "discrete":
            return self.diffusion_steps
        return [self.diffusion_step_to_interpolation_step(d) for d in self.diffusion_steps]

    @property
    def sampling_schedule(self):
        return self._sampling_schedule

    @sampling_schedule.setter
    def sampling_schedule(self, schedule):
        name = schedule
        if isinstance(schedule, str):
            base = [0] + list(self.dynamical_steps.keys())
            artificial = list(self.artificial_interpolation_steps.keys())
            if "only_dynamics" in name:
                schedule = []
                if "only_dynamics_plus" in name:
                    plus_n = int(name.replace("only_dynamics_plus", "").replace("_discrete", ""))
                    schedule = list(np.linspace(0, base[1], plus_n + 1, endpoint=False))
                    if "_discrete" in name:
                        schedule = [int(np.floor(s)) for s in schedule]
                else:
                    assert name == "only_dynamics", f"Invalid sampling schedule: {schedule}"
            elif name.startswith("every"):
                nth = int(name.replace("every", "").replace("th", "").replace("nd", "").replace("rd", ""))
                assert 1 <= nth <= self.num_timesteps, f"Invalid sampling schedule: {name}"
                schedule = artificial[::nth]
            elif name.startswith("first"):
                first_n = float(name.replace("first", "").replace("v2", ""))
                if first_n < 1:
                    assert 0 < first_n < 1, f"Invalid sampling schedule: {name}, must end with number/float > 0"
                    schedule = artificial[: int(np.ceil(first_n * len(artificial)))]
                else:
                    assert first_n.is_integer() and 1 <= first_n <= self.num_timesteps, f"Invalid sampling schedule: {name}"
                    schedule = artificial[: int(first_n)]
            else:
                raise ValueError(f"Invalid sampling schedule: ``{name}``. ")
            schedule = list(sorted(set(schedule + base)))
        schedule = list(schedule)
        assert 1 <= schedule[-1] <= self.num_timesteps, \
            f"Invalid sampling schedule: {schedule}, must end with number/float <= {self.num_timesteps}"
        if schedule[0] != 0:
            schedule = [0] + schedule
        for i in range(1, len(schedule)):
            assert schedule[i] > schedule[i - 1], f"Invalid sampling schedule not monotonically increasing: {schedule}"
        if all(float(s).is_integer() for s in schedule):
            schedule = [int(s) for s in schedule]
        self._sampling_schedule = schedule

    # ---- network calls ------------------------------------------------------------------------------------------
    @staticmethod
    def _time_tensor(value: float, like: Tensor) -> Tensor:
        return torch.full((like.shape[0],), float(value), dtype=torch.float32, device=like.device)

    def q_sample(self, x0, x_end, t, interpolation_time=None, is_artificial_step: bool = True, **kwargs) -> Tensor:
        """Interpolator call (dyffusion.py:190-240).  `t` is a host scalar diffusion step."""
        assert t is None or interpolation_time is None, "Either t or interpolation_time must be None."
        i_n = interpolation_time if t is None else self.diffusion_step_to_interpolation_step(t)
        dyn = kwargs.pop("dynamical_condition", None)
        if dyn is not None:
            kwargs["condition"] = self.interpolator.get_dynamical_condition(dyn, i_n)
        time = self._time_tensor(i_n, x0)
        do_enable = bool(
            self.enable_interpolator_dropout in [True, "always"]
            or (self.enable_interpolator_dropout == "except_dynamical_steps" and is_artificial_step)
        )
        kwargs.pop("num_predictions", None)
        with self.interpolator.inference_dropout_scope(condition=do_enable):
            return self._interpolate(initial_condition=x_end, x_last=x0, t=time, t_host=i_n, **kwargs)

    def _interpolate(self, initial_condition: Tensor, x_last: Tensor, t: Tensor, t_host=None, num_predictions: int = 1,
                     **kwargs) -> Tensor:
        """dyffusion.py:642-662."""
        if t_host is not None:
            assert 0 < t_host < self.interpolator_horizon, \
                f"interpolate time must be in (0, {self.interpolator_horizon}), got {t_host}"
        hack = self.hparams.hack_for_imprecise_interpolation
        pieces = [initial_condition] + ([initial_condition[:, :1]] if hack else []) + [x_last]
        inputs = ops.concat_channels(pieces)
        out = self.interpolator.predict_packed(inputs, time=t, **kwargs)["preds"]
        if hack:
            out = ops.concat_channels([initial_condition[:, :1], out])
        return out

    def predict_x_last(self, initial_condition: Tensor, x_t: Tensor, t, **kwargs) -> Tensor:
        """Forecaster call (dyffusion.py:286-355); `t` is a host scalar diffusion step."""
        assert 0 <= t <= self.num_timesteps - 1, f"Invalid timestep: {t}. {self.num_timesteps=}"
        dyn = kwargs.pop("dynamical_condition", None)
        cond = None
        if dyn is not None:
            assert dyn.shape[1] == self.num_timesteps + 1, f"{dyn.shape}[1] != {self.num_timesteps + 1}"
            sel = self.hparams.dynamic_cond_from_t
            if sel == "0":
                cond = dyn[:, 0]
            elif sel == "h":
                cond = dyn[:, -1]
            elif sel == "t":
                cond = dyn[:, int(t)]
            else:
                raise ValueError(f"Invalid dynamic_cond_from_t: {sel}")
        time_v = t if self.hparams.time_encoding == "discrete" else self.diffusion_step_to_interpolation_step(t)
        time = self._time_tensor(time_v, x_t)
        return self.model.predict_forward(x_t, time=time, condition=cond, **kwargs)

    # ---- sampler (dyffusion.py:457-577) -----------------------------------------------------------------------------
    def sample_loop(self, initial_condition, log_every_t=None, num_predictions: int = None, verbose=True, **kwargs):
        hp = self.hparams
        log_every_t = log_every_t or hp.log_every_t
        log_every_t = log_every_t if log_every_t != "auto" else 1
        sched = self.sampling_schedule
        assert len(initial_condition.shape) == 4, f"condition.shape: {initial_condition.shape} (should be 4D)"
        N = self.num_timesteps
        hack = hp.hack_for_imprecise_interpolation
        intermediates, xhat_th, dynamics_pred_step = dict(), None, 0
        last_p1 = sched[-1] + 1
        triples = zip(sched, sched[1:] + [last_p1], sched[2:] + [last_p1, last_p1 + 1])
        x_s = initial_condition
        for s, s_next, s_nnext in triples:
            is_last_step = s == N - 1
            xhat_th = self.predict_x_last(initial_condition=initial_condition, x_t=x_s, t=s, **dict(kwargs))
            time_i_n = self.diffusion_step_to_interpolation_step(s_next) if not is_last_step else np.inf
            is_dynamics_pred = float(time_i_n).is_integer() or is_last_step
            q_kwargs = dict(x0=xhat_th, x_end=initial_condition, is_artificial_step=not is_dynamics_pred)
            if s_next <= N - 1:
                x_ip_next = self.q_sample(**q_kwargs, t=s_next, **dict(kwargs))
            else:
                assert is_last_step, f"Invalid s_next: {s_next} (should be <= {N - 1})"
                x_ip_next = xhat_th
                if hack:
                    x_ip_next = ops.concat_channels([initial_condition[:, :1], x_ip_next])
            x_ip_s = None
            if hp.sampling_type == "cold":
                if not hp.use_cold_sampling_for_last_step and is_last_step:
                    if self.use_cold_sampling_for_init_of_ar_step:
                        x_ip_s = self.q_sample(**q_kwargs, t=s, **dict(kwargs))
                        ar_init = ops.cold_update(x_s, xhat_th, x_ip_s)
                        intermediates["preds_autoregressive_init"] = ar_init[:, 1:] if hack else ar_init
                    x_s = xhat_th
                else:
                    x_ip_s = self.q_sample(**q_kwargs, t=s, **dict(kwargs)) if s > 0 else None
                    x_s = ops.cold_update(x_s, x_ip_next, x_ip_s)     # x_s + (x_ip_next - x_ip_s); s=0: x_ip_s == x_s
            elif hp.sampling_type == "naive":
                x_s = x_ip_next
            else:
                raise ValueError(f"unknown sampling type {hp.sampling_type}")
            dynamics_pred_step = int(time_i_n) if s < N - 1 else dynamics_pred_step + 1
            if is_dynamics_pred:
                preds_t = x_s if (hp.use_cold_sampling_for_intermediate_steps or is_last_step) else x_ip_next
                intermediates[f"t{dynamics_pred_step}_preds"] = preds_t[:, 1:] if hack else preds_t
                if log_every_t is not None:
                    intermediates[f"t{dynamics_pred_step}_preds2"] = x_ip_next
            if log_every_t is not None:
                intermediates[f"x_{s}_dmodel"] = x_s
                intermediates[f"intermediate_{s}_x0hat"] = xhat_th
                intermediates[f"xipol_{s}_dmodel"] = x_ip_next
                if hp.sampling_type == "cold" and x_ip_s is not None:
                    intermediates[f"xipol_{s}_dmodel2"] = x_ip_s
        if hp.refine_intermediate_predictions:
            steps = hp.prediction_timesteps or list(self.dynamical_steps.values())
            for i_n in [i for i in steps if i < N]:
                key = int(i_n) if float(i_n).is_integer() else i_n
                assert not float(i_n).is_integer() or f"t{key}_preds" in intermediates, f"t{key}_preds not in intermediates"
                r = self.q_sample(x0=xhat_th, x_end=initial_condition, is_artificial_step=False, t=None,
                                  interpolation_time=i_n, **dict(kwargs))
                intermediates[f"t{key}_preds"] = r[:, 1:] if hack else r
        if last_p1 < N:
            return x_s, intermediates
        return xhat_th, intermediates

    @torch.inference_mode()
    def sample(self, initial_condition, num_samples=1, **kwargs) -> Dict[str, Tensor]:
        _, intermediates = self.sample_loop(initial_condition, **kwargs)
        return intermediates

    def predict_forward(self, *inputs, metadata: Any = None, **kwargs):
        assert len(inputs) == 1, "Only one input tensor is allowed for the forward pass"
        return self.sample(inputs[0], **kwargs)
