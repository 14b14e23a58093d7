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
        fuse_interpolator_pair_max_batch: int = 8,
        reuse_interpolator_encoder: bool = True,
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
        # `log_every_t` only adds diagnostic tensors to what `sample_loop` returns (dyffusion.py:535-547); `sample()` drops
        # them again, so a checkpoint that sets it loads and samples the same: accepted, nothing is logged.
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
        # A cold-sampling step interpolates the same (x_0, forecast) pair to two times (s' and s).  Up to this batch size the
        # two calls run as ONE forward of 2B rows (per-row time, per-row dropout call number: identical results); larger
        # batches already fill the GPU and would only pay for the stacked copy of the inputs.
        self.fuse_interpolator_pair_max_batch = int(fuse_interpolator_pair_max_batch)
        # Beyond that batch size the pair runs as two forwards, the second one restarting from the first one's encoder output
        # (same inputs: sdy_sfno_fwd_args.reuse_encoder); False = two full forwards (A/B, tests)
        import os
        self.reuse_interpolator_encoder = bool(reuse_interpolator_encoder) and os.environ.get("SDY_NO_ENCODER_REUSE") is None
        # the stacked pair hands its (identical) inputs over once and encodes them once; SDY_NO_SHARED_INPUTS=1: stacked copies (A/B)
        self.share_pair_inputs = os.environ.get("SDY_NO_SHARED_INPUTS") is None
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

    def _named_schedule(self, name: str) -> List[float]:
        """The reference's schedule names (documented at dyffusion.py:369-382): every name visits step 0 and all steps that
        land on a data time step; what differs is which of the ARTIFICIAL steps (additional_interpolation_steps > 0) join:
        "only_dynamics" none; "only_dynamics_plus<N>[_discrete]" N + 1 points spaced evenly in [0, first data step) (floored
        when discrete); "every<N>[th|nd|rd]" every N-th artificial step; "first<N>" the first N of them, "first<x>" with
        0 < x < 1 the first ceil(x * count)."""
        import math
        import re

        artificial = sorted(self.artificial_interpolation_steps)
        first_data_step = min(self.dynamical_steps) if self.dynamical_steps else self.num_timesteps - 1
        m = re.fullmatch(r"only_dynamics(?:_plus(_discrete)?(\d+)(_discrete)?)?", name)
        if m:
            extra: List[float] = []
            if m.group(2) is not None:
                n = int(m.group(2)) + 1
                extra = [first_data_step * i / n for i in range(n)]
                if m.group(1) or m.group(3):
                    extra = [float(math.floor(v)) for v in extra]
        elif (m := re.fullmatch(r"every(\d+)(?:th|nd|rd|st)?", name)):
            stride = int(m.group(1))
            assert 1 <= stride <= self.num_timesteps, f"Invalid sampling schedule: {name}"
            extra = artificial[::stride]
        elif (m := re.fullmatch(r"first(\d*\.?\d+)(?:v2)?", name)):
            amount = float(m.group(1))
            if amount < 1:
                assert amount > 0, f"Invalid sampling schedule: {name}, must end with number/float > 0"
                count = math.ceil(amount * len(artificial))
            else:
                assert amount.is_integer(), f"If first_n >= 1, it must be an integer, but got {amount}"
                assert amount <= self.num_timesteps, f"Invalid sampling schedule: {name}"
                count = int(amount)
            extra = artificial[:count]
        else:
            raise ValueError(f"Invalid sampling schedule: ``{name}``. ")
        return sorted({0, *self.dynamical_steps, *extra})

    @sampling_schedule.setter
    def sampling_schedule(self, schedule):
        """Diffusion steps the sampler visits (reference setter: dyffusion.py:384-455): an explicit list, or one of the
        reference's schedule names (`_named_schedule`)."""
        if isinstance(schedule, str):
            schedule = self._named_schedule(schedule)
        steps = [int(v) if float(v).is_integer() else float(v) for v in schedule]
        if not steps or steps[0] != 0:
            steps = [0] + steps
        assert 1 <= steps[-1] <= self.num_timesteps, \
            f"Invalid sampling schedule: {steps}, must end with number/float <= {self.num_timesteps}"
        assert all(b > a for a, b in zip(steps, steps[1:])), \
            f"Invalid sampling schedule not monotonically increasing: {steps}"
        self._sampling_schedule = steps

    # ---- network calls ------------------------------------------------------------------------------------------
    @staticmethod
    def _time_tensor(value: float, like: Tensor) -> Tensor:
        return torch.full((like.shape[0],), float(value), dtype=torch.float32, device=like.device)

    @staticmethod
    def _check_time_range(net, value: float) -> None:
        """The network's own range check (`sfnonet.py:780-782`, a device assert there) on the host scalar the time tensor is
        built from: sampling with artificial steps asks an interpolator for times in (0, 1), which one whose range
        `InterpolationExperiment` pinned to the data steps [1, horizon - 1] (`interpolation.py:24-31`) refuses."""
        lo, hi = getattr(net, "min_time", None), getattr(net, "max_time", None)
        if lo is not None and hi is not None:
            assert lo <= value <= hi, f"time must be in [{lo}, {hi}], but time is {value}"

    def q_sample(self, x0, x_end, t, interpolation_time=None, is_artificial_step: bool = True, _keep_packed=None,
                 **kwargs) -> Tensor:
        """Interpolator call (dyffusion.py:190-240).  `t` is a host scalar diffusion step.  `_keep_packed`: a list that
        receives the call's packed input concat (q_sample_two hands it to its second call; nothing is kept on the module)."""
        assert t is None or interpolation_time is None, "Either t or interpolation_time must be None."
        i_n = interpolation_time if t is None else self.diffusion_step_to_interpolation_step(t)
        dyn = kwargs.pop("dynamical_condition", None)
        if dyn is not None:
            kwargs["condition"] = self.interpolator.get_dynamical_condition(dyn, i_n)
        time = self._time_tensor(i_n, x0)
        self._check_time_range(getattr(self.interpolator, "model", None), float(i_n))
        do_enable = bool(
            self.enable_interpolator_dropout in [True, "always"]
            or (self.enable_interpolator_dropout == "except_dynamical_steps" and is_artificial_step)
        )
        kwargs.pop("num_predictions", None)
        with self.interpolator.inference_dropout_scope(condition=do_enable):
            return self._interpolate(initial_condition=x_end, x_last=x0, t=time, t_host=i_n, _keep_packed=_keep_packed, **kwargs)

    def _interpolate(self, initial_condition: Tensor, x_last: Tensor, t: Tensor, t_host=None, num_predictions: int = 1,
                     _packed_inputs: Optional[Tensor] = None, _keep_packed=None, **kwargs) -> Tensor:
        """dyffusion.py:642-662.  `_packed_inputs`: the channel concat of a previous call on the same (x_end, x0)."""
        if t_host is not None:
            assert 0 < t_host < self.interpolator_horizon, \
                f"interpolate time must be in (0, {self.interpolator_horizon}), got {t_host}"
        hack = self.hparams.hack_for_imprecise_interpolation
        if _packed_inputs is None:
            pieces = [initial_condition] + ([initial_condition[:, :1]] if hack else []) + [x_last]
            inputs = ops.concat_channels(pieces)
        else:
            inputs = _packed_inputs
        if _keep_packed is not None:
            _keep_packed.append(inputs)
        out = self.interpolator.predict_packed(inputs, time=t, **kwargs)["preds"]
        if hack:
            out = ops.concat_channels([initial_condition[:, :1], out])
        return out

    def _can_reuse_encoder(self, kwargs) -> bool:
        """The second interpolation of a cold-sampling step can restart from the first one's encoder output when the two
        calls see the same inputs: same (x_0, forecast) by construction, and no time-dependent condition."""
        net = getattr(self.interpolator, "model", None)
        return (self.reuse_interpolator_encoder and kwargs.get("dynamical_condition") is None
                and hasattr(net, "batch_offset") and getattr(net, "mask_injector", None) is None)

    def q_sample_two(self, x0, x_end, t_first, t_second, is_artificial_step: bool = True, **kwargs):
        """`(q_sample(t=t_first), q_sample(t=t_second))` as two interpolator forwards that share ONE input concat and ONE
        encoder pass (`reuse_encoder`): the encoder sees only (x_end, x0, static condition), which are the same for both
        calls; time embedding and dropout enter behind it.  Bit-identical to the two plain calls."""
        packed = []     # (lives for this call only: 0.8 GB at 25 members, 180 x 360, 128 channels)
        first = self.q_sample(x0=x0, x_end=x_end, t=t_first, is_artificial_step=is_artificial_step, _keep_packed=packed,
                              **dict(kwargs))
        second = self.q_sample(x0=x0, x_end=x_end, t=t_second, is_artificial_step=is_artificial_step,
                               _packed_inputs=packed[0], reuse_encoder=True, **dict(kwargs))
        return first, second

    def _can_stack_calls(self) -> bool:
        """Stacked calls need a network that numbers its dropout calls per row (`rows_per_call`); injected masks (tests that
        replay the reference's recorded masks) address one call per forward."""
        net = getattr(self.interpolator, "model", None)
        return hasattr(net, "batch_offset") and getattr(net, "mask_injector", None) is None

    def q_sample_pair(self, x0, x_end, t_first, t_second, is_artificial_step: bool = True, **kwargs):
        """`(q_sample(t=t_first), q_sample(t=t_second))` for the same (x0, x_end) as ONE interpolator forward of 2B rows
        (reference call sites dyffusion.py:497 and :515): rows 0..B-1 carry `t_first` and the dropout stream of the first
        call, rows B..2B-1 `t_second` and that of the second call, so the results equal the two separate calls."""
        B = x0.shape[0]
        i_a, i_b = (self.diffusion_step_to_interpolation_step(t) for t in (t_first, t_second))
        for i_n in (i_a, i_b):
            assert 0 < i_n < self.interpolator_horizon, f"interpolate time must be in (0, {self.interpolator_horizon}), got {i_n}"
            self._check_time_range(getattr(self.interpolator, "model", None), float(i_n))
        kwargs.pop("num_predictions", None)
        dyn = kwargs.pop("dynamical_condition", None)
        two = lambda v: torch.cat([v, v], dim=0)  # noqa: E731
        time = torch.cat([self._time_tensor(i_a, x0), self._time_tensor(i_b, x0)])
        hack = self.hparams.hack_for_imprecise_interpolation
        inputs = ops.concat_channels([x_end] + ([x_end[:, :1]] if hack else []) + [x0])
        # Without a time-dependent condition the two calls see the SAME input rows: they are handed over once
        # (`shared_inputs`: row b of the stacked batch reads input row b % B) and the encoder runs once.
        share = dyn is None and self.share_pair_inputs and \
            getattr(getattr(self.interpolator, "model", None), "supports_shared_inputs", False)
        if not share:
            kwargs = {k: (two(v) if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v) for k, v in kwargs.items()}
            if dyn is not None:   # the condition at each call's own time
                kwargs["condition"] = torch.cat([self.interpolator.get_dynamical_condition(dyn, i_a),
                                                 self.interpolator.get_dynamical_condition(dyn, i_b)], dim=0)
            inputs = two(inputs)
        else:
            kwargs["shared_inputs"] = True
        do_enable = bool(
            self.enable_interpolator_dropout in [True, "always"]
            or (self.enable_interpolator_dropout == "except_dynamical_steps" and is_artificial_step)
        )
        with self.interpolator.inference_dropout_scope(condition=do_enable):
            out = self.interpolator.predict_packed(inputs, time=time, rows_per_call=B, **kwargs)["preds"]
        if hack:
            out = ops.concat_channels([two(x_end[:, :1]), out])
        return out[:B], out[B:]

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
        self._check_time_range(self.model, float(time_v))
        return self.model.predict_forward(x_t, time=time, condition=cond, **kwargs)

    # ---- sampler (dyffusion.py:457-577) -----------------------------------------------------------------------------
    def _drop_carried_channel(self, x: Tensor) -> Tensor:
        """`hack_for_imprecise_interpolation`: the state carries one input-only channel in front (dyffusion.py:41-44),
        which every emitted prediction drops again (:509-510,532-533)."""
        return x[:, 1:] if self.hparams.hack_for_imprecise_interpolation else x

    def sample_loop(self, initial_condition, log_every_t=None, num_predictions: int = None, verbose=True, **kwargs):
        """Cold sampling (Alg. 2 of the paper; reference dyffusion.py:457-567): per visited diffusion step s one forecaster
        call x_hat = F(x_s, s), then x_{s'} = x_s + I(x_0, x_hat, s') - I(x_0, x_hat, s) with the interpolator I drawing
        fresh dropout masks on every call.  Returns (final state, {"t{k}_preds": ...}) like the reference, minus the
        `log_every_t` diagnostics (which `sample()` discards anyway)."""
        hp = self.hparams
        if hp.sampling_type not in ("cold", "naive"):
            raise ValueError(f"unknown sampling type {hp.sampling_type}")
        assert len(initial_condition.shape) == 4, f"condition.shape: {initial_condition.shape} (should be 4D)"
        sched, last = self.sampling_schedule, self.num_timesteps - 1
        cold = hp.sampling_type == "cold"
        preds: Dict[str, Tensor] = {}
        x_s, x_hat, k = initial_condition, None, 0
        for pos, s in enumerate(sched):
            s_next = sched[pos + 1] if pos + 1 < len(sched) else sched[-1] + 1
            final = s == last                      # the forecast itself is the last state: no interpolation to s_next
            x_hat = self.predict_x_last(initial_condition=initial_condition, x_t=x_s, t=s, **dict(kwargs))
            x_at_s = None          # I(x_0, x_hat, s), when it came out of a fused pair
            i_next = self.diffusion_step_to_interpolation_step(s_next) if not final else None
            lands_on_data = final or float(i_next).is_integer()
            ipol = dict(x0=x_hat, x_end=initial_condition, is_artificial_step=not lands_on_data)
            if final:
                assert s_next > last, f"Invalid s_next: {s_next} (should be <= {last})"
                x_next = ops.concat_channels([initial_condition[:, :1], x_hat]) if hp.hack_for_imprecise_interpolation \
                    else x_hat
            else:
                assert s_next <= last, f"Invalid s_next: {s_next} (should be <= {last})"
                if cold and s > 0 and x_s.shape[0] <= self.fuse_interpolator_pair_max_batch and self._can_stack_calls():
                    # the step's two interpolations (to s_next, then to s: the reference's call order) as one 2B forward
                    x_next, x_at_s = self.q_sample_pair(x_hat, initial_condition, s_next, s,
                                                        is_artificial_step=not lands_on_data, **dict(kwargs))
                elif cold and s > 0 and self._can_reuse_encoder(kwargs):
                    # larger batches: two forwards, the second restarting behind the first one's encoder
                    x_next, x_at_s = self.q_sample_two(x_hat, initial_condition, s_next, s,
                                                       is_artificial_step=not lands_on_data, **dict(kwargs))
                else:
                    x_next = self.q_sample(**ipol, t=s_next, **dict(kwargs))
            if not cold:
                x_s = x_next
            elif final and not hp.use_cold_sampling_for_last_step:
                if self.use_cold_sampling_for_init_of_ar_step:       # cold state only seeds the next AR window (:503-510)
                    ar_init = ops.cold_update(x_s, x_hat, self.q_sample(**ipol, t=s, **dict(kwargs)))
                    preds["preds_autoregressive_init"] = self._drop_carried_channel(ar_init)
                x_s = x_hat
            else:
                # x_s + (x_next - I(x_0, x_hat, s)); at s = 0 the interpolation "at time 0" is x_s itself
                if s > 0 and x_at_s is None:
                    x_at_s = self.q_sample(**ipol, t=s, **dict(kwargs))
                x_s = ops.cold_update(x_s, x_next, x_at_s)
            k = int(i_next) if not final else k + 1
            if lands_on_data:
                emit = x_s if (hp.use_cold_sampling_for_intermediate_steps or final) else x_next
                preds[f"t{k}_preds"] = self._drop_carried_channel(emit)
        if hp.refine_intermediate_predictions:
            # a second sweep (dyffusion.py:551-563): every intermediate data time is re-interpolated between x_0 and the LAST
            # forecast; the final time keeps the forecast itself
            times = hp.prediction_timesteps or list(self.dynamical_steps.values())
            for i_n in (t for t in times if t < self.num_timesteps):
                label = int(i_n) if float(i_n).is_integer() else i_n
                assert not float(i_n).is_integer() or f"t{label}_preds" in preds, f"t{label}_preds not in intermediates"
                again = self.q_sample(x0=x_hat, x_end=initial_condition, t=None, interpolation_time=i_n,
                                      is_artificial_step=False, **dict(kwargs))
                preds[f"t{label}_preds"] = self._drop_carried_channel(again)
        return (x_hat if sched[-1] + 1 >= self.num_timesteps else x_s), preds

    @torch.inference_mode()
    def sample(self, initial_condition, num_samples=1, **kwargs) -> Dict[str, Tensor]:
        _, intermediates = self.sample_loop(initial_condition, **kwargs)
        return intermediates

    def predict_forward(self, *inputs, metadata: Any = None, **kwargs):
        assert len(inputs) == 1, "Only one input tensor is allowed for the forward pass"
        return self.sample(inputs[0], **kwargs)
