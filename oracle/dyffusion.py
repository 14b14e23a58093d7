"""Oracle: restatement of the DYffusion cold-sampling loop (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `src/diffusion/dyffusion.py`:
  * `BaseDYffusion.__init__` schedule maths                      :21-128
  * `diffusion_step_to_interpolation_step` (before_t1_only)       :134-188
  * `q_sample` / `DYffusion._interpolate`                         :190-240, :642-662
  * `predict_x_last` / `_predict_last_dynamics`                   :286-355
  * `sampling_schedule` setter (the named schedules)              :367-455
  * `sample_loop` / `sample`                                      :457-572
and `InterpolationExperiment.get_dynamical_condition` (`src/experiment_types/interpolation.py:133-141`).

`forecaster(x, time, **cond)` and `interpolator(x, time, **cond)` are callables (e.g. OracleSFNO.forward
wrapped with a mask provider).  Only what the shipped configs exercise is restated
(`schedule="before_t1_only"`, `forward_conditioning="none"`, `time_encoding="dynamics"`).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch


class OracleDYffusion:
    def __init__(
        self,
        forecaster: Callable,
        interpolator: Callable,
        timesteps: int,
        additional_interpolation_steps: int = 0,
        sampling_type: str = "cold",
        use_cold_sampling_for_intermediate_steps: bool = True,
        use_cold_sampling_for_last_step: bool = True,
        use_cold_sampling_for_init_of_ar_step: Optional[bool] = None,
        refine_intermediate_predictions: bool = False,
        hack_for_imprecise_interpolation: bool = False,
        dynamic_cond_from_t: str = "h",
        sampling_schedule=None,
        enable_interpolator_dropout=True,
    ):
        self.forecaster, self.interpolator = forecaster, interpolator
        self.horizon = timesteps
        self.k = additional_interpolation_steps
        self.num_timesteps = timesteps + self.k
        self.sampling_type = sampling_type
        self.cold_intermediate = use_cold_sampling_for_intermediate_steps
        self.cold_last = use_cold_sampling_for_last_step
        self.cold_ar_init = use_cold_sampling_for_last_step if use_cold_sampling_for_init_of_ar_step is None \
            else use_cold_sampling_for_init_of_ar_step
        self.refine = refine_intermediate_predictions
        self.hack = hack_for_imprecise_interpolation
        self.dynamic_cond_from_t = dynamic_cond_from_t
        d_to_i = {d: self.d2i(d) for d in range(1, self.num_timesteps)}
        self.dynamical_steps = {d: i for d, i in d_to_i.items() if float(i).is_integer()}
        self.artificial_steps = {d: i for d, i in d_to_i.items() if not float(i).is_integer()}
        self.enable_interpolator_dropout = enable_interpolator_dropout
        # q_sample's per-call dropout rule (:226-230), published for the interpolator callable of the tests: a callable that
        # replays recorded masks consumes a mask set only when the reference's network drew one
        self.dropout_on = None
        self.sampling_schedule = self.schedule_from(sampling_schedule or list(range(self.num_timesteps)))

    def schedule_from(self, schedule):
        """dyffusion.py:367-455: a list, or one of the names `only_dynamics[_plus[_discrete]N]`, `everyN[th|nd|rd]`,
        `firstN` / `first0.x`; every named schedule keeps step 0 and the steps that land on data times."""
        name = schedule
        if isinstance(name, str):
            base = [0] + list(self.dynamical_steps.keys())
            art = list(self.artificial_steps.keys())
            if "only_dynamics" in name:
                schedule = []
                if "only_dynamics_plus" in name:
                    n = int(name.replace("only_dynamics_plus", "").replace("_discrete", ""))
                    schedule = list(np.linspace(0, base[1], n + 1, endpoint=False))
                    if "_discrete" in name:
                        schedule = [int(np.floor(v)) for v in schedule]
                else:
                    assert name == "only_dynamics"
            elif name.startswith("every"):
                n = int(name.replace("every", "").replace("th", "").replace("nd", "").replace("rd", ""))
                assert 1 <= n <= self.num_timesteps
                schedule = art[::n]
            elif name.startswith("first"):
                n = float(name.replace("first", "").replace("v2", ""))
                if n < 1:
                    assert 0 < n
                    n = int(np.ceil(n * len(art)))
                else:
                    assert n.is_integer() and 1 <= n <= self.num_timesteps
                    n = int(n)
                schedule = art[:n]
            else:
                raise ValueError(name)
            schedule = sorted(set(list(schedule) + base))
        schedule = list(schedule)
        assert 1 <= schedule[-1] <= self.num_timesteps
        if schedule[0] != 0:
            schedule = [0] + schedule
        assert all(b > a for a, b in zip(schedule, schedule[1:]))
        if all(float(v).is_integer() for v in schedule):
            schedule = [int(v) for v in schedule]
        return schedule

    def d2i(self, d):
        assert 0 <= d <= self.num_timesteps - 1
        if d >= self.k + 1:
            return d - self.k
        return d / (self.k + 1)

    # -- network calls ----------------------------------------------------------------------
    def predict_x_last(self, x0, x_t, t, **kwargs):
        B = x0.shape[0]
        time = torch.full((B,), float(self.d2i(t)), dtype=torch.float32)
        dyn = kwargs.pop("dynamical_condition", None)
        cond = None
        if dyn is not None:
            assert dyn.shape[1] == self.num_timesteps + 1
            cond = {"0": dyn[:, 0], "h": dyn[:, -1]}[self.dynamic_cond_from_t]
        return self.forecaster(x_t, time=time, condition=cond, **kwargs)

    def q_sample(self, x0, x_end, t, interpolation_time=None, is_artificial_step=True, **kwargs):
        i_n = interpolation_time if t is None else self.d2i(t)
        self.dropout_on = bool(self.enable_interpolator_dropout in (True, "always") or
                               (self.enable_interpolator_dropout == "except_dynamical_steps" and is_artificial_step))
        dyn = kwargs.pop("dynamical_condition", None)
        if dyn is not None:
            assert isinstance(i_n, (int, np.integer))
            kwargs["condition"] = dyn[:, i_n]
        B = x0.shape[0]
        time = torch.full((B,), float(i_n), dtype=torch.float32)
        assert 0 < i_n < self.horizon
        x_last = torch.cat([x_end[:, :1], x0], dim=1) if self.hack else x0
        out = self.interpolator(torch.cat([x_end, x_last], dim=1), time=time, **kwargs)
        if self.hack:
            out = torch.cat([x_end[:, :1], out], dim=1)
        return out

    # -- sampler ----------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, initial_condition, **kwargs) -> Dict[str, torch.Tensor]:
        sched = self.sampling_schedule
        N = self.num_timesteps
        last_p1 = sched[-1] + 1
        triples = zip(sched, sched[1:] + [last_p1], sched[2:] + [last_p1, last_p1 + 1])
        out: Dict[str, torch.Tensor] = {}
        x_s, dyn_step, xhat = initial_condition, 0, None
        for s, s_next, s_nnext in triples:
            is_last = s == N - 1
            xhat = self.predict_x_last(initial_condition, x_s, s, **dict(kwargs))
            time_i_n = self.d2i(s_next) if not is_last else np.inf
            is_dyn = float(time_i_n).is_integer() or is_last
            art = dict(is_artificial_step=not is_dyn)
            if s_next <= N - 1:
                x_ip_next = self.q_sample(xhat, initial_condition, s_next, **art, **dict(kwargs))
            else:
                x_ip_next = xhat
                if self.hack:
                    x_ip_next = torch.cat([initial_condition[:, :1], x_ip_next], dim=1)
            if self.sampling_type == "cold":
                if not self.cold_last and is_last:
                    if self.cold_ar_init:
                        x_ip_s = self.q_sample(xhat, initial_condition, s, **art, **dict(kwargs))
                        ar = x_s + xhat - x_ip_s
                        out["preds_autoregressive_init"] = ar[:, 1:] if self.hack else ar
                    x_s = xhat
                else:
                    x_ip_s = self.q_sample(xhat, initial_condition, s, **art, **dict(kwargs)) if s > 0 else x_s
                    x_s = x_s + (x_ip_next - x_ip_s)
            elif self.sampling_type == "naive":
                x_s = x_ip_next
            else:
                raise ValueError(self.sampling_type)
            dyn_step = int(time_i_n) if s < N - 1 else dyn_step + 1
            if is_dyn:
                preds = x_s if (self.cold_intermediate or is_last) else x_ip_next
                out[f"t{dyn_step}_preds"] = preds[:, 1:] if self.hack else preds
        if self.refine:
            for i_n in [i for i in self.dynamical_steps.values() if i < N]:
                r = self.q_sample(xhat, initial_condition, None, interpolation_time=int(i_n), is_artificial_step=False,
                                  **dict(kwargs))
                out[f"t{int(i_n)}_preds"] = r[:, 1:] if self.hack else r
        return out
