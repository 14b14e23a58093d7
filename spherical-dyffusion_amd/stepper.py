"""Autoregressive stepper around the sampler: host mirror of `MultiStepStepper.run_on_batch` /
`run_on_batch_multistep` (`src/ace_inference/core/stepper_multistep.py:149-193,298-466`).

Same call (`run_on_batch(data, optimization, n_forward_steps, aggregator)`), same `SteppedData` result
(`metrics`, `gen_data`, `target_data`, `gen_data_norm`, `target_data_norm`; generated timelines include the initial
condition as their first time step), same semantics of `Packer`, `StandardNormalizer`, `Prescriber`, the
`hack_for_imprecise_interpolation` channel handling and the relative `LpLoss` metrics.

MI355X-first differences: every variable stays on the device for the whole window (the reference moves each generated
step to the CPU, `stepper_multistep.py:410`); normalise+pack, prescriber+unpack+denormalise+AR-feedback and the loss
terms are one HIP launch each per step (`sdy_norm_pack`, `sdy_step_finish`, `sdy_lp_rel_terms`); the only host sync is
one read of the loss terms at the end of the window -- and with `defer_metrics=True` not even that: the terms and the
device's sticky status word travel to pinned host memory behind the window's last launch and `SteppedData.metrics`
waits for them on first access (the window driver reads them while the NEXT window computes).  torch owns the buffers;
no arithmetic runs in torch.
"""
from __future__ import annotations

import ctypes as C
from collections.abc import Mapping
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from ._lib import SDY_MAX_VARS, SdyStepFinishArgs, SdyVarTable, check, current_stream, lib, ptr


@dataclass
class Prescriber:
    """`src/ace_inference/core/prescriber.py:52-92`"""
    prescribed_name: str
    mask_name: str
    mask_value: int
    interpolate: bool = False

    def __post_init__(self):
        if self.interpolate and self.mask_value != 1:
            raise ValueError(f"Interpolation requires mask_value to be 1, but it is set to {self.mask_value}.")


class WindowMetrics(Mapping):
    """`SteppedData.metrics` of a window whose LpLoss terms (and status word) are still on their way to the host: a read-only
    mapping `loss`, `loss_step_<i>` (LpLoss.rel, `src/ace_inference/training/utils/darcy_loss.py:214-228`) that waits for the
    copy's event on first access, then raises if a kernel of the window flagged an fp16-range overflow / non-finite
    statistics (include/sdy_amd.h, `sdy_status_flags`)."""

    def __init__(self, terms_host: torch.Tensor, flags_host: Optional[torch.Tensor], event, where: str):
        self._terms, self._flags, self._event, self._where = terms_host, flags_host, event, where
        self._values: Optional[Dict[str, torch.Tensor]] = None

    def ready(self) -> bool:
        return self._values is not None or self._event.query()

    def _resolve(self) -> Dict[str, torch.Tensor]:
        if self._values is None:
            self._event.synchronize()
            if self._flags is not None:
                from . import ops
                flags = int(self._flags[0])
                # a NaN slips through the kernels' max-based range guards (fmax ignores NaN) and the last block's MLP output
                # meets no InstanceNorm: the prediction error read back here is the last line of defence
                if not bool(torch.isfinite(self._terms[..., 0]).all()) and bool(torch.isfinite(self._terms[..., 1]).all()):
                    flags |= ops.FLAG_NONFINITE
                ops.raise_on_status_flags(flags, self._where)
            terms = self._terms
            per_step = (terms[..., 0].sqrt() / terms[..., 1].sqrt()).mean(dim=1)
            vals = {f"loss_step_{i}": per_step[i].to(torch.float32) for i in range(per_step.shape[0])}
            vals["loss"] = per_step.sum().to(torch.float32)
            self._values = vals
        return self._values

    def __getitem__(self, key):
        return self._resolve()[key]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return len(self._resolve())


@dataclass
class SteppedData:
    """`src/ace_inference/core/stepper.py:178-183`"""
    metrics: Dict[str, torch.Tensor]
    gen_data: Dict[str, torch.Tensor]
    target_data: Dict[str, torch.Tensor]
    gen_data_norm: Dict[str, torch.Tensor]
    target_data_norm: Dict[str, torch.Tensor]


class MultiStepStepper:
    def __init__(self, module, in_names: List[str], out_names: List[str], forcing_names: List[str],
                 means: Dict[str, float], stds: Dict[str, float], prescriber: Optional[Prescriber] = None):
        self.module = module
        # init_packers (stepper_multistep.py:219-223): the in packer excludes the forcings
        self.in_names = [n for n in in_names if n not in forcing_names]
        self.out_names = list(out_names)
        self.forcing_names = list(forcing_names)
        self.means = {k: float(v) for k, v in means.items()}
        self.stds = {k: float(v) for k, v in stds.items()}
        if prescriber is not None and not (prescriber.prescribed_name in self.in_names
                                           and prescriber.prescribed_name in self.out_names):
            raise ValueError("Variables which are being prescribed in masked regions must be in in_names and out_names, "
                             f"but {prescriber.prescribed_name} is not.")     # prescriber.py:39-43
        self.prescriber = prescriber
        # once per window: raise if a kernel flagged an fp16-range overflow / non-finite statistics (include/sdy_amd.h)
        self.check_status = True
        if max(len(self.in_names), len(self.out_names), len(self.forcing_names)) > SDY_MAX_VARS:
            raise ValueError(f"more than {SDY_MAX_VARS} variables")
        # one entry per distinct variable of (in packer) U (out packer)
        self._entries = list(dict.fromkeys(self.in_names + self.out_names))

    @classmethod
    def from_statistics(cls, module, in_names: List[str], out_names: List[str], forcing_names: Optional[List[str]] = None,
                        data_dir_stats=None, data_dir=None, prescriber: Optional[Prescriber] = None) -> "MultiStepStepper":
        """The reference's constructor path (`stepper_multistep.py:103-131`): forcings default to the input-only names,
        the scalars of `normalize_names` = in U out come from `centering` / `scaling` found in `data_dir_stats`, `data_dir`
        or the packaged statistics (`normalizer.find_statistics`)."""
        from .normalizer import find_statistics, get_normalizer
        if forcing_names is None:
            forcing_names = [n for n in in_names if n not in out_names]
        path_mean, path_std = find_statistics(data_dir_stats, data_dir)
        norm = get_normalizer(path_mean, path_std, list(dict.fromkeys(list(in_names) + list(out_names))))
        return cls(module, in_names, out_names, forcing_names, norm.means, norm.stds, prescriber)

    # ---- helpers ------------------------------------------------------------------------------------------------
    def _table(self, names: List[str], data: Dict[str, torch.Tensor]) -> SdyVarTable:
        t = SdyVarTable()
        t.nvars = len(names)
        for i, n in enumerate(names):
            t.data[i] = ptr(data[n])
            t.mean[i] = self.means.get(n, 0.0) if n in self.means else 0.0
            t.std[i] = self.stds.get(n, 1.0) if n in self.means else 1.0
        return t

    # ---- run_on_batch -------------------------------------------------------------------------------------------------
    def run_on_batch(self, data: Dict[str, torch.Tensor], optimization=None, n_forward_steps: int = 1,
                     aggregator=None, defer_metrics: bool = False) -> SteppedData:
        """`defer_metrics=True` (not in the reference): return without draining the stream; `metrics` then waits for the
        window's loss terms on first access (`WindowMetrics`)."""
        any_t = next(iter(data.values()))
        assert any_t.dim() == 4, "expected (n_sample, n_timesteps, n_lat, n_lon) per variable"
        B, T1, H, W = any_t.shape
        assert T1 == n_forward_steps + 1, f"{T1=} must be n_forward_steps + 1"      # stepper_multistep.py:347
        if not any_t.is_cuda:
            raise RuntimeError("sdy_amd stepper runs on the GPU only (no CPU fallback); move the window to cuda")
        dev = any_t.device
        HW = H * W
        data = {k: v.to(dev, torch.float32).contiguous() for k, v in data.items()}
        mod = self.module
        horizon = mod.true_horizon
        hack = bool(getattr(getattr(mod.model, "hparams", None), "hack_for_imprecise_interpolation", False))
        stream = current_stream

        n_in, n_out, n_f = len(self.in_names), len(self.out_names), len(self.forcing_names)
        in_tab, out_tab = self._table(self.in_names, data), self._table(self.out_names, data)
        f_tab = self._table(self.forcing_names, data) if n_f else None
        # generated timelines (normalised / denormalised), slot 0 = initial condition
        gen_norm = {n: torch.empty(B, T1, H, W, dtype=torch.float32, device=dev) for n in self.out_names}
        gen = {n: torch.empty(B, T1, H, W, dtype=torch.float32, device=dev) for n in self.out_names}
        with torch.cuda.device(dev):
            tln = (C.c_void_p * n_out)(*[ptr(gen_norm[n]) for n in self.out_names])
            tld = (C.c_void_p * n_out)(*[ptr(gen[n]) for n in self.out_names])
            check(lib.sdy_init_timeline(C.byref(out_tab), T1, B, HW, tln, tld, stream()), "sdy_init_timeline")
            state = torch.empty(B, n_in, H, W, dtype=torch.float32, device=dev)
            check(lib.sdy_norm_pack(C.byref(in_tab), 0, T1, B, HW, ptr(state), stream()), "sdy_norm_pack")
            loss_terms = torch.zeros(n_forward_steps, B, 2, dtype=torch.float64, device=dev)

            fa = SdyStepFinishArgs()
            fa.B, fa.HW, fa.T1 = B, HW, T1
            fa.n_out, fa.n_in, fa.n_entries = n_out, n_in, len(self._entries)
            fa.presc_entry = -1
            for e, name in enumerate(self._entries):
                fa.out_idx[e] = self.out_names.index(name) if name in self.out_names else -1
                fa.in_idx[e] = self.in_names.index(name) if name in self.in_names else -1
                if name in self.out_names:
                    fa.gen_norm_tl[e], fa.gen_tl[e] = ptr(gen_norm[name]), ptr(gen[name])
                fa.mean[e] = self.means[name] if name in self.means else 0.0
                fa.std[e] = self.stds[name] if name in self.means else 1.0
                if hack and fa.out_idx[e] < 0 and name != "HGTsfc":
                    raise ValueError("hack_for_imprecise_interpolation carries exactly the input-only variable 'HGTsfc' "
                                     "(stepper_multistep.py:421-422)")
            if self.prescriber is not None:
                p = self.prescriber
                fa.presc_entry = self._entries.index(p.prescribed_name)
                fa.presc_target, fa.presc_mask = ptr(data[p.prescribed_name]), ptr(data[p.mask_name])
                fa.mask_value, fa.interpolate = int(p.mask_value), int(p.interpolate)
            if not hack and any(fa.out_idx[e] < 0 for e in range(len(self._entries))):
                raise ValueError("every model input must be produced by the model unless hack_for_imprecise_interpolation "
                                 "is set (the reference would fail with a KeyError when re-packing the inputs)")

            forcing = None
            if n_f:
                forcing = torch.empty(B, n_f, H, W, dtype=torch.float32, device=dev)
                check(lib.sdy_norm_pack(C.byref(f_tab), 0, T1, B, HW, ptr(forcing), stream()), "sdy_norm_pack")
            for th in range(1, n_forward_steps + 1):
                h = th % horizon or horizon                                         # stepper_multistep.py:370-372
                batch = {"dynamics": state}
                if hack:
                    batch["static_condition"] = forcing                              # :383-384
                with mod.ema_scope(), mod.inference_dropout_scope():
                    res = mod.get_preds_at_t_for_batch(batch, horizon=h, split="predict", ensemble=False,
                                                       is_autoregressive=th > horizon, prepare_inputs=False,
                                                       num_predictions=1)
                g = res[f"t{h}_preds_normed"].contiguous()
                check(lib.sdy_lp_rel_terms(ptr(g), C.byref(out_tab), th, T1, B, HW, ptr(loss_terms[th - 1]), stream()),
                      "sdy_lp_rel_terms")
                ar = None
                if "preds_autoregressive_init_normed" in res:   # the state handed to the next window differs from the
                    ar = res["preds_autoregressive_init_normed"].contiguous()   # prediction (stepper_multistep.py:412-418)
                nxt = torch.empty_like(state)
                fa.t, fa.gen, fa.prev_in, fa.next_in = th, ptr(g), ptr(state), ptr(nxt)
                fa.ar_init = ptr(ar)
                check(lib.sdy_step_finish(C.byref(fa), stream()), "sdy_step_finish")
                state = nxt
                if n_f:
                    forcing = torch.empty(B, n_f, H, W, dtype=torch.float32, device=dev)
                    check(lib.sdy_norm_pack(C.byref(f_tab), th, T1, B, HW, ptr(forcing), stream()), "sdy_norm_pack")

            # metrics (LpLoss.rel, darcy_loss.py:214-228): one device->host copy for the whole window, into pinned memory
            # behind the window's last launch, with the sticky status word (4 bytes) right behind it
            terms_host = torch.empty(loss_terms.shape, dtype=torch.float64, pin_memory=True)
            terms_host.copy_(loss_terms, non_blocking=True)
            flags_host = None
            if self.check_status:
                flags_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
                check(lib.sdy_status_flags_async(ptr(flags_host), 1, stream()), "sdy_status_flags_async")
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
        metrics = WindowMetrics(terms_host, flags_host, done, "MultiStepStepper.run_on_batch")
        # normalised targets for the caller (full_data_norm of the reference): one launch per variable, the (B, T1) axes
        # flattened into the batch axis of the same kernel
        target_norm = {}
        with torch.cuda.device(dev):
            for name, v in data.items():
                if name in self.means:
                    out = torch.empty_like(v)
                    tab = self._table([name], {name: v})
                    check(lib.sdy_norm_pack(C.byref(tab), 0, 1, B * T1, HW, ptr(out), stream()), "sdy_norm_pack")
                    target_norm[name] = out
                else:
                    target_norm[name] = v
        if not defer_metrics:
            len(metrics)          # waits for the copy; raises on a flagged window
        if aggregator is not None:
            aggregator.record_batch(float(metrics["loss"]), target_data=data, gen_data=gen, target_data_norm=target_norm,
                                    gen_data_norm=gen_norm)
        return SteppedData(metrics=metrics, gen_data=gen, target_data=data, gen_data_norm=gen_norm,
                           target_data_norm=target_norm)
