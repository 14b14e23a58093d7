"""Window / member driver around the stepper: host mirror of `run_inference` + `WindowStitcher`
(`src/ace_inference/inference/loop.py:26-117,158-264`).

Same call (`run_inference(aggregator, stepper, data, n_forward_steps, forward_steps_in_memory, n_ensemble_members,
eval_device, writer)`), same objects on the other side (`writer.append_batch(target=, prediction=, start_timestep=,
start_sample=, batch_times=)`, `aggregator.record_batch(loss=, target_data=, gen_data=, target_data_norm=,
gen_data_norm=, i_time_start=)`), same stitching rules (the first time of every window but the first is dropped, the
last generated state of a member is the next window's initial condition of that member, members are stacked on a
leading axis) and the same `timers` dictionary.

MI355X-first differences:
* ensemble members are BATCHED, not looped: a window runs once with batch = samples x members, IC-major (the reference
  calls `run_on_batch` once per member, `loop.py:199-208`); trajectory (initial condition ic, member m) has the GLOBAL
  index `ic * members + m` (`ensemble.rank_units`) and draws the dropout stream of that index (`batch_offset` + row in
  the C ABI), so its results do not depend on how initial conditions / members are grouped or sharded over GPUs;
* the carried state never leaves the device (the reference moves it to the CPU and back every window, `loop.py:78-83,115`);
* writer / aggregator receive device tensors; `host_outputs=True` hands the writer pinned host copies made on a side
  stream while the next window computes.
Derived variables (`compute_derived_quantities`) are not on the sampling path: pass `derive=` to apply your own.
"""
from __future__ import annotations

import time
from collections import defaultdict
from typing import Callable, Dict, Mapping, Optional, Tuple

import torch

from .ensemble import plan_rows
from .stepper import SteppedData


class NullDataWriter:
    """`src/ace_inference/inference/data_writer/main.py:171-187`"""

    def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
        pass

    def flush(self):
        pass


class NullAggregator:
    """`src/ace_inference/core/aggregator/null.py`"""

    def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start=0):
        pass

    def get_logs(self, label: str):
        return {}


class _DeferredHostWriter:
    """Wraps a writer: device tensors of window i are copied to pinned host memory on a side stream and handed to the
    wrapped writer when window i + 1 arrives (or at flush), so the D2H transfer overlaps the next window's compute."""

    def __init__(self, writer, device):
        self.writer, self.device = writer, device
        self.stream = torch.cuda.Stream(device=device)
        self.pending = None

    def _host(self, d):
        return {k: torch.empty(v.shape, dtype=v.dtype, device="cpu", pin_memory=True).copy_(v, non_blocking=True)
                for k, v in d.items()}

    def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
        self.flush_pending()
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            ht, hp = self._host(target), self._host(prediction)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        for v in list(target.values()) + list(prediction.values()):
            v.record_stream(self.stream)
        self.pending = (ev, ht, hp, start_timestep, start_sample, batch_times)

    def flush_pending(self):
        if self.pending is not None:
            ev, ht, hp, st, ss, bt = self.pending
            ev.synchronize()
            self.writer.append_batch(target=ht, prediction=hp, start_timestep=st, start_sample=ss, batch_times=bt)
            self.pending = None

    def flush(self):
        self.flush_pending()
        if hasattr(self.writer, "flush"):
            self.writer.flush()


class WindowStitcher:
    """`src/ace_inference/inference/loop.py:26-117` (first time of every window but the first dropped, last state carried
    into the next window: targets per initial condition, generated variables per trajectory), with the carried state
    resident on the device and kept FLAT: one row per trajectory of the device batch, whatever the ensemble geometry."""

    def __init__(self, n_forward_steps: int, writer, is_ensemble: bool = False):
        self.i_time = 0
        self.n_forward_steps = n_forward_steps
        self.writer = writer
        self.is_ensemble = is_ensemble
        self._carry_target: Optional[Dict[str, torch.Tensor]] = None     # name -> (n_sample, H, W)
        self._carry_gen: Optional[Dict[str, torch.Tensor]] = None        # name -> (rows, H, W)

    def append(self, data: Mapping[str, torch.Tensor], gen_data: Mapping[str, torch.Tensor], batch_times=None,
               last_state: Optional[Mapping[str, torch.Tensor]] = None, start_sample: int = 0) -> None:
        """`last_state`: name -> (rows, H, W), the last generated state of every trajectory of the device batch."""
        n_time = next(iter(data.values())).shape[1]
        self.writer.append_batch(target=data, prediction=gen_data, start_timestep=self.i_time, start_sample=start_sample,
                                 batch_times=batch_times)
        self.i_time += n_time
        if self.i_time < self.n_forward_steps:      # only store if needed
            self._carry_target = {k: v[:, -1].detach().clone() for k, v in data.items()}
            src = last_state if last_state is not None else {k: v[..., -1, :, :].reshape(-1, *v.shape[-2:])
                                                             for k, v in gen_data.items()}
            self._carry_gen = {k: v.detach().clone() for k, v in src.items()}

    def apply_initial_condition(self, batch: Mapping[str, torch.Tensor], ic_rows: Optional[torch.Tensor] = None) -> None:
        """`batch` tensors are (rows, n_time, H, W), one row per trajectory; `ic_rows[r]` = initial condition (row of the
        window data) trajectory r belongs to.  The first time of every variable becomes the state carried from the previous
        window: per trajectory for the generated variables, per initial condition for everything else."""
        if self.i_time > self.n_forward_steps:
            raise ValueError("Cannot apply initial condition after the last segment has been appended, currently at "
                             f"time index {self.i_time} with {self.n_forward_steps} max forward steps.")
        if self._carry_target is None:
            return
        for k, v in batch.items():
            if k in self._carry_gen:
                v[:, 0] = self._carry_gen[k].to(v.device)
            else:
                ic = self._carry_target[k].to(v.device)
                v[:, 0] = ic if ic_rows is None else ic.index_select(0, ic_rows)


def _remove_ic(d: Mapping[str, torch.Tensor], ensemble: bool) -> Dict[str, torch.Tensor]:
    return {k: (v[:, :, 1:] if ensemble else v[:, 1:]) for k, v in d.items()}


def run_inference(aggregator, stepper, data, n_forward_steps: int, forward_steps_in_memory: int,
                  n_ensemble_members: int = 1, eval_device=None, writer=None, derive: Optional[Callable] = None,
                  host_outputs: bool = False, trajectory_offset: int = 0,
                  unit_range: Optional[Tuple[int, int]] = None) -> Dict[str, float]:
    """`data`: an object with `.loader` (iterable of windows with `.data`: name -> (n_sample, steps + 1, H, W) and
    `.times`) or such an iterable.

    Multi-GPU (one process per GPU, no collective on the data path).  A job is (initial conditions) x (members); its
    trajectories are numbered globally `u = ic * n_ensemble_members + member` (`ensemble.rank_units`), and trajectory u
    draws dropout stream u whatever the sharding.
      * `trajectory_offset`: global index of the first initial condition the windows hold (the reference shards whole
        initial conditions over ranks, `data_loading/inference.py:110-113`);
      * `unit_range=(start, count)`: run only global trajectories start ... start + count - 1 (`ensemble.shard`) - members of
        ONE initial condition split over ranks, or any ragged share of ICs x members.  Writer / aggregator then receive
        flat `(count, time, H, W)` predictions with `start_sample=start`, targets for the initial conditions touched.
    Without `unit_range` every member of every initial condition present runs and predictions are presented as the
    reference stacks them: `(members, n_sample, time, H, W)`."""
    writer = writer if writer is not None else NullDataWriter()
    aggregator = aggregator if aggregator is not None else NullAggregator()
    members = int(n_ensemble_members)
    ens = members > 1
    dev = torch.device(eval_device) if eval_device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("sdy_amd.run_inference runs on the GPU only (no CPU fallback)")
    if host_outputs:
        writer = _DeferredHostWriter(writer, dev)
    stitcher = WindowStitcher(n_forward_steps, writer, is_ensemble=ens)
    loader = data.loader if hasattr(data, "loader") else data
    timers: Dict[str, float] = defaultdict(float)
    now = time.time()
    module = stepper.module
    n_rows = 0
    for i, window in enumerate(loader):
        timers["data_loading"] += time.time() - now
        now = time.time()
        i_time = i * forward_steps_in_memory
        win = {k: v.to(dev, torch.float32, non_blocking=True) for k, v in window.data.items()}
        n_sample = next(iter(win.values())).shape[0]
        start, n_rows, ic_list, _, rect = plan_rows(n_sample, members, trajectory_offset, unit_range)
        ic_rows = torch.tensor(ic_list, dtype=torch.long, device=dev)
        # IC-major batch: row r is global trajectory start + r = (IC ic_rows[r], member (start + r) % members)
        batch = {k: v.index_select(0, ic_rows) for k, v in win.items()}
        # the stitcher carries targets for the initial conditions this process touches (all of them unless ragged)
        stitcher.apply_initial_condition(batch, ic_rows if rect else ic_rows - ic_list[0])
        if hasattr(module, "set_batch_offset"):
            module.set_batch_offset(start)
        stepped = stepper.run_on_batch(batch, None, n_forward_steps=forward_steps_in_memory)
        last_state = {k: v[:, -1] for k, v in stepped.gen_data.items()}
        if rect:       # present like the reference: members on a leading axis (a strided view, no copy)
            ics = slice(None)
            unfold = (lambda d: {k: v.view(n_sample, members, *v.shape[1:]).transpose(0, 1) for k, v in d.items()}) \
                if ens else (lambda d: d)
            first = slice(0, n_rows, members)
        else:          # a ragged share: flat rows, targets of the initial conditions touched
            ics = slice(ic_list[0], ic_list[-1] + 1)
            unfold = lambda d: d  # noqa: E731
            first = torch.tensor([ic_list.index(c) for c in range(ic_list[0], ic_list[-1] + 1)], device=dev)
        flat = not rect
        win = {k: v[ics] for k, v in win.items()}
        target_data = derive(win) if derive is not None else win
        gen_data, gen_norm = unfold(stepped.gen_data), unfold(stepped.gen_data_norm)
        if derive is not None:
            gen_data = derive(gen_data)
        tgt_norm = {k: v[first] for k, v in stepped.target_data_norm.items()}
        out = SteppedData(metrics=stepped.metrics, gen_data=gen_data, target_data=target_data, gen_data_norm=gen_norm,
                          target_data_norm=tgt_norm)
        torch.cuda.current_stream(dev).synchronize()
        timers["run_on_batch"] += time.time() - now
        now = time.time()
        # ---- _inference_internal_loop (loop.py:120-153)
        times = window.times
        stacked = ens and not flat
        if i_time > 0:
            out = SteppedData(metrics=out.metrics, gen_data=_remove_ic(out.gen_data, stacked),
                              target_data={k: v[:, 1:] for k, v in out.target_data.items()},
                              gen_data_norm=_remove_ic(out.gen_data_norm, stacked),
                              target_data_norm={k: v[:, 1:] for k, v in out.target_data_norm.items()})
            if times is not None and hasattr(times, "isel"):
                times = times.isel(time=slice(1, None))
            i_time_agg = i_time + 1
        else:
            i_time_agg = i_time
        stitcher.append(out.target_data, out.gen_data, times, last_state=last_state, start_sample=start if flat else 0)
        aggregator.record_batch(loss=float(out.metrics["loss"]), target_data=out.target_data, gen_data=out.gen_data,
                                target_data_norm=out.target_data_norm, gen_data_norm=out.gen_data_norm,
                                i_time_start=i_time_agg)
        del stepped, out
        timers["writer_and_aggregator"] += time.time() - now
        now = time.time()
    if hasattr(writer, "flush"):
        writer.flush()
    total = timers["run_on_batch"]
    if total > 0:
        # the reference logs n_forward_steps x n_ICs per second (inference.py:294-298); here: x trajectories of this process
        timers["forecast_steps_per_second"] = stitcher.i_time * max(n_rows, 1) / total
    for name, duration in timers.items():
        print(f"{name}: {duration:.2f}" + ("" if name.endswith("per_second") else "s"))
    return dict(timers)
