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
  stream while the next window computes;
* the loop never drains the device between windows (`prefetch` > 0): window i + 1 is pulled from the loader on a
  background thread, staged in pinned host memory and uploaded on a side stream while window i computes (the reference gets
  the same overlap from DataLoader workers, `src/ace_inference/core/data_loading/getters.py:160-166`); a window's loss terms
  and the device's status word come back asynchronously and are read -- by `aggregator.record_batch(loss=...)` -- one window
  later, while the next one is already running;
* `forecast_steps_per_second` is the reference's "Total steps per second" (`src/ace_inference/inference/inference.py:294-298`:
  steps x trajectories over the WHOLE duration of the call, loading included); the rate over the device time of the windows
  alone is reported beside it (`forecast_steps_per_second_run_on_batch`).
Derived variables (`compute_derived_quantities`) are not on the sampling path: pass `derive=` to apply your own.
"""
from __future__ import annotations

import queue
import threading
import time
from collections import defaultdict
from typing import Callable, Dict, List, Mapping, Optional, Tuple

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
               last_state: Optional[Mapping[str, torch.Tensor]] = None, start_sample: int = 0,
               defer_write: bool = False) -> Optional[Callable[[], None]]:
        """`last_state`: name -> (rows, H, W), the last generated state of every trajectory of the device batch.
        `defer_write=True`: the carry-over is taken now, the hand-off to the writer is RETURNED as a callable instead of
        made -- the window driver calls it once the window's status word has come back clean, so a window a kernel flagged
        (fp16 range / non-finite) never reaches the writer."""
        n_time = next(iter(data.values())).shape[1]
        writer, i_time = self.writer, self.i_time

        def write():
            writer.append_batch(target=data, prediction=gen_data, start_timestep=i_time, start_sample=start_sample,
                                batch_times=batch_times)

        if not defer_write:
            write()
        self.i_time += n_time
        if self.i_time < self.n_forward_steps:      # only store if needed
            self._carry_target = {k: v[:, -1].detach().clone() for k, v in data.items()}
            src = last_state if last_state is not None else {k: v[..., -1, :, :].reshape(-1, *v.shape[-2:])
                                                             for k, v in gen_data.items()}
            self._carry_gen = {k: v.detach().clone() for k, v in src.items()}
        return write if defer_write else None

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


class _WindowPrefetcher:
    """Iterates `loader` on a background thread, up to `depth` windows ahead of the consumer: every tensor of a window is
    staged in pinned host memory and uploaded on a side stream, so neither the loader's own work (file reads, decoding --
    here: whatever `next(loader)` costs) nor the host-to-device copy sits between two windows of compute.  Yields
    `(window, device tensors, ready event)`; the consumer makes its stream wait for the event.  The reference overlaps
    loading through DataLoader worker processes (`src/ace_inference/core/data_loading/getters.py:160-166`)."""

    _END = object()

    def __init__(self, loader, dev: torch.device, depth: int):
        self.dev = dev
        self.stream = torch.cuda.Stream(device=dev)
        self.q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._stop = False
        self.thread = threading.Thread(target=self._run, args=(loader,), name="sdy-window-prefetch", daemon=True)
        self.thread.start()

    def _put(self, item) -> bool:
        while not self._stop:
            try:
                self.q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, loader):
        try:
            torch.cuda.set_device(self.dev)
            for window in loader:
                staged = {}
                for k, v in window.data.items():
                    if not v.is_cuda and not v.is_pinned():
                        v = torch.empty(v.shape, dtype=v.dtype, pin_memory=True).copy_(v)
                    staged[k] = v
                with torch.cuda.stream(self.stream):
                    win = {k: v.to(self.dev, torch.float32, non_blocking=True) for k, v in staged.items()}
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                if not self._put((window, win, ev)):
                    return
            self._put(self._END)
        except BaseException as exc:       # handed to the consumer, which re-raises it in its own thread
            self._put(exc)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is self._END:
                return
            if isinstance(item, BaseException):
                raise item
            window, win, ev = item
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(ev)
            for v in win.values():
                v.record_stream(cur)      # allocated on the side stream, used (and released) on the compute stream
            yield window, win

    def close(self, timeout: float = 10.0):
        """Stops the loader thread.  A loader blocked inside `next()` cannot be interrupted: after `timeout` seconds the
        (daemon) thread is abandoned with a warning, so that an exception pending in the caller still propagates."""
        self._stop = True
        deadline = time.time() + timeout
        while self.thread.is_alive():
            try:
                self.q.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
            if time.time() > deadline:
                import warnings
                warnings.warn("sdy_amd.run_inference: the window loader did not return within "
                              f"{timeout:.0f} s of the stop request; abandoning its daemon thread")
                break


def _sync_windows(loader, dev: torch.device):
    for window in loader:
        yield window, {k: v.to(dev, torch.float32, non_blocking=True) for k, v in window.data.items()}


class _ChunkedMetrics(Mapping):
    """`metrics` of a window that ran as several device batches (`max_batch`): the reference's loss is a mean over the batch
    rows (`darcy_loss.py:214-228`), so the window's value is the row-weighted mean of the chunks' values."""

    def __init__(self, parts: List[Tuple[int, Mapping[str, torch.Tensor]]]):
        self._parts, self._values = parts, None

    def _resolve(self):
        if self._values is None:
            n = float(sum(r for r, _ in self._parts))
            keys = list(self._parts[0][1].keys())
            self._values = {k: sum(m[k] * (r / n) for r, m in self._parts) for k in keys}
        return self._values

    def __getitem__(self, key):
        return self._resolve()[key]

    def __iter__(self):
        return iter(self._resolve())

    def __len__(self):
        return len(self._resolve())


def run_inference(aggregator, stepper, data, n_forward_steps: int, forward_steps_in_memory: int,
                  n_ensemble_members: int = 1, eval_device=None, writer=None, derive: Optional[Callable] = None,
                  host_outputs: bool = False, trajectory_offset: int = 0,
                  unit_range: Optional[Tuple[int, int]] = None, prefetch: int = 2,
                  max_batch: Optional[int] = None, relay=None, relay_comm=None) -> Dict[str, float]:
    """`data`: an object with `.loader` (iterable of windows with `.data`: name -> (n_sample, steps + 1, H, W) and
    `.times`) or such an iterable.

    Multi-GPU (one process per GPU, no collective on the data path).  A job is (initial conditions) x (members); its
    trajectories are numbered globally `u = ic * n_ensemble_members + member` (`ensemble.rank_units`), and trajectory u
    draws dropout stream u whatever the sharding.
      * `trajectory_offset`: global index of the first initial condition the windows hold (the reference shards whole
        initial conditions over ranks, `data_loading/inference.py:110-113`);
      * `unit_range=(start, count)`: run only global trajectories start ... start + count - 1 (`ensemble.shard`) - members of
        ONE initial condition split over ranks, or any ragged share of ICs x members.  Writer / aggregator then receive
        flat `(count, time, H, W)` predictions with `start_sample=start`, targets for the initial conditions touched;
      * `relay=ensemble.relay_plan(n_trajectories, world, n_windows, rank)` (+ `relay_comm`, default
        `ensemble.RelayComm()` over the initialised process group): the job's trajectories do not divide by the ranks (25
        members over 8 GPUs).  This rank runs the plan's resident block like a `unit_range` share for the whole job; each
        remainder trajectory is RELAYED: it visits the ranks in turn, every host advancing it through its slice of the windows
        as a batch of one (own `batch_offset`, the window's own dropout call numbers: the samples are those of the unsharded
        job) and handing the stitcher's carried state to the next host with one send / recv.  A host turns to a relay window
        once it has loaded that window itself and the state is there (`ensemble.RelayRunner`: lockstep with catch-up), so the
        loader is consumed in order and no rank waits before the end of the job.  Relay rows reach writer / aggregator as
        flat one-row batches (`start_sample` = the trajectory's global index), window by window, from whichever rank hosts
        them; the windows must hold the relay trajectories' initial conditions as well.
    Without `unit_range` / `relay` every member of every initial condition present runs and predictions are presented as the
    reference stacks them: `(members, n_sample, time, H, W)`.

    `prefetch`: windows pulled ahead of the compute on a background thread (0 = pull each window on the calling thread, after
    the previous one has been handed to the writer, as the reference's loop does without DataLoader workers).  THREADING
    CONTRACT of `prefetch > 0` (the default): the loader is iterated from ONE non-main daemon thread -- it must not depend on
    thread-local or main-thread-only state (torch's default generator, netCDF4 / HDF5 handles opened on the main thread,
    the current CUDA device); pass `prefetch=0` for such a loader.  Window i is handed to the writer / aggregator while
    window i + 1 computes, and only once its status word has come back clean: a flagged window raises before it is written.
    `max_batch`: at most this many trajectories per device batch; a larger share runs as consecutive chunks of the same
    window (each with its own `batch_offset`, so every trajectory still draws the stream of its global index) whose outputs
    are concatenated -- bounds the network workspace (about 0.9 GB per trajectory at 180 x 360, E = 256)."""
    writer = writer if writer is not None else NullDataWriter()
    aggregator = aggregator if aggregator is not None else NullAggregator()
    members = int(n_ensemble_members)
    ens = members > 1
    dev = torch.device(eval_device) if eval_device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("sdy_amd.run_inference runs on the GPU only (no CPU fallback)")
    if max_batch is not None and max_batch < 1:
        raise ValueError(f"max_batch must be >= 1, got {max_batch}")
    if relay is not None:
        if unit_range is not None:
            raise ValueError("relay= carries the rank's resident block: do not pass unit_range as well")
        if not hasattr(stepper.module, "set_dropout_calls"):
            raise ValueError("relay= needs a module whose dropout call counters can be set (set_dropout_calls)")
        unit_range = (relay.start, relay.count)
    if host_outputs:
        writer = _DeferredHostWriter(writer, dev)
    stitcher = WindowStitcher(n_forward_steps, writer, is_ensemble=ens)
    loader = data.loader if hasattr(data, "loader") else data
    timers: Dict[str, float] = defaultdict(float)
    t_begin = now = time.time()
    module = stepper.module
    prefetcher = _WindowPrefetcher(loader, dev, prefetch) if prefetch > 0 else None
    windows = prefetcher if prefetcher is not None else _sync_windows(loader, dev)
    device_spans = []          # (start event, end event) around each batch's device work
    pending: List[tuple] = []  # record_batch calls waiting for THEIR loss terms (flushed one batch later: no drain)
    unit_steps = 0             # trajectories x forecast steps advanced by this process

    def flush(p):
        # A window reaches the writer and the aggregator only after its loss terms and the device's sticky status word have
        # arrived: a window a kernel flagged (fp16 range overflow, non-finite statistics) raises HERE, before either sees it.
        out, i_time_agg, weights, write = p
        loss = float(out.metrics["loss"])      # waits for the window's read-back; raises SdyError on a flagged window
        write()
        kw = {"sample_weights": weights} if (weights is not None and getattr(aggregator, "accepts_sample_weights", False)) \
            else {}
        aggregator.record_batch(loss=loss, target_data=out.target_data, gen_data=out.gen_data,
                                target_data_norm=out.target_data_norm, gen_data_norm=out.gen_data_norm,
                                i_time_start=i_time_agg, **kw)

    def hand_over(entry):
        # batch k - 1: its loss arrived long ago; the device is busy with batch k meanwhile
        pending.append(entry)
        while len(pending) > (1 if prefetch > 0 else 0):
            flush(pending.pop(0))

    def advance(window, win, i, rows_range, stitch, calls0):
        """One device batch of window i: the global trajectories `rows_range` (None: every member of every initial condition
        present) through the stepper, stitched by `stitch`; `calls0`: the dropout call counters window i starts from."""
        nonlocal unit_steps
        i_time = i * forward_steps_in_memory
        cur = torch.cuda.current_stream(dev)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(cur)
        n_sample = next(iter(win.values())).shape[0]
        start, n_rows, ic_list, _, rect = plan_rows(n_sample, members, trajectory_offset, rows_range)
        ic_rows = torch.tensor(ic_list, dtype=torch.long, device=dev)
        # IC-major batch: row r is global trajectory start + r = (IC ic_rows[r], member (start + r) % members)
        batch = {k: v.index_select(0, ic_rows) for k, v in win.items()}
        # the stitcher carries targets for the initial conditions this process touches (all of them unless ragged)
        stitch.apply_initial_condition(batch, ic_rows if rect else ic_rows - ic_list[0])
        step = n_rows if max_batch is None else min(n_rows, int(max_batch))
        parts = []
        for r0 in range(0, n_rows, step):
            r1 = min(n_rows, r0 + step)
            if hasattr(module, "set_batch_offset"):
                module.set_batch_offset(start + r0)
            if calls0 is not None:      # every chunk of the window replays the same call numbers of the dropout streams
                module.set_dropout_calls(calls0)
            chunk = batch if (r0 == 0 and r1 == n_rows) else {k: v[r0:r1] for k, v in batch.items()}
            parts.append((r1 - r0, stepper.run_on_batch(chunk, None, n_forward_steps=forward_steps_in_memory,
                                                        defer_metrics=True)))
        if len(parts) == 1:
            stepped = parts[0][1]
        else:
            cat = lambda name: {k: torch.cat([getattr(s, name)[k] for _, s in parts], dim=0)  # noqa: E731
                                for k in getattr(parts[0][1], name)}
            stepped = SteppedData(metrics=_ChunkedMetrics([(r, s.metrics) for r, s in parts]), gen_data=cat("gen_data"),
                                  target_data=batch, gen_data_norm=cat("gen_data_norm"),
                                  target_data_norm=cat("target_data_norm"))
        del parts
        last_state = {k: v[:, -1] for k, v in stepped.gen_data.items()}
        weights = None
        if rect:       # present like the reference: members on a leading axis (a strided view, no copy)
            ics = slice(None)
            unfold = (lambda d: {k: v.view(n_sample, members, *v.shape[1:]).transpose(0, 1) for k, v in d.items()}) \
                if ens else (lambda d: d)
            first = slice(0, n_rows, members)
        else:          # a ragged share: flat rows, targets of the initial conditions touched
            ics = slice(ic_list[0], ic_list[-1] + 1)
            unfold = lambda d: d  # noqa: E731
            touched = list(range(ic_list[0], ic_list[-1] + 1))
            first = torch.tensor([ic_list.index(c) for c in touched], device=dev)
            # share of each touched initial condition's members that runs in THIS batch: what its targets weigh in a mean
            # over batches and ranks (an initial condition cut by a shard boundary is touched more than once)
            weights = [ic_list.count(c) / members for c in touched]
        flat = not rect
        win_t = {k: v[ics] for k, v in win.items()}
        target_data = derive(win_t) if derive is not None else win_t
        gen_data, gen_norm = unfold(stepped.gen_data), unfold(stepped.gen_data_norm)
        if derive is not None:
            gen_data = derive(gen_data)
        tgt_norm = {k: v[first] for k, v in stepped.target_data_norm.items()}
        out = SteppedData(metrics=stepped.metrics, gen_data=gen_data, target_data=target_data, gen_data_norm=gen_norm,
                          target_data_norm=tgt_norm)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(cur)
        device_spans.append((ev0, ev1))
        unit_steps += n_rows * forward_steps_in_memory
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
        # the carry-over for window i + 1 is taken now; the hand-off to the writer waits for the window's status word
        write = stitch.append(out.target_data, out.gen_data, times, last_state=last_state,
                              start_sample=start if flat else 0, defer_write=True)
        return out, i_time_agg, weights, write

    # ---- relayed remainder trajectories (ensemble.RelayRunner drives relay_step at window boundaries)
    runner = None
    retained: Dict[int, tuple] = {}        # window index -> (window, device tensors, last time step of the window before)
    calls_base = module.dropout_calls() if hasattr(module, "dropout_calls") else None
    calls_per_window = None

    def calls_at(w):
        return tuple(b + w * d for b, d in zip(calls_base, calls_per_window)) if w > 0 else tuple(calls_base)

    if relay is not None and relay.tasks:
        from . import ensemble

        comm = relay_comm if relay_comm is not None else ensemble.RelayComm(device=dev)
        relay_stitch: Dict[int, WindowStitcher] = {}
        gen_names: List[str] = list(stepper.out_names)

        def pack_state(st):        # the stitcher's carried generated state of a one-row batch: (variables, H, W)
            return torch.stack([st._carry_gen[k][0] for k in gen_names], dim=0)

        def state_like(task):
            any_win = next(iter(retained.values()))[1]
            h, wd = next(iter(any_win.values())).shape[-2:]
            return torch.empty(len(gen_names), h, wd, dtype=torch.float32, device=dev)

        def relay_step(task, w, state):
            window, win, prev_last = retained[w]
            st = relay_stitch.get(task.unit)
            if st is None:          # the trajectory arrives here: a stitcher that stands where window w begins
                st = relay_stitch[task.unit] = WindowStitcher(n_forward_steps, writer, is_ensemble=ens)
                if w > 0:
                    ic = task.unit // members - trajectory_offset
                    st.i_time = w * forward_steps_in_memory + 1
                    st._carry_gen = {k: state[j:j + 1] for j, k in enumerate(gen_names)}
                    st._carry_target = {k: v[ic:ic + 1] for k, v in prev_last.items()}
            # (the window's own call numbers; the resident batch sets its own again at its next window)
            hand_over(advance(window, win, w, (task.unit, 1), st, calls_at(w)))
            if w + 1 >= task.w_end:          # the slice ends: what travels is the stitcher's carried generated state
                relay_stitch.pop(task.unit)
                return pack_state(st) if w + 1 < n_windows else None
            return state

        runner = ensemble.RelayRunner(relay, comm, relay_step, lambda task: None, state_like)
    n_windows = n_forward_steps // forward_steps_in_memory
    last_step = None           # last time step of the previous window's data (a relay trajectory's carried targets)

    try:
        for i, (window, win) in enumerate(windows):
            timers["data_loading"] += time.time() - now
            now = time.time()
            calls0 = None
            if calls_base is not None:
                # (relay work in between moves the module's counters: with a relay every window starts from its own call
                #  numbers, base + i x the calls of one window)
                calls0 = calls_at(i) if (runner is not None and (calls_per_window is not None or i == 0)) \
                    else module.dropout_calls()
            if unit_range is None or unit_range[1] > 0:
                entry = advance(window, win, i, unit_range, stitcher, calls0)
            else:
                entry = None
            if calls_base is not None and calls_per_window is None and entry is not None:
                calls_per_window = tuple(b - a for a, b in zip(calls0, module.dropout_calls()))
            if prefetch <= 0:      # the reference's timer semantics: the window is complete when the clock is read
                torch.cuda.current_stream(dev).synchronize()
            timers["run_on_batch_host"] += time.time() - now
            now = time.time()
            if entry is not None:
                hand_over(entry)
                del entry
            if runner is not None:
                # a window some hosted slice still has to go through stays (its device tensors; a slice whose state has not
                # arrived yet catches up later), with the last time step of the window before it
                if i in runner.pending_windows():
                    retained[i] = (window, win, last_step)
                last_step = {k: v[:, -1].clone() for k, v in win.items()}
                t_relay = time.time()
                runner.after_window(i)
                timers["relay_host"] += time.time() - t_relay      # (host time of the hosted relay windows; the device time
                keep = runner.pending_windows()                     #  of their batches is part of run_on_batch)
                for w in [w for w in retained if w not in keep]:
                    del retained[w]
                now += time.time() - t_relay
            timers["writer_and_aggregator"] += time.time() - now
            now = time.time()
        if runner is not None:
            t_relay = time.time()
            runner.drain(n_windows - 1)
            timers["relay_host"] += time.time() - t_relay
            now += time.time() - t_relay
            timers["relay_recv_wait"] = getattr(runner.comm, "recv_wait_s", 0.0)
        while pending:
            flush(pending.pop(0))
    finally:
        if prefetcher is not None:
            prefetcher.close()
    if hasattr(writer, "flush"):
        writer.flush()
    torch.cuda.current_stream(dev).synchronize()
    timers["writer_and_aggregator"] += time.time() - now
    timers["run_on_batch"] = sum(a.elapsed_time(b) for a, b in device_spans) * 1e-3     # device time of the batches
    wall = time.time() - t_begin
    timers["wall"] = wall
    timers["trajectory_steps"] = float(unit_steps)         # trajectories x forecast steps this process advanced
    if wall > 0:
        # the reference logs n_forward_steps x n_ICs over the whole duration (inference.py:294-298); here: x trajectories
        timers["forecast_steps_per_second"] = unit_steps / wall
    if timers["run_on_batch"] > 0:
        timers["forecast_steps_per_second_run_on_batch"] = unit_steps / timers["run_on_batch"]
    for name, duration in timers.items():
        print(f"{name}: {duration:.2f}" + ("" if ("per_second" in name or name == "trajectory_steps") else "s"))
    return dict(timers)
