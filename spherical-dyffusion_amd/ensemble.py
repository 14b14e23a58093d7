"""Static partition of independent trajectories (initial condition x ensemble member) over ranks.

The reference shards initial conditions over ranks (`src/ace_inference/core/data_loading/inference.py:110-113`,
`i_sample % world_size != rank -> skip`) and loops members serially (`src/ace_inference/inference/loop.py:199-208`).
Trajectories are independent, so here every (IC, member) pair is a unit, units are split in contiguous blocks over
ranks (batched on the GPU), and there is NO collective on the data path.  A unit's dropout stream depends only on its
GLOBAL index (`batch_offset` in the C ABI), so results do not depend on how units are sharded.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple


def partition(n_units: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous block split: [(start, count)] per rank; sizes differ by at most one (e.g. 25 over 8 -> 4,3,3,3,3,3,3,3)."""
    assert n_units >= 0 and world_size >= 1
    q, r = divmod(n_units, world_size)
    out, start = [], 0
    for rank in range(world_size):
        cnt = q + (1 if rank < r else 0)
        out.append((start, cnt))
        start += cnt
    return out


def rank_units(n_ics: int, n_members: int, rank: int, world_size: int) -> List[Tuple[int, int]]:
    """(ic, member) pairs owned by `rank`; global unit index = ic * n_members + member."""
    start, cnt = partition(n_ics * n_members, world_size)[rank]
    return [divmod(u, n_members) for u in range(start, start + cnt)]


def shard(n_ics: int, n_members: int, rank: int, world_size: int) -> Tuple[int, int, int, int]:
    """This rank's share of an (n_ics x n_members) job: (first global unit, unit count, first IC, IC count).  The units
    are contiguous in the global index `ic * n_members + member`, so a rank's device batch is simply its unit range in
    that order and `batch_offset = first global unit` keys every row's dropout stream; the ICs it needs from the data
    loader are `first IC ... first IC + IC count - 1` (a shard may start or end in the middle of an IC's members)."""
    start, cnt = partition(n_ics * n_members, world_size)[rank]
    if cnt == 0:
        return start, 0, 0, 0
    ic_lo, ic_hi = start // n_members, (start + cnt - 1) // n_members
    return start, cnt, ic_lo, ic_hi - ic_lo + 1


def plan_rows(n_sample: int, n_members: int, first_ic: int = 0, unit_range: Optional[Tuple[int, int]] = None):
    """Row layout of one device batch.

    `n_sample` ICs are present (global indices first_ic ... first_ic + n_sample - 1), each with `n_members` members.
    Rows are the global units `start ... start + count - 1` (default: every member of every IC present), in the order of
    the global unit index u = ic * n_members + member, i.e. IC-major: all members of an IC are adjacent.
    Returns (start, count, ic_rows, members, rectangular): `ic_rows[r]` = row of the window data unit r starts from,
    `members[r]` = its member number, `rectangular` = the rows are exactly (all ICs present) x (all members)."""
    assert n_sample >= 1 and n_members >= 1 and first_ic >= 0
    if unit_range is None:
        start, count = first_ic * n_members, n_sample * n_members
    else:
        start, count = int(unit_range[0]), int(unit_range[1])
    units = range(start, start + count)
    ic_rows = [u // n_members - first_ic for u in units]
    if count and not (0 <= ic_rows[0] and ic_rows[-1] < n_sample):
        raise ValueError(f"units {start}..{start + count - 1} need initial conditions {start // n_members}.."
                         f"{(start + count - 1) // n_members}, the window holds {first_ic}..{first_ic + n_sample - 1}")
    rect = start == first_ic * n_members and count == n_sample * n_members
    return start, count, ic_rows, [u % n_members for u in units], rect


def batches(units: List[Tuple[int, int]], max_batch: int) -> List[List[Tuple[int, int]]]:
    """Split a rank's units into device batches of at most `max_batch` trajectories."""
    assert max_batch >= 1
    return [units[i:i + max_batch] for i in range(0, len(units), max_batch)]


# ---- relay schedule: the remainder trajectories of an uneven split ----------------------------------------------------
# 25 members over 8 GPUs are 4, 3, 3, ... as a static split: the 4-member rank sets the pace and the job runs at 25 / 32 = 78 %
# of the machine.  The reference never meets this (it shards whole initial conditions, data_loading/inference.py:110-113);
# the metric here does.  Trajectories are independent and a trajectory is a chain of windows, so the remainder can be cut
# along TIME instead: every rank keeps q = n // N resident trajectories for the whole job, and each of the r = n % N relay
# trajectories visits the ranks in turn -- rank (s + j stride) % N advances relay trajectory j through the s-th slice of the
# job's windows as a batch of its own, then hands its state (one tensor, 16 MB at 63 x 180 x 360, plus the stream position)
# to the next host.  A relay trajectory alone advances faster than any resident batch, so it is never what a rank waits
# for; every rank ends up with q + r / N trajectories' worth of work.  Every row is keyed by (global trajectory index,
# window), so no result depends on the schedule (tests/test_distributed_cpu.py, tests/test_gpu_fullsize.py).
@dataclass(frozen=True)
class RelayTask:
    unit: int                 # global index of the relay trajectory
    w_begin: int              # windows [w_begin, w_end) of the job are advanced on this rank
    w_end: int
    src: Optional[int]        # rank the state comes from (None: the trajectory starts here, from its initial condition)
    dst: Optional[int]        # rank the state goes to afterwards (None: the trajectory ends here)


@dataclass(frozen=True)
class RelayPlan:
    start: int                # resident trajectories of this rank: global units start .. start + count - 1
    count: int
    tasks: Tuple[RelayTask, ...]   # relay slices hosted by this rank, by rising w_begin


def relay_plan(n_units: int, world_size: int, n_windows: int, rank: int) -> RelayPlan:
    """This rank's share of `n_units` independent trajectories over `n_windows` windows: resident block + relay slices.
    With n_units % world_size == 0 (or fewer units than ranks) it is `partition`'s block and no relay."""
    assert n_units >= 0 and world_size >= 1 and n_windows >= 0 and 0 <= rank < world_size
    q, r = divmod(n_units, world_size)
    if q == 0 or r == 0 or n_windows == 0:
        start, cnt = partition(n_units, world_size)[rank]
        return RelayPlan(start, cnt, ())
    stride = world_size // r                                  # relay trajectories start on ranks 0, stride, 2 stride, ...
    cuts = [s * n_windows // world_size for s in range(world_size + 1)]
    slices = [s for s in range(world_size) if cuts[s + 1] > cuts[s]]          # (fewer windows than ranks: some slices are empty)
    tasks = []
    for j in range(r):
        hosts = [(s + j * stride) % world_size for s in slices]
        for i, s in enumerate(slices):
            if hosts[i] == rank:
                tasks.append(RelayTask(unit=world_size * q + j, w_begin=cuts[s], w_end=cuts[s + 1],
                                       src=hosts[i - 1] if i > 0 else None, dst=hosts[i + 1] if i + 1 < len(slices) else None))
    tasks.sort(key=lambda t: (t.w_begin, t.unit))
    return RelayPlan(rank * q, q, tuple(tasks))


class RelayComm:
    """Hand-over of a relay trajectory's state between ranks over `torch.distributed` point-to-point (RCCL over xGMI with the
    "nccl" backend and device tensors; gloo with host tensors in the CPU tests and in `--share-gpu` test runs).

    A receive is never posted before its send exists: the sender announces every message in the process group's store right
    after `isend`, `ready(task, like)` is a non-blocking `store.check` of the next announcement (and takes the message when
    it is there), and a blocking `recv(task, like)` only happens at the end of the job, when there is nothing left to overlap.  An early-posted RCCL receive would sit on the GPU
    spinning for the whole time its sender needs to get there; this way the only wait is the sender's `isend`, which the
    receiver picks up at its next window boundary.  `job` keys one run (ranks must agree on it: a counter per process).
    """
    _jobs = 0

    def __init__(self, device=None, job=None):
        import torch
        import torch.distributed as dist

        assert dist.is_initialized(), "RelayComm needs an initialised torch.distributed process group"
        self._torch, self._dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.on_device = dist.get_backend() == "nccl"
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if self.on_device else torch.device("cpu"))
        self.store = dist.distributed_c10d._get_default_store()
        if job is None:
            job = RelayComm._jobs
            RelayComm._jobs += 1
        self.job = job
        self._sends = []
        self._seq_out, self._seq_in, self._inbox = {}, {}, {}
        self.recv_wait_s = 0.0        # time spent blocked in recv (the schedule's idle time on this rank)

    def _key(self, src: int, dst: int, seq: int) -> str:
        return f"sdy_relay/{self.job}/{src}>{dst}/{seq}"

    def warm_up(self) -> None:
        """Connects every (rank, rank + 1) pair of the ring with the SAME unbatched send / recv the hand-overs use (RCCL
        creates a two-rank communicator per pair on first use: that belongs in front of a timed region).  Even ranks send
        first, odd ranks receive first, so no two ranks wait for each other."""
        if self.world < 2:
            return
        t, dist = self._torch, self._dist
        dev = self.device if self.on_device else t.device("cpu")
        ping, pong = t.zeros(8, device=dev), t.empty(8, device=dev)
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        if self.rank % 2 == 0:
            dist.send(ping, nxt)
            dist.recv(pong, prv)
        else:
            dist.recv(pong, prv)
            dist.send(ping, nxt)

    def send(self, task: "RelayTask", state) -> None:
        """Non-blocking: the tensor is handed to the backend (a private copy) and announced in the store.  Messages of one
        (source, destination) pair are numbered, and the announcement names the trajectory: the receiver takes them in the
        order they were sent (RCCL matches point-to-point messages of a pair in order; it has no tags)."""
        buf = state.detach().to(self.device if self.on_device else "cpu").contiguous().clone()
        req = self._dist.isend(buf, dst=task.dst)
        self._sends.append((req, buf))
        seq = self._seq_out.get(task.dst, 0)
        self._seq_out[task.dst] = seq + 1
        self.store.set(self._key(self.rank, task.dst, seq), f"{task.unit},{task.w_end}")

    def _take(self, src: int, like, block: bool) -> bool:
        """Receives the next announced message of `src` into the inbox; False when none is announced (and not `block`)."""
        import time

        seq = self._seq_in.get(src, 0)
        key = self._key(src, self.rank, seq)
        if not block and not self.store.check([key]):
            return False
        t0 = time.perf_counter()
        unit, w = (int(v) for v in self.store.get(key).decode().split(","))      # (store.get waits for the key)
        buf = self._torch.empty(like.shape, dtype=like.dtype, device=self.device if self.on_device else "cpu")
        self._dist.recv(buf, src=src)
        self.recv_wait_s += time.perf_counter() - t0
        self._seq_in[src] = seq + 1
        self._inbox[(unit, w)] = buf
        return True

    def ready(self, task: "RelayTask", like) -> bool:
        while (task.unit, task.w_begin) not in self._inbox and self._take(task.src, like, block=False):
            pass
        return (task.unit, task.w_begin) in self._inbox

    def recv(self, task: "RelayTask", like):
        """The state `task.src` sent for this slice (blocking until it has been announced and received); `like` gives shape /
        dtype / device of the result."""
        while (task.unit, task.w_begin) not in self._inbox:
            self._take(task.src, like, block=True)
        return self._inbox.pop((task.unit, task.w_begin)).to(like.device)

    def finish(self) -> None:
        for req, _ in self._sends:
            req.wait()
        self._sends.clear()


class RelayRunner:
    """One rank's relay work, driven from its window loop.  Policy: LOCKSTEP WITH CATCH-UP -- after the rank has seen window
    w (its data has been loaded and its resident batch advanced), every hosted relay trajectory whose state is here is
    advanced through all of its windows <= w; one whose state has not arrived yet is simply skipped (the resident batch goes
    on) and catches up in one burst when it does.  No rank ever idles before the end of the job, nothing needs window data
    the loader has not delivered yet, and a rank's total is q windows of the resident batch plus its slices of the relay
    trajectories: the makespan of the ideal balanced schedule (tools/relay_projection.py simulates it).

    `step(task, w, state) -> state` advances the trajectory through window w; `initial_state(task)` starts one;
    `comm` is a `RelayComm` (or anything with send(task, state) / ready(task, like) / recv(task, like) / finish());
    `like(task)` describes the state tensor a receive produces.  `pending_windows()` tells the caller which windows' data it still has to keep."""

    def __init__(self, plan: RelayPlan, comm, step: Callable, initial_state: Callable, like: Callable):
        self.plan, self.comm, self.step, self.initial_state, self.like = plan, comm, step, initial_state, like
        self._todo = [{"task": t, "next": t.w_begin, "state": None, "have": False} for t in plan.tasks]
        self.finals = {}
        self.log: List[Tuple[int, int]] = []          # (unit, window) in the order they were advanced

    def pending_windows(self) -> set:
        return {w for e in self._todo for w in range(e["next"], e["task"].w_end)}

    def _acquire(self, e, block: bool) -> bool:
        if e["have"]:
            return True
        t = e["task"]
        if t.src is None:
            e["state"] = self.initial_state(t)
        elif block or self.comm.ready(t, self.like(t)):
            e["state"] = self.comm.recv(t, self.like(t))
        else:
            return False
        e["have"] = True
        return True

    def _advance(self, e, upto: int) -> None:
        t = e["task"]
        while e["next"] <= min(upto, t.w_end - 1):
            e["state"] = self.step(t, e["next"], e["state"])
            self.log.append((t.unit, e["next"]))
            e["next"] += 1
        if e["next"] >= t.w_end:
            if t.dst is None:
                self.finals[t.unit] = e["state"]
            else:
                self.comm.send(t, e["state"])
            e["state"] = None

    def after_window(self, w: int) -> None:
        """The caller has loaded window w (and advanced its resident batch through it)."""
        for e in list(self._todo):
            if e["task"].w_begin <= w and self._acquire(e, block=False):
                self._advance(e, w)
                if e["next"] >= e["task"].w_end:
                    self._todo.remove(e)

    def drain(self, last_window: int) -> dict:
        """End of the job: whatever is still outstanding, in the order of the slices, with blocking receives."""
        for e in sorted(self._todo, key=lambda e: (e["task"].w_begin, e["task"].unit)):
            self._acquire(e, block=True)
            self._advance(e, last_window)
        self._todo.clear()
        self.comm.finish()
        return self.finals


def run_relay(plan: RelayPlan, n_windows: int, resident_step: Callable[[int], None],
              relay_step: Callable[[RelayTask, int, object], object], initial_state: Callable[[int], object],
              comm, like: Optional[Callable] = None) -> dict:
    """Drives one rank through its plan when windows need no loader (bench.py's sampler-only job, tests): `resident_step(w)`
    advances the resident batch through window w, the `RelayRunner` policy does the rest.  Returns {unit: final state} for
    the relay trajectories that END on this rank."""
    runner = RelayRunner(plan, comm, relay_step, lambda t: initial_state(t.unit), like or (lambda t: None))
    for w in range(n_windows):
        if plan.count > 0:
            resident_step(w)
        runner.after_window(w)
    return runner.drain(n_windows - 1)
