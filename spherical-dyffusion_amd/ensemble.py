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


def relay_due(task: RelayTask, resident: int) -> int:
    """Resident window index before which a rank turns to a relay slice.  The slice's state arrives when its previous hosts
    have advanced the trajectory through w_begin windows as a batch of ONE, which costs at most ~1.5 x a single trajectory's
    share of a resident window; turning to it any earlier would only wait for the hand-over."""
    return int(task.w_begin * min(1.0, 1.5 / max(resident, 1)))


def run_relay(plan: RelayPlan, n_windows: int, resident_step: Callable[[int], None],
              relay_step: Callable[[RelayTask, int, object], object], initial_state: Callable[[int], object],
              recv: Callable[[RelayTask], object], send: Callable[[RelayTask, object], None]) -> dict:
    """Drives one rank through its plan: `resident_step(w)` advances the resident batch through window w;
    `relay_step(task, w, state) -> state` advances a relay trajectory through window w; `recv(task)` / `send(task, state)`
    move a relay trajectory's state between ranks (blocking receive, non-blocking send); `initial_state(unit)` starts one.
    Returns {unit: final state} for the relay trajectories that END on this rank."""
    finals = {}
    pending = list(plan.tasks)

    def run(task: RelayTask):
        state = initial_state(task.unit) if task.src is None else recv(task)
        for w in range(task.w_begin, task.w_end):
            state = relay_step(task, w, state)
        if task.dst is None:
            finals[task.unit] = state
        else:
            send(task, state)

    for w in range(n_windows):
        while pending and (plan.count == 0 or relay_due(pending[0], plan.count) <= w):
            run(pending.pop(0))
        if plan.count > 0:
            resident_step(w)
    while pending:
        run(pending.pop(0))
    return finals
