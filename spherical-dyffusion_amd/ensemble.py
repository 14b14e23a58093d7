"""Static partition of independent trajectories (initial condition x ensemble member) over ranks.

The reference shards initial conditions over ranks (`src/ace_inference/core/data_loading/inference.py:110-113`,
`i_sample % world_size != rank -> skip`) and loops members serially (`src/ace_inference/inference/loop.py:199-208`).
Trajectories are independent, so here every (IC, member) pair is a unit, units are split in contiguous blocks over
ranks (batched on the GPU), and there is NO collective on the data path.  A unit's dropout stream depends only on its
GLOBAL index (`batch_offset` in the C ABI), so results do not depend on how units are sharded.
"""
from __future__ import annotations

from typing import List, Optional, Tuple


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
