"""Static partition of independent trajectories (initial condition x ensemble member) over ranks.

The reference shards initial conditions over ranks (`src/ace_inference/core/data_loading/inference.py:110-113`,
`i_sample % world_size != rank -> skip`) and loops members serially (`src/ace_inference/inference/loop.py:199-208`).
Trajectories are independent, so here every (IC, member) pair is a unit, units are split in contiguous blocks over
ranks (batched on the GPU), and there is NO collective on the data path.  A unit's dropout stream depends only on its
GLOBAL index (`batch_offset` in the C ABI), so results do not depend on how units are sharded.
"""
from __future__ import annotations

from typing import List, Tuple


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


def batches(units: List[Tuple[int, int]], max_batch: int) -> List[List[Tuple[int, int]]]:
    """Split a rank's units into device batches of at most `max_batch` trajectories."""
    assert max_batch >= 1
    return [units[i:i + max_batch] for i in range(0, len(units), max_batch)]
