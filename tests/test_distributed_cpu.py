"""world_size-2 gloo test of the N>1 path: units are sharded with no data-path collective, the dropout stream of a
trajectory depends only on its global index, and the bench's max-over-ranks timing reduction works."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_units, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sdy_amd
    from oracle.philox import drop_path_keep, element_keep_mask

    start, cnt = sdy_amd.ensemble.partition(n_units, world)[rank]
    # this rank's share of the dropout stream, addressed by GLOBAL trajectory index (batch_offset = start)
    m = element_keep_mask(seed=99, call=3, layer=1, kind=0, p=0.1, B=cnt, C=8, H=4, W=8, batch_offset=start)
    dpk = drop_path_keep(seed=99, call=3, layer=2, p=0.3, B=cnt, batch_offset=start)
    t = torch.tensor([0.5 + rank], dtype=torch.float64)        # per-rank wall time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py: max over ranks
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([cnt]))
    ret[rank] = (start, cnt, m, dpk, float(t), [int(c) for c in counts])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_is_invariant():
    world, n_units = 2, 5
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_units, ret), nprocs=world, join=True)
    from oracle.philox import drop_path_keep, element_keep_mask

    full = element_keep_mask(seed=99, call=3, layer=1, kind=0, p=0.1, B=n_units, C=8, H=4, W=8)
    full_dp = drop_path_keep(seed=99, call=3, layer=2, p=0.3, B=n_units)
    got = np.concatenate([ret[r][2] for r in range(world)], axis=0)
    got_dp = np.concatenate([ret[r][3] for r in range(world)], axis=0)
    assert np.array_equal(got, full) and np.array_equal(got_dp, full_dp)
    assert [ret[r][1] for r in range(world)] == [3, 2] and ret[0][5] == [3, 2]
    assert ret[0][4] == ret[1][4] == 1.5
