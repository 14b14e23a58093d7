"""world_size-2 gloo tests of the N>1 path (CPU): the product's own sharding plumbing - `ensemble.shard` ->
`ensemble.plan_rows` -> `MultiHorizonForecastingDYffusion.set_batch_offset` - gives every (initial condition, member)
trajectory the same global index on every world size, ranks tile the job exactly with no data-path collective, and the
bench's max-over-ranks timing reduction works.  (The arithmetic itself needs a GPU: `-m gpu`,
tests/test_gpu_dyffusion.py::test_sharding_invariance_of_trajectories.)"""
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


def _tiny_module():
    import sdy_amd

    kw = dict(spatial_shape_in=(16, 32), embed_dim=8, num_layers=1, with_time_emb=True)
    fnet = sdy_amd.SphericalFourierNeuralOperatorNet(4, 4, num_conditional_channels=1, **kw)
    inet = sdy_amd.SphericalFourierNeuralOperatorNet(8, 4, num_conditional_channels=1, dropout_mlp=0.1, drop_path_rate=0.1,
                                                     **kw)
    return sdy_amd.MultiHorizonForecastingDYffusion(fnet, sdy_amd.InterpolationExperiment(inet, horizon=6), horizon=6)


def _worker(rank, world, port, n_ics, members, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.philox import drop_path_keep, element_keep_mask
    from sdy_amd import ensemble

    # the product's plumbing, exactly as loop.run_inference / bench.py drive it
    start, cnt, ic_lo, n_ic = ensemble.shard(n_ics, members, rank, world)
    s2, c2, ic_rows, mem, rect = ensemble.plan_rows(n_ic, members, first_ic=ic_lo, unit_range=(start, cnt))
    module = _tiny_module()
    module.set_batch_offset(s2)
    offs = (module.model.model.batch_offset, module.model.interpolator.model.batch_offset)
    units = [(ic_lo + r, m) for r, m in zip(ic_rows, mem)]                    # (global IC, member) of every device row
    glob = [offs[1] + r for r in range(c2)]                                   # global index the C ABI gives row r
    # this rank's share of the dropout stream, addressed the way the device addresses it (batch_offset + row)
    m = element_keep_mask(seed=99, call=3, layer=1, kind=0, p=0.1, B=cnt, C=8, H=4, W=8, batch_offset=offs[1])
    dpk = drop_path_keep(seed=99, call=3, layer=2, p=0.3, B=cnt, batch_offset=offs[1])
    t = torch.tensor([0.5 + rank], dtype=torch.float64)        # per-rank wall time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py: max over ranks
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([cnt]))
    ret[rank] = dict(start=start, cnt=cnt, offs=offs, units=units, glob=glob, mask=m, dp=dpk, t=float(t),
                     counts=[int(c) for c in counts], rect=rect, ics=(ic_lo, n_ic))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_ics, members):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_ics, members, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def test_two_rank_member_split_is_invariant():
    """BASELINE C4 shape in miniature: ONE initial condition, 5 members over 2 ranks -> 3 + 2."""
    from oracle.philox import drop_path_keep, element_keep_mask

    n_ics, members = 1, 5
    r = _run(2, n_ics, members)
    assert [x["cnt"] for x in r] == [3, 2] and r[0]["counts"] == [3, 2]
    assert [x["offs"] for x in r] == [(0, 0), (3, 3)]                          # both networks keyed by the first global unit
    assert r[0]["units"] + r[1]["units"] == [(0, m) for m in range(members)]
    assert r[0]["glob"] + r[1]["glob"] == list(range(members))                 # global index = ic * members + member
    full = element_keep_mask(seed=99, call=3, layer=1, kind=0, p=0.1, B=members, C=8, H=4, W=8)
    full_dp = drop_path_keep(seed=99, call=3, layer=2, p=0.3, B=members)
    assert np.array_equal(np.concatenate([x["mask"] for x in r], 0), full)
    assert np.array_equal(np.concatenate([x["dp"] for x in r], 0), full_dp)
    assert r[0]["t"] == r[1]["t"] == 1.5


def test_two_rank_ragged_ic_member_split():
    """BASELINE C5 shape in miniature: 3 initial conditions x 3 members over 2 ranks -> 5 + 4, the cut falls inside IC 1."""
    n_ics, members = 3, 3
    r = _run(2, n_ics, members)
    assert [x["cnt"] for x in r] == [5, 4]
    assert r[0]["ics"] == (0, 2) and r[1]["ics"] == (1, 2)                      # IC 1 is needed by both ranks
    assert not r[0]["rect"] and not r[1]["rect"]
    units = r[0]["units"] + r[1]["units"]
    assert units == [(ic, m) for ic in range(n_ics) for m in range(members)]   # exact tiling, IC-major
    glob = r[0]["glob"] + r[1]["glob"]
    assert glob == [ic * members + m for ic, m in units]                       # what ensemble.rank_units promises
    # the same job on ONE rank numbers every trajectory identically
    from sdy_amd import ensemble
    s, c, rows, mem, rect = ensemble.plan_rows(n_ics, members)
    assert rect and [s + i for i in range(c)] == glob and list(zip(rows, mem)) == units


def _reduce_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sdy_amd.metrics import TorchDistributed

    d = TorchDistributed()
    m = d.reduce_mean(torch.full((4, 8), float(rank + 1)))          # a rank's time-mean map
    ret[rank] = (d.world_size, m)
    dist.barrier()
    dist.destroy_process_group()


def test_time_mean_maps_reduce_over_ranks():
    """The aggregator's only collective (off the sampling path): `reduce_mean` of the (H, W) maps over ranks, as the
    reference's Distributed.reduce_mean (time_mean.py:147-148)."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reduce_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        assert ret[r][0] == 2 and torch.equal(ret[r][1], torch.full((4, 8), 1.5))
    from sdy_amd.metrics import TorchDistributed
    assert TorchDistributed().reduce_mean(torch.ones(2)).tolist() == [1.0, 1.0]     # no process group: identity



def _weighted_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sdy_amd import ensemble
    from sdy_amd.metrics import TimeMeanAggregator, TorchDistributed

    # 5 members of one initial condition over 2 ranks -> 3 + 2 rows; two windows.  What record_batch leaves behind on a rank
    # (the sums over ITS rows of the rows' time means, and its row count) is written directly: the accumulation kernel needs a
    # GPU (tests/test_gpu_golden.py), the combination over ranks is what this test is about.
    g = torch.Generator().manual_seed(5)
    tm = torch.randn(2, 5, 4, 8, generator=g)                 # per-window, per-trajectory time means
    tgt = torch.randn(2, 4, 8, generator=g)                   # per-window target time mean (one IC)
    start, cnt, _, _ = ensemble.shard(1, 5, rank, world)
    agg = TimeMeanAggregator(torch.ones(4, 8), dist=TorchDistributed(), is_ensemble=True)
    agg._gen_data = {"a": tm[:, start:start + cnt].sum(dim=(0, 1))}
    agg._gen_rows = 2.0 * cnt
    w = cnt / 5.0                                             # run_inference's sample_weights for a cut initial condition
    agg._target_data = {"a": (tgt * w).sum(dim=0)}
    agg._target_rows = 2.0 * w
    agg._n_batches = 2
    maps = agg.time_mean_maps()
    ret[rank] = (cnt, maps["gen"]["a"], maps["target"]["a"], tm.mean(dim=(0, 1)), tgt.mean(dim=0))
    dist.barrier()
    dist.destroy_process_group()


def test_time_mean_maps_weigh_ranks_by_their_rows():
    """Uneven shards (25 members over 8 GPUs are 4, 3, 3, ...; here 5 over 2 = 3 + 2): the maps over ranks must equal the
    single-process maps, i.e. every trajectory weighs the same -- sums and row counts are all-reduced, not rank means."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_weighted_worker, args=(2, port, ret), nprocs=2, join=True)
    assert [ret[r][0] for r in range(2)] == [3, 2]
    for r in range(2):
        _, gen, tgt, gen_want, tgt_want = ret[r]
        assert torch.allclose(gen, gen_want, rtol=1e-6, atol=1e-6)
        assert torch.allclose(tgt, tgt_want, rtol=1e-6, atol=1e-6)
    # an unweighted mean of the two ranks' means would NOT be the single-process map
    g = torch.Generator().manual_seed(5)
    tm = torch.randn(2, 5, 4, 8, generator=g)
    naive = 0.5 * (tm[:, :3].mean(dim=(0, 1)) + tm[:, 3:].mean(dim=(0, 1)))
    assert not torch.allclose(naive, tm.mean(dim=(0, 1)), rtol=1e-3, atol=1e-3)


def _advance(x, unit, w):
    """Stand-in for one window of one trajectory (the network needs a GPU): a deterministic, order-sensitive update keyed
    by (global trajectory, window) -- any mix-up of states, owners or windows changes the result."""
    return x * 1.0001 + torch.sin(torch.arange(4, dtype=torch.float64) + 7.0 * unit + 0.37 * w) * (1.0 + x.abs().sum())


def _relay_worker(rank, world, port, n_units, n_windows, delays, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sdy_amd import ensemble

    plan = ensemble.relay_plan(n_units, world, n_windows, rank)
    res = {u: torch.full((4,), float(u), dtype=torch.float64) for u in range(plan.start, plan.start + plan.count)}
    log = []

    def resident_step(w):
        for u in res:
            res[u] = _advance(res[u], u, w)
        log.append(("res", w))
        if delays:                      # skewed ranks: some run ahead of their neighbours, some behind
            time.sleep(delays[rank])

    def relay_step(task, w, x):
        assert ("res", w) in log or plan.count == 0, "a relay window is advanced only after the rank has seen that window"
        log.append(("relay", task.unit, w))
        return _advance(x, task.unit, w)

    comm = ensemble.RelayComm()         # the product's transport: isend / recv announced through the group's store (gloo here)
    comm.warm_up()
    finals = ensemble.run_relay(plan, n_windows, resident_step, relay_step,
                                lambda u: torch.full((4,), float(u), dtype=torch.float64), comm,
                                like=lambda task: torch.empty(4, dtype=torch.float64))
    out = dict(res)
    out.update(finals)
    ret[rank] = (plan, out, log, comm.recv_wait_s)
    dist.barrier()
    dist.destroy_process_group()


def _serial(n_units, n_windows):
    out = {}
    for u in range(n_units):
        x = torch.full((4,), float(u), dtype=torch.float64)
        for w in range(n_windows):
            x = _advance(x, u, w)
        out[u] = x
    return out


def test_relayed_remainder_trajectories_equal_the_unsharded_run():
    """ensemble.relay_plan / RelayRunner / RelayComm: 5 trajectories over 2 ranks = 2 resident each + ONE relayed in two time
    slices (the 25-over-8 case in miniature: 3 resident each + one relayed through 8 slices); 7 over 3 = 2 each + one relayed;
    8 over 3 = 2 each + TWO relayed (their chains start on different ranks and wrap around the ring); with the product's
    transport between the ranks (gloo here, RCCL on GPUs) and with ranks that run at different speeds, so that states arrive
    early on some hosts and late on others.  Every trajectory's final state equals the serial run bit for bit, every
    (trajectory, window) is advanced exactly once -- a relay window never before its host has seen that window -- and the
    work is even."""
    for world, n_units, n_windows, delays in ((2, 5, 6, None), (3, 7, 7, None), (3, 8, 5, None), (3, 7, 6, (0.0, 0.03, 0.0)),
                                              (3, 8, 6, (0.03, 0.0, 0.01))):
        port = _free_port()
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_relay_worker, args=(world, port, n_units, n_windows, delays, ret), nprocs=world, join=True)
        want = _serial(n_units, n_windows)
        got, seen, load = {}, [], []
        for r in range(world):
            plan, out, log, _ = ret[r]
            assert plan.count == n_units // world
            got.update(out)
            seen += [(u, w) for kind, *rest in log if kind == "res" for w in rest for u in range(plan.start, plan.start + plan.count)]
            seen += [(rest[0], rest[1]) for kind, *rest in log if kind == "relay"]
            load.append(plan.count * n_windows + sum(t.w_end - t.w_begin for t in plan.tasks))
        assert sorted(got) == list(range(n_units))
        for u in range(n_units):
            assert torch.equal(got[u], want[u]), (world, n_units, u)
        assert sorted(seen) == [(u, w) for u in range(n_units) for w in range(n_windows)]
        assert max(load) - min(load) <= (n_units % world) * (-(-n_windows // world)), load


def test_relay_schedule_reaches_the_balanced_makespan():
    """The lockstep-with-catch-up policy of RelayRunner, simulated with the measured one-GPU pass times (tools/relay_projection.py):
    no rank idles before the end of the job, so the job takes what the most loaded rank's own work takes -- q resident windows
    each plus its slices of the relay trajectory -- against the 4-member rank's 20 windows of a static split."""
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "relay_projection", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "relay_projection.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    t = {1: 51.0, 3: 126.6, 4: 164.1, 6: 239.0, 7: 277.3, 12: 466.9, 13: 502.9}
    for world in (2, 4, 8):
        makespan, finish = rp.simulate(25, world, 20, t)
        q = 25 // world
        own = [20 * t[q] + sum(tk.w_end - tk.w_begin for tk in rp.ensemble.relay_plan(25, world, 20, r).tasks) * t[1]
               for r in range(world)]
        assert abs(makespan - max(own)) < 1e-6 * makespan, (world, makespan, max(own))
        assert makespan < 20 * t[q + 1]                      # the static split's pace


def test_relay_plan_of_the_headline_job():
    """25 members over 8 GPUs, 20 windows: 3 resident members per rank, member 24 relayed through all eight ranks in slices of
    2-3 windows; every rank carries 62-63 member-windows instead of 80 on the rank a static 4,3,3,... split overloads."""
    from sdy_amd import ensemble

    plans = [ensemble.relay_plan(25, 8, 20, r) for r in range(8)]
    assert [(p.start, p.count) for p in plans] == [(3 * r, 3) for r in range(8)]
    tasks = [t for p in plans for t in p.tasks]
    assert all(t.unit == 24 for t in tasks) and len(tasks) == 8
    chain = sorted(tasks, key=lambda t: t.w_begin)
    assert chain[0].src is None and chain[-1].dst is None and chain[0].w_begin == 0 and chain[-1].w_end == 20
    for a, b in zip(chain, chain[1:]):
        assert a.w_end == b.w_begin
    hosts = [r for t in chain for r in range(8) if t in plans[r].tasks]
    assert hosts == list(range(8))
    for a, b, h in zip(chain, chain[1:], hosts):
        assert a.dst == h + 1 and b.src == h
    load = [p.count * 20 + sum(t.w_end - t.w_begin for t in p.tasks) for p in plans]
    assert sum(load) == 25 * 20 and max(load) <= 63
    # divisible jobs and jobs with fewer trajectories than ranks fall back to the static split
    assert ensemble.relay_plan(24, 8, 20, 3) == ensemble.RelayPlan(9, 3, ())
    assert [ensemble.relay_plan(5, 8, 20, r).count for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
