#!/usr/bin/env python3
"""BASELINE.json configs[3] / [4] through the whole product path: `run_inference` (window driver, loader prefetch) ->
`MultiStepStepper` (normalise, pack, prescriber-free AR loop, denormalise, LpLoss terms) -> `get_preds_at_t_for_batch` ->
DYffusion sampler -> SFNO, with the on-device `TimeMeanAggregator`, for (initial conditions) x (members) trajectories on
synthetic standardised data.  Rank 0 prints one JSON line: member-forecast-steps/s of the driver BY WALL TIME (the
reference's "Total steps per second", `src/ace_inference/inference/inference.py:294-298`, x trajectories) with the device
time of the windows beside it, finiteness and the aggregator's channel-mean RMSE.

    python tools/c4_rollout.py --steps 102 --members 25                    # C4: 25 members x ~100 steps = 17 windows of 6
    python tools/c4_rollout.py --steps 600 --members 25                    # a C5-style long sample, extrapolated
    python tools/c4_rollout.py --steps 102 --members 25 --gpus 8           # C4 as specified: 3 members per GPU + 1 relayed
    python tools/c4_rollout.py --steps 600 --members 25 --ics 4 --gpus 8   # C5's job (100 trajectories: 12 per GPU + 4 relayed)

`--gpus N` works by itself: the parent starts one child per GPU before anything touches HIP (never re-executes a process
that has).  A job whose trajectories divide by the ranks runs `run_inference(unit_range=ensemble.shard(...))` on every rank;
one that does not (25 over 8) runs `run_inference(relay=ensemble.relay_plan(...))`: equal resident blocks and the remainder
trajectories relayed between the ranks in time slices (`--no-relay`: the static 4,3,3,... split, whose largest share sets the
pace).  No collective on the data path; the `TimeMeanAggregator(dist=TorchDistributed())` combines the ranks' maps once, at
log time (RCCL).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(device, n_out=63, n_forc=2, layers=8, embed=256, nlat=180, nlon=360, horizon=6):
    """The shipped layout: one input-only channel (HGTsfc) carried in front of the state (hack_for_imprecise_interpolation)."""
    from sdy_amd import synthetic

    exp, _, _ = synthetic.build_sampler(device, state_chans=n_out, forcing_chans=n_forc, nlat=nlat, nlon=nlon, embed=embed,
                                        layers=layers, horizon=horizon, carried_input_only_channel=True)
    stepper, names, out_names = synthetic.build_stepper(exp, n_out, n_forc, carried_input_only_channel=True)
    return exp, stepper, names, out_names


def windows(names, n_windows, window, nlat, nlon, seed=1234, n_ics=1):
    from sdy_amd import synthetic

    return synthetic.windows(names, n_windows, window, nlat, nlon, n_ics=n_ics, seed=seed)


def run(device, steps, members, window=6, layers=8, embed=256, nlat=180, nlon=360, aggregate=True, warmup=True, n_ics=1,
        rank=0, world=1, prefetch=2, max_batch=None, dist=None, relay=True):
    import torch

    import sdy_amd
    from sdy_amd import ensemble

    assert steps % window == 0, "--steps must be a multiple of the window (6)"
    exp, stepper, names, out_names = build(device, layers=layers, embed=embed, nlat=nlat, nlon=nlon)
    start, cnt, ic_lo, n_ic = ensemble.shard(n_ics, members, rank, world)
    kw = dict(n_ensemble_members=members, eval_device=device, prefetch=prefetch, max_batch=max_batch)
    n_traj = n_ics * members
    plan = ensemble.relay_plan(n_traj, world, steps // window, rank) if (relay and world > 1) else None
    if plan is not None and not any(ensemble.relay_plan(n_traj, world, steps // window, r).tasks for r in range(world)):
        plan = None                                      # the job divides by the ranks (or has fewer trajectories): static split
    if plan is not None:
        # resident block + relayed remainder: the windows hold every initial condition (a relay trajectory's may be any)
        start, cnt, ic_lo, n_ic = plan.start, plan.count, 0, n_ics
        kw.update(relay=plan, trajectory_offset=0)
    elif world > 1:
        kw.update(unit_range=(start, cnt), trajectory_offset=ic_lo)
    agg = None
    if aggregate:
        w = sdy_amd.metrics.spherical_area_weights(torch.linspace(-89.5, 89.5, nlat), nlon)
        agg = sdy_amd.metrics.TimeMeanAggregator(w, is_ensemble=members > 1, dist=dist)
    finite = {"ok": True}

    class Writer:     # stands for the reference's data writer: only checks what it is handed
        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            v = prediction[out_names[0]]
            finite["ok"] = finite["ok"] and bool(torch.isfinite(v).all())
            finite["shape"] = tuple(v.shape)

    def loader(n_windows, seed):     # a rank's loader delivers only the initial conditions its share touches
        for wdw in windows(names, n_windows, window, nlat, nlon, seed=seed, n_ics=n_ics):
            if world > 1 and plan is None:
                wdw.data = {k: v[ic_lo:ic_lo + n_ic] for k, v in wdw.data.items()}
            yield wdw

    assert cnt > 0, "every rank has a share (main() rejects world > ics x members)"
    if warmup:   # one untimed window: native objects, weight upload and the workspace are created on first use (~8 s)
        wkw = {k: v for k, v in kw.items() if k != "relay"}
        if plan is not None:
            wkw.update(unit_range=(start, cnt))
        sdy_amd.run_inference(None, stepper, loader(1, 7), window, window, **wkw)
        if plan is not None and plan.tasks:      # (a batch of one as well: its workspace, and the ring's connections)
            sdy_amd.run_inference(None, stepper, loader(1, 7), window, window, **dict(wkw, unit_range=(plan.tasks[0].unit, 1)))
    comm = None
    if plan is not None:
        comm = ensemble.RelayComm(device=device)
        comm.warm_up()
        kw.update(relay_comm=comm)
    exp.set_dropout_calls((0, 0))        # every rank numbers the job's dropout calls from the same origin
    if dist is not None and world > 1:
        torch.cuda.synchronize(device)
        import torch.distributed as td
        td.barrier()
    t0 = time.perf_counter()
    timers = sdy_amd.run_inference(agg, stepper, loader(steps // window, 1234), steps, window, writer=Writer(), **kw)
    wall = time.perf_counter() - t0
    res = {"steps": steps, "members": members, "ics": n_ics, "rows": cnt, "windows": steps // window, "wall_s": round(wall, 2),
           "relayed_windows": sum(t.w_end - t.w_begin for t in plan.tasks) if plan is not None else 0,
           "relay_recv_wait_s": round(timers.get("relay_recv_wait", 0.0), 3),
           "trajectory_steps": timers["trajectory_steps"],
           "run_on_batch_s": round(timers["run_on_batch"], 2), "data_loading_wait_s": round(timers["data_loading"], 2),
           "member_forecast_steps_per_s": round(timers["forecast_steps_per_second"], 2),
           "member_forecast_steps_per_s_device_time": round(timers["forecast_steps_per_second_run_on_batch"], 2),
           "wall_over_run_on_batch": round(wall / max(timers["run_on_batch"], 1e-9), 3),
           "finite": finite["ok"], "prediction_shape": finite.get("shape")}
    if agg is not None:
        res["time_mean_rmse_channel_mean"] = round(agg.get_logs("")["rmse/channel_mean"], 5)
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(n):
    """One child per GPU, started before this process imports torch or touches HIP (same rule as bench.py)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, pending = 0, list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is None:
                continue
            pending.remove(p)
            if r != 0:
                rc = rc or r
                for q in pending:       # a rank failed: the others would wait in a barrier forever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=102)
    ap.add_argument("--members", type=int, default=25)
    ap.add_argument("--ics", type=int, default=1, help="initial conditions (C5: 4)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--prefetch", type=int, default=2, help="windows pulled ahead of the compute (0: the synchronous loop)")
    ap.add_argument("--max-batch", type=int, default=None, help="at most this many trajectories per device batch")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--embed", type=int, default=256)
    ap.add_argument("--grid", type=int, nargs=2, default=(180, 360))
    ap.add_argument("--no-relay", action="store_true",
                    help="uneven jobs as a static split (25 over 8 = 4,3,3,...) instead of equal blocks + relayed remainder")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: all ranks on GPU 0 with the gloo backend (exercises the N>1 path on a 1-GPU box)")
    a = ap.parse_args()
    if a.gpus > a.ics * a.members:
        # a rank with an empty share would skip the barrier and the aggregator's reduce_sum collectives the others enter
        raise SystemExit(f"--gpus {a.gpus} exceeds the job's {a.ics * a.members} trajectories (ics x members): every rank needs a share")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_children(a.gpus))

    import torch
    import torch.distributed as td

    import sdy_amd

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.share_gpu:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm
        dist = sdy_amd.metrics.TorchDistributed() if not a.share_gpu else _HostDist(td)
    r = run(dev, a.steps, a.members, layers=a.layers, embed=a.embed, nlat=a.grid[0], nlon=a.grid[1], n_ics=a.ics, rank=rank,
            world=world, prefetch=a.prefetch, max_batch=a.max_batch, dist=dist, relay=not a.no_relay)
    if world > 1:     # whole-job rate: every rank's trajectories over the slowest rank's wall time
        t = torch.tensor([r.get("wall_s", 0.0), float(r.get("trajectory_steps", 0.0))], dtype=torch.float64,
                         device="cpu" if a.share_gpu else dev)
        tmax, tsum = t.clone(), t.clone()
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        td.all_reduce(tsum, op=td.ReduceOp.SUM)
        r["n_gpus"] = world
        r["job_wall_s"] = round(float(tmax[0]), 2)
        r["job_member_forecast_steps_per_s"] = round(float(tsum[1]) / max(float(tmax[0]), 1e-9), 2)
    if rank == 0:
        rate = r.get("job_member_forecast_steps_per_s", r.get("member_forecast_steps_per_s", 0.0))
        if rate:
            r["c5_hours"] = round(100 * 14600 / rate / 3600.0, 2)    # 4 ICs x 25 members x 14600 steps at this job's rate
        print(json.dumps(r), flush=True)
    if world > 1:
        td.barrier()
        td.destroy_process_group()


class _HostDist:
    """--share-gpu only: gloo cannot reduce device tensors, so the maps take a round trip through the host."""

    def __init__(self, td):
        self.td = td

    @property
    def world_size(self):
        return self.td.get_world_size()

    def reduce_sum(self, t):
        h = t.detach().cpu().clone()
        self.td.all_reduce(h)
        return h.to(t.device)

    def reduce_mean(self, t):
        return self.reduce_sum(t) / self.world_size


if __name__ == "__main__":
    main()
