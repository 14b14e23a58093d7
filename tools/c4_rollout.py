#!/usr/bin/env python3
"""BASELINE.json configs[3] / [4] through the whole product path: `run_inference` (window driver) -> `MultiStepStepper`
(normalise, pack, prescriber-free AR loop, denormalise, LpLoss terms) -> `get_preds_at_t_for_batch` -> DYffusion sampler ->
SFNO, with the on-device `TimeMeanAggregator`, for an M-member ensemble of one initial condition on synthetic standardised
data.  Prints one JSON line: member-forecast-steps/s of the driver (its own `forecast_steps_per_second` timer, i.e. the
reference's "Total steps per second" log line x trajectories), finiteness and the aggregator's channel-mean RMSE.

    python tools/c4_rollout.py --steps 102 --members 25          # C4: 25 members x ~100 steps = 17 windows of 6
    python tools/c4_rollout.py --steps 600 --members 25          # a C5-style long sample (600 steps), extrapolated
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(device, n_out=63, n_forc=2, layers=8, embed=256, nlat=180, nlon=360, horizon=6):
    """The shipped layout: one input-only channel (HGTsfc) carried in front of the state (hack_for_imprecise_interpolation)."""
    import torch

    import sdy_amd
    from helpers import make_pair
    from oracle.sfno import SFNOConfig

    cs = n_out + 1
    fcfg = SFNOConfig(in_chans=cs + n_forc, out_chans=n_out, nlat=nlat, nlon=nlon, embed_dim=embed, num_layers=layers,
                      with_time_emb=True, min_time=0.0, max_time=horizon - 1.0)
    icfg = SFNOConfig(in_chans=2 * cs + n_forc, out_chans=n_out, nlat=nlat, nlon=nlon, embed_dim=embed, num_layers=layers,
                      with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=horizon - 1.0)
    with torch.cuda.device(device):
        fnet, _, _ = make_pair(fcfg, cs, n_forc, seed=4321)
        inet, _, _ = make_pair(icfg, 2 * cs, n_forc, seed=4322, net_seed=1000)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=horizon), horizon=horizon,
        diffusion_config=dict(hack_for_imprecise_interpolation=True))
    out_names = [f"v{i}" for i in range(n_out)]
    in_names = ["HGTsfc"] + out_names
    forcing = [f"f{i}" for i in range(n_forc)]
    names = in_names + forcing
    stepper = sdy_amd.MultiStepStepper(exp, names, out_names, forcing, {n: 0.0 for n in names}, {n: 1.0 for n in names}, None)
    return exp, stepper, names, out_names


def windows(names, n_windows, window, nlat, nlon, seed=1234):
    """Synthetic standardised series, generated window by window on the host (targets are only used for the loss terms)."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed)
    last = {n: torch.randn(1, 1, nlat, nlon, generator=g) for n in names}
    for _ in range(n_windows):
        data = {}
        for n in names:
            nxt = torch.randn(1, window, nlat, nlon, generator=g)
            data[n] = torch.cat([last[n], nxt], dim=1)
            last[n] = data[n][:, -1:]
        yield types.SimpleNamespace(data=data, times=None)


def run(device, steps, members, window=6, layers=8, embed=256, nlat=180, nlon=360, aggregate=True, warmup=True):
    import torch

    import sdy_amd

    assert steps % window == 0, "--steps must be a multiple of the window (6)"
    exp, stepper, names, out_names = build(device, layers=layers, embed=embed, nlat=nlat, nlon=nlon)
    agg = None
    if aggregate:
        w = sdy_amd.metrics.spherical_area_weights(torch.linspace(-89.5, 89.5, nlat), nlon)
        agg = sdy_amd.metrics.TimeMeanAggregator(w, is_ensemble=members > 1)
    finite = {"ok": True}

    class Writer:     # stands for the reference's data writer: only checks what it is handed
        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            v = prediction[out_names[0]]
            finite["ok"] = finite["ok"] and bool(torch.isfinite(v).all())
            finite["shape"] = tuple(v.shape)

    if warmup:   # one untimed window: native objects, weight upload and the workspace are created on first use (~8 s)
        sdy_amd.run_inference(None, stepper, windows(names, 1, window, nlat, nlon, seed=7), window, window,
                              n_ensemble_members=members, eval_device=device)
    t0 = time.perf_counter()
    timers = sdy_amd.run_inference(agg, stepper, windows(names, steps // window, window, nlat, nlon), steps, window,
                                   n_ensemble_members=members, eval_device=device, writer=Writer())
    wall = time.perf_counter() - t0
    res = {"steps": steps, "members": members, "windows": steps // window, "wall_s": round(wall, 2),
           "run_on_batch_s": round(timers["run_on_batch"], 2),
           "member_forecast_steps_per_s": round(timers["forecast_steps_per_second"], 2),
           "finite": finite["ok"], "prediction_shape": finite.get("shape")}
    if agg is not None:
        res["time_mean_rmse_channel_mean"] = round(agg.get_logs("")["rmse/channel_mean"], 5)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=102)
    ap.add_argument("--members", type=int, default=25)
    a = ap.parse_args()
    import torch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    r = run(dev, a.steps, a.members)
    r["c5_hours_1gpu"] = round(100 * 14600 / r["member_forecast_steps_per_s"] / 3600.0, 2)   # 4 ICs x 25 members x 14600 steps
    print(json.dumps(r), flush=True)
