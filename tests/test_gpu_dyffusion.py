"""Sampler-level parity: DYffusion.sample / get_preds_at_t_for_batch vs the oracle restatement of sample_loop."""
import pytest
import torch

from conftest import rel_l2
from helpers import PhiloxMasks, make_pair
from oracle.dyffusion import OracleDYffusion
from oracle.sfno import SFNOConfig

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _build(hack=False, dropout=True, nlat=32, nlon=64, E=16, L=2, C=6, n_forc=2, horizon=6, seed_f=11, seed_i=22):
    import sdy_amd

    cs = C + (1 if hack else 0)   # state channels carried by the sampler
    fcfg = SFNOConfig(in_chans=cs + n_forc, out_chans=C, nlat=nlat, nlon=nlon, embed_dim=E, num_layers=L,
                      with_time_emb=True, min_time=0.0, max_time=float(horizon - 1))
    icfg = SFNOConfig(in_chans=2 * cs + n_forc, out_chans=C, nlat=nlat, nlon=nlon, embed_dim=E, num_layers=L,
                      with_time_emb=True, dropout_mlp=0.1 if dropout else 0.0, drop_path_rate=0.1 if dropout else 0.0,
                      min_time=1.0, max_time=float(horizon - 1))
    fnet, fora, _ = make_pair(fcfg, cs, n_forc, seed=seed_f)
    inet, iora, _ = make_pair(icfg, 2 * cs, n_forc, seed=seed_i, net_seed=4242)
    ipol = sdy_amd.InterpolationExperiment(inet, horizon=horizon)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, ipol, horizon=horizon,
        diffusion_config=dict(hack_for_imprecise_interpolation=hack, enable_interpolator_dropout=dropout))
    masks = PhiloxMasks(icfg, seed=4242)
    calls = {"n": 0}

    def ora_f(x, time, condition=None, static_condition=None):
        return fora(x, time=time, condition=condition, static_condition=static_condition)

    def ora_i(x, time, condition=None, static_condition=None):
        masks.call = calls["n"]
        calls["n"] += 1
        return iora(x, time=time, condition=condition, static_condition=static_condition,
                    mask_fn=masks if dropout else None)

    oracle = OracleDYffusion(ora_f, ora_i, timesteps=horizon, hack_for_imprecise_interpolation=hack)
    oracle._calls, oracle._masks = calls, masks      # the window-driver test replays per-member dropout streams
    return exp, oracle, cs, n_forc


@pytest.mark.parametrize("hack", [False, True])
def test_sample_matches_oracle(hack):
    exp, oracle, cs, n_forc = _build(hack=hack)
    g = torch.Generator(device="cpu").manual_seed(1234)
    B = 2
    x0 = torch.randn(B, cs, 32, 64, generator=g)
    if hack:   # run_inference path: forcings arrive as static_condition (stepper_multistep.py:383-384)
        kw = {"static_condition": torch.randn(B, n_forc, 32, 64, generator=g)}
    else:      # Lightning path: dynamical_condition (B, T+1, Cc, H, W)
        kw = {"dynamical_condition": torch.randn(B, 7, n_forc, 32, 64, generator=g)}
    ref = oracle.sample(x0, **kw)
    got = exp.model.sample(x0.cuda(), **{k: v.cuda() for k, v in kw.items()})
    assert sorted(got.keys()) == sorted(ref.keys()) == [f"t{i}_preds" for i in range(1, 7)]
    for k in ref:
        err = rel_l2(got[k], ref[k])
        assert got[k].shape == ref[k].shape
        assert err < TOL, f"{k} (hack={hack}): rel L2 {err:.3e}"


@pytest.mark.parametrize("schedule", [None, "every2nd"])
def test_sample_with_artificial_steps_and_per_call_dropout_matches_oracle(schedule):
    """additional_interpolation_steps = 2 with `enable_interpolator_dropout="except_dynamical_steps"` on the PRODUCT's own dropout
    stream (Philox, replayed by the oracle): the steps that land on an artificial time run their interpolations with dropout on,
    the others with it off -- in the product as stacked pairs with shared inputs (B = 2 <= 8), in the oracle call by call.  The
    call counter advances with every interpolator forward whether or not it draws, so the oracle numbers its calls likewise.
    (The reference's recorded masks drive the same configuration in tests/test_gpu_golden.py.)"""
    import sdy_amd

    horizon, hack = 6, True
    cs, n_forc, C = 7, 2, 6
    fcfg = SFNOConfig(in_chans=cs + n_forc, out_chans=C, nlat=32, nlon=64, embed_dim=16, num_layers=2, with_time_emb=True,
                      min_time=0.0, max_time=5.0)
    icfg = SFNOConfig(in_chans=2 * cs + n_forc, out_chans=C, nlat=32, nlon=64, embed_dim=16, num_layers=2, with_time_emb=True,
                      dropout_mlp=0.1, drop_path_rate=0.1, min_time=0.0, max_time=5.0)
    fnet, fora, _ = make_pair(fcfg, cs, n_forc, seed=11)
    inet, iora, _ = make_pair(icfg, 2 * cs, n_forc, seed=22, net_seed=4242)
    ipol = sdy_amd.InterpolationExperiment(inet, horizon=horizon)
    inet.set_min_max_time(0.0, 5.0)          # (the experiment pins [1, 5]; artificial steps need the range opened)
    extra = dict(additional_interpolation_steps=2, enable_interpolator_dropout="except_dynamical_steps")
    if schedule:
        extra["sampling_schedule"] = schedule
    exp = sdy_amd.MultiHorizonForecastingDYffusion(fnet, ipol, horizon=horizon,
                                                   diffusion_config=dict(hack_for_imprecise_interpolation=hack, **extra))
    masks = PhiloxMasks(icfg, seed=4242)
    calls = {"n": 0, "on": 0}
    oracle = None

    def ora_i(x, time, condition=None, static_condition=None):
        masks.call = calls["n"]
        calls["n"] += 1
        calls["on"] += int(oracle.dropout_on)
        return iora(x, time=time, condition=condition, static_condition=static_condition,
                    mask_fn=masks if oracle.dropout_on else None)

    oracle = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(
        x, time=time, condition=condition, static_condition=static_condition), ora_i, timesteps=horizon,
        hack_for_imprecise_interpolation=hack, **extra)
    g = torch.Generator(device="cpu").manual_seed(4)
    x0 = torch.randn(2, cs, 32, 64, generator=g)
    sc = torch.randn(2, n_forc, 32, 64, generator=g)
    ref = oracle.sample(x0, static_condition=sc)
    got = exp.model.sample(x0.cuda(), static_condition=sc.cuda())
    assert inet._call == calls["n"] and 0 < calls["on"] < calls["n"]
    assert sorted(got) == sorted(ref) == [f"t{i}_preds" for i in range(1, 7)]
    for k in ref:
        err = rel_l2(got[k], ref[k])
        assert err < 2e-5, f"{k}: rel L2 {err:.3e}"


def test_get_preds_at_t_for_batch_surface():
    """The stepper's call pattern (stepper_multistep.py:365-427): horizon 1 computes, 2..6 pop the cache."""
    exp, oracle, cs, n_forc = _build(hack=True)
    g = torch.Generator(device="cpu").manual_seed(7)
    x0 = torch.randn(1, cs, 32, 64, generator=g)
    sc = torch.randn(1, n_forc, 32, 64, generator=g)
    ref = oracle.sample(x0, static_condition=sc)
    for h in range(1, 7):
        batch = {"dynamics": x0.cuda(), "static_condition": sc.cuda()}
        with exp.ema_scope(), exp.inference_dropout_scope():
            out = exp.get_preds_at_t_for_batch(batch, horizon=h, split="predict", is_autoregressive=False,
                                               prepare_inputs=False, ensemble=False, num_predictions=1)
        assert list(out.keys()) == [f"t{h}_preds_normed"]
        err = rel_l2(out[f"t{h}_preds_normed"], ref[f"t{h}_preds"])
        assert err < TOL, f"h={h}: rel L2 {err:.3e}"
    assert exp._current_preds is None
    with pytest.raises(AssertionError):
        exp.get_preds_at_t_for_batch({"dynamics": x0.cuda()}, horizon=7, split="predict", prepare_inputs=False)


def test_call_trace_is_16_forwards():
    """6 forecaster + 10 interpolator calls per horizon-6 pass, in the reference's order."""
    exp, _, cs, n_forc = _build(hack=False, dropout=False)
    trace = []
    f_net, i_net = exp.model.model, exp.model.interpolator.model
    f_orig, i_orig = f_net.forward, i_net.forward
    f_net.forward = lambda *a, **k: (trace.append(("F", float(k["time"][0]))), f_orig(*a, **k))[1]
    i_net.forward = lambda *a, **k: (trace.append(("I", float(k["time"][0]))), i_orig(*a, **k))[1]
    x0 = torch.randn(1, cs, 32, 64).cuda()
    dyn = torch.randn(1, 7, n_forc, 32, 64).cuda()
    exp.model.fuse_interpolator_pair_max_batch = 0       # the reference's own sequence: one forward per call
    exp.model.sample(x0, dynamical_condition=dyn)
    expect = [("F", 0.0), ("I", 1.0), ("F", 1.0), ("I", 2.0), ("I", 1.0), ("F", 2.0), ("I", 3.0), ("I", 2.0),
              ("F", 3.0), ("I", 4.0), ("I", 3.0), ("F", 4.0), ("I", 5.0), ("I", 4.0), ("F", 5.0), ("I", 5.0)]
    assert trace == expect
    # small batches stack the two interpolations of a step (times s + 1 and s) into one forward of 2B rows: 12 launches of
    # the network for the same 16 calls, in the same call order (row blocks: [s + 1 | s])
    del trace[:]
    f_net.forward = lambda *a, **k: (trace.append(("F", [float(k["time"][0])])), f_orig(*a, **k))[1]
    i_net.forward = lambda *a, **k: (trace.append(("I", [float(v) for v in k["time"]])), i_orig(*a, **k))[1]
    exp.model.fuse_interpolator_pair_max_batch = 8
    f_net._call = i_net._call = 0
    exp.model.sample(x0, dynamical_condition=dyn)
    assert trace == [("F", [0.0]), ("I", [1.0]), ("F", [1.0]), ("I", [2.0, 1.0]), ("F", [2.0]), ("I", [3.0, 2.0]), ("F", [3.0]),
                     ("I", [4.0, 3.0]), ("F", [4.0]), ("I", [5.0, 4.0]), ("F", [5.0]), ("I", [5.0])]
    assert (f_net._call, i_net._call) == (6, 10)


def test_stepper_interpolating_prescriber_matches_oracle():
    """MultiStepStepper with an interpolating prescriber (prescriber.py:78-81) and dropout on, vs the oracle stepper."""
    import sdy_amd
    from oracle.stepper import run_on_batch

    exp, oracle, cs, n_forc = _build(hack=True, dropout=True)
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]
    out_names, forcing_names = in_names[1:], ["f0", "f1"]
    g = torch.Generator(device="cpu").manual_seed(99)
    B, T1 = 2, 8
    means = {n: float(torch.randn((), generator=g)) for n in in_names + forcing_names}
    stds = {n: float(torch.rand((), generator=g) + 0.5) for n in in_names + forcing_names}
    data = {n: torch.randn(B, T1, 32, 64, generator=g) * stds[n] + means[n] for n in in_names + forcing_names}
    data["frac"] = torch.rand(B, T1, 32, 64, generator=g)
    pres = dict(prescribed_name="v3", mask_name="frac", mask_value=1, interpolate=True)

    class OMod:   # oracle-side module with the reference's prediction cache
        true_horizon = 6

        def __init__(self):
            self.cache = None
        from contextlib import nullcontext
        ema_scope = inference_dropout_scope = staticmethod(nullcontext)

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            if horizon == 1:
                self.cache = oracle.sample(batch["dynamics"], static_condition=batch["static_condition"])
            return {f"t{horizon}_preds_normed": self.cache[f"t{horizon}_preds"]}

    tm = {k: torch.tensor(v) for k, v in means.items()}
    ts = {k: torch.tensor(v) for k, v in stds.items()}
    metrics, gen, gen_norm = run_on_batch(data, OMod(), in_names, out_names, forcing_names, tm, ts, T1 - 1, pres, hack=True)
    stepper = sdy_amd.MultiStepStepper(exp, in_names + forcing_names, out_names, forcing_names, means, stds,
                                       sdy_amd.Prescriber(**pres))
    out = stepper.run_on_batch({k: v.cuda() for k, v in data.items()}, None, n_forward_steps=T1 - 1)
    for n in out_names:
        err = rel_l2(out.gen_data[n], gen[n])
        assert err < TOL, f"{n}: {err:.3e}"
    assert abs(float(out.metrics["loss"]) - metrics["loss"]) < 1e-3 * metrics["loss"]


def test_window_driver_batched_members_match_serial_oracle():
    """run_inference with dropout ON: 3 members x 2 samples batched on the device over two windows == the oracle's
    serial member loop, every trajectory (sample s, member m) drawing the Philox stream of its global index
    s * members + m (ensemble.rank_units), whatever the batch it runs in."""
    import types

    import sdy_amd
    from oracle.loop import run_inference as oracle_run
    from oracle.stepper import run_on_batch

    exp, oracle, cs, n_forc = _build(hack=True, dropout=True)
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]
    out_names, forcing_names = in_names[1:], ["f0", "f1"]
    g = torch.Generator(device="cpu").manual_seed(5)
    n_sample, members, n_mem, n_total = 2, 3, 6, 12
    means = {n: float(torch.randn((), generator=g)) for n in in_names + forcing_names}
    stds = {n: float(torch.rand((), generator=g) + 0.5) for n in in_names + forcing_names}
    series = {n: torch.randn(n_sample, n_total + 1, 32, 64, generator=g) * stds[n] + means[n] for n in in_names + forcing_names}
    windows = [{k: v[:, i * n_mem:(i + 1) * n_mem + 1] for k, v in series.items()} for i in range(n_total // n_mem)]
    tm = {k: torch.tensor(v) for k, v in means.items()}
    ts = {k: torch.tensor(v) for k, v in stds.items()}
    state = {"window": -1, "last_member": members}

    class OMod:
        true_horizon = 6
        from contextlib import nullcontext
        ema_scope = inference_dropout_scope = staticmethod(nullcontext)

        def __init__(self):
            self.cache = None

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            if horizon == 1:
                self.cache = oracle.sample(batch["dynamics"], static_condition=batch["static_condition"])
            return {f"t{horizon}_preds_normed": self.cache[f"t{horizon}_preds"]}

    def rob(data, m):
        if m <= state["last_member"] and m == 0:
            state["window"] += 1
        state["last_member"] = m
        oracle._calls["n"] = 10 * state["window"]          # 10 interpolator calls per horizon-6 pass = per window
        oracle._masks.rows = [s_ * members + m for s_ in range(n_sample)]      # global index of (IC s_, member m)
        return run_on_batch(data, OMod(), in_names, out_names, forcing_names, tm, ts, n_mem, None, hack=True)

    wref, aref = oracle_run(windows, rob, n_total, n_mem, members)
    got = []

    class W:
        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            got.append((start_timestep, {k: v.clone() for k, v in prediction.items()}))

    stepper = sdy_amd.MultiStepStepper(exp, in_names + forcing_names, out_names, forcing_names, means, stds, None)
    loader = [types.SimpleNamespace(data=w, times=None) for w in windows]
    sdy_amd.run_inference(None, stepper, loader, n_total, n_mem, members, writer=W())
    assert [c[0] for c in got] == [c[0] for c in wref]
    for (_, pr), (_, pg) in zip(wref, got):
        for n in out_names:
            assert pg[n].shape == pr[n].shape == (members, n_sample, pr[n].shape[2], 32, 64)
            err = rel_l2(pg[n], pr[n])
            assert err < TOL, f"{n}: {err:.3e}"
    # members really differ (dropout streams are per trajectory)
    v = got[0][1][out_names[0]]
    assert float((v[0] - v[1]).abs().max()) > 1e-3


def _run_sharded(exp, stepper, windows, n_total, n_mem, members, **kw):
    """One run_inference from a fresh dropout-call counter; returns {global trajectory: (time, H, W) per variable}."""
    import types

    import sdy_amd

    exp.model.model._call = 0
    exp.model.interpolator.model._call = 0
    got = []

    class W:
        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            got.append((start_sample, {k: v.clone() for k, v in prediction.items()},
                        {k: tuple(v.shape) for k, v in target.items()}))

    sdy_amd.run_inference(None, stepper, [types.SimpleNamespace(data=w, times=None) for w in windows], n_total, n_mem,
                          members, writer=W(), **kw)
    return got


def test_sharding_invariance_of_trajectories():
    """a trajectory's result does not depend on how (initial condition x member) units are sharded.
    (a) one IC x 5 members on one process == members 0-2 on one shard + members 3-4 on another (`unit_range`, the
        BASELINE 25-member split); (b) 2 ICs x 2 members on one process == one IC per process (`trajectory_offset`, the
        reference's IC sharding) == a ragged 3 + 1 split.  Dropout is ON: equality needs every row to draw the stream of
        its GLOBAL index ic * members + member.  Masks are bit-identical; values agree to fp32 rounding (the InstanceNorm
        sums are accumulated in an order that depends on the batch size)."""
    import sdy_amd
    from sdy_amd import ensemble

    exp, _, cs, n_forc = _build(hack=True, dropout=True)
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]
    out_names, forcing_names = in_names[1:], ["f0", "f1"]
    g = torch.Generator(device="cpu").manual_seed(17)
    n_mem, n_total = 6, 12
    means = {n: 0.1 * i for i, n in enumerate(in_names + forcing_names)}
    stds = {n: 1.0 + 0.1 * i for i, n in enumerate(in_names + forcing_names)}
    stepper = sdy_amd.MultiStepStepper(exp, in_names + forcing_names, out_names, forcing_names, means, stds, None)

    def series(n_sample):
        s = {n: torch.randn(n_sample, n_total + 1, 32, 64, generator=g) * stds[n] + means[n]
             for n in in_names + forcing_names}
        return [{k: v[:, i * n_mem:(i + 1) * n_mem + 1] for k, v in s.items()} for i in range(n_total // n_mem)]

    def close(a, b, what):
        e = rel_l2(a, b)
        assert e < 2e-5, f"{what}: {e:.3e}"

    # ---- (a) members of one IC split 3 + 2
    wins = series(1)
    members = 5
    full = _run_sharded(exp, stepper, wins, n_total, n_mem, members)
    assert full[0][1][out_names[0]].shape[:2] == (members, 1)
    for rank in range(2):
        start, cnt, ic_lo, n_ic = ensemble.shard(1, members, rank, 2)
        assert (start, cnt, ic_lo, n_ic) == ((0, 3, 0, 1) if rank == 0 else (3, 2, 0, 1))
        part = _run_sharded(exp, stepper, wins, n_total, n_mem, members, unit_range=(start, cnt))
        for w in range(len(wins)):
            assert part[w][0] == start
            for n in out_names:
                assert part[w][1][n].shape[0] == cnt
                close(part[w][1][n], full[w][1][n][start:start + cnt, 0], f"members split, window {w}, {n}, rank {rank}")
    v = full[0][1][out_names[0]]
    assert float((v[0] - v[1]).abs().max()) > 1e-3          # members really differ

    # ---- (b) 2 ICs x 2 members: whole, one IC per process, ragged 3 + 1
    wins = series(2)
    members = 2
    full = _run_sharded(exp, stepper, wins, n_total, n_mem, members)
    for ic in range(2):
        part = _run_sharded(exp, stepper, [{k: v[ic:ic + 1] for k, v in w.items()} for w in wins], n_total, n_mem,
                            members, trajectory_offset=ic)
        for w in range(len(wins)):
            for n in out_names:
                assert part[w][1][n].shape[:2] == (members, 1)
                close(part[w][1][n][:, 0], full[w][1][n][:, ic], f"IC sharding, window {w}, {n}, ic {ic}")
    for start, cnt in ((0, 3), (3, 1)):
        part = _run_sharded(exp, stepper, wins, n_total, n_mem, members, unit_range=(start, cnt))
        for w in range(len(wins)):
            n_ic_touched = (start + cnt - 1) // members - start // members + 1
            assert all(s[0] == n_ic_touched for s in part[w][2].values())
            for n in out_names:
                for r in range(cnt):
                    ic, m = divmod(start + r, members)
                    close(part[w][1][n][r], full[w][1][n][m, ic], f"ragged, window {w}, {n}, unit {start + r}")
    with pytest.raises(ValueError):      # a shard that needs an IC the window does not hold
        _run_sharded(exp, stepper, [{k: v[:1] for k, v in w.items()} for w in wins], n_total, n_mem, members,
                     unit_range=(1, 3))


@pytest.mark.parametrize("mode", ["h3", "f32"])
def test_stepper_raises_on_fp16_range_overflow(mode, monkeypatch):
    """|x| = 1e4 after normalisation on a network of the GENERIC width (E = 16: the tile GEMM kernels with the fixed x16
    pre-scale; the production width's fused encoder / decoder scale their tiles dynamically and take such inputs, see
    tests/test_gpu_sfno.py::test_network_inputs_of_any_magnitude_in_split_fp16_mode): the split-precision tile kernels
    cannot represent it (x16 -> fp16 overflow) and the window must fail loudly, naming the fp32 mode -- before anything
    reaches a writer; the fp32-MFMA path runs the same data."""
    import sdy_amd
    from sdy_amd._lib import SdyError

    monkeypatch.setenv("SDY_GEMM_MODE", mode)
    exp, _, cs, n_forc = _build(hack=True, dropout=False)
    assert exp.model.model.gemm_mode == mode
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]
    out_names, forcing_names = in_names[1:], ["f0", "f1"]
    names = in_names + forcing_names
    stepper = sdy_amd.MultiStepStepper(exp, names, out_names, forcing_names, {n: 0.0 for n in names},
                                       {n: 1.0 for n in names}, None)
    g = torch.Generator(device="cpu").manual_seed(3)
    data = {n: torch.randn(1, 7, 32, 64, generator=g).cuda() for n in names}
    sdy_amd.ops.status_flags(reset=True)
    out = stepper.run_on_batch(data, None, n_forward_steps=6)            # ordinary data: fine in both modes
    assert all(torch.isfinite(v).all() for v in out.gen_data.values())
    data["v1"][0, 0, 4, 9] = 1.0e4
    if mode == "h3":
        with pytest.raises(SdyError, match="SDY_GEMM_MODE=f32"):
            stepper.run_on_batch(data, None, n_forward_steps=6)
        assert sdy_amd.ops.status_flags(reset=True) == 0                 # the check consumed the flags
    else:
        out = stepper.run_on_batch(data, None, n_forward_steps=6)
        assert all(torch.isfinite(v).all() for v in out.gen_data.values())


@pytest.mark.parametrize("hack", [False, True])
def test_fused_interpolator_pair_equals_two_calls(hack):
    """A cold-sampling step interpolates the same (x_0, forecast) pair to two times (reference dyffusion.py:497 and :515).
    Small batches run both as ONE forward of 2B rows with per-row time and per-row dropout call number
    (`rows_per_call`): the samples must equal the two-call path bit for bit, dropout ON."""
    exp, oracle, cs, n_forc = _build(hack=hack, dropout=True)
    sampler, fnet, inet = exp.model, exp.model.model, exp.model.interpolator.model
    g = torch.Generator(device="cpu").manual_seed(77)
    B = 3
    x0 = torch.randn(B, cs, 32, 64, generator=g).cuda()
    kw = ({"static_condition": torch.randn(B, n_forc, 32, 64, generator=g).cuda()} if hack else
          {"dynamical_condition": torch.randn(B, 7, n_forc, 32, 64, generator=g).cuda()})
    exp.set_batch_offset(5)
    outs = []
    for limit in (0, 8):            # 0: never fuse; 8: fuse (B = 3)
        sampler.fuse_interpolator_pair_max_batch = limit
        fnet._call = inet._call = 0
        outs.append(sampler.sample(x0, **kw))
        assert (fnet._call, inet._call) == (6, 10)          # the fused forwards advance the call counter by two
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k}: fused pair differs from two calls"
    # and the per-row stream really is per call: rows B.. of a stacked forward differ from rows 0..B-1
    inet.enable_inference_dropout()
    inp = torch.randn(B, inet.num_input_channels, 32, 64, generator=g).cuda()
    c = torch.randn(B, n_forc, 32, 64, generator=g).cuda()
    t = torch.full((2 * B,), 2.0).cuda()
    y = inet(torch.cat([inp, inp]), time=t, static_condition=torch.cat([c, c]), rows_per_call=B)
    assert not torch.equal(y[:B], y[B:])
    inet.disable_inference_dropout()
    with pytest.raises(AssertionError):
        inet(torch.cat([inp, inp]), time=t, static_condition=torch.cat([c, c]), rows_per_call=4)


@pytest.mark.parametrize("shape", ["production", "generic"])
def test_stacked_calls_that_share_their_inputs_run_the_encoder_once(shape):
    """`forward(shared_inputs=True, rows_per_call=n)`: the stacked calls of a cold-sampling step read the same n input rows
    (reference call sites dyffusion.py:497,515: same x_0, forecast and static condition, other time and dropout call).  On the
    production shape (180 x 360, E = 256, equiangular data grid: fft360 + leg_par, fused encoder) the encoder runs on n rows
    and the first block's FFT reads its rows modulo n; other shapes fall back to encoding every row.  Either way the result
    equals the forward on stacked copies bit for bit, dropout and drop path on, and the stage timer sees the encoder's rows."""
    import sdy_amd
    from sdy_amd import synthetic

    g = torch.Generator(device="cpu").manual_seed(31)
    if shape == "production":
        nlat, nlon, embed = 180, 360, 256
    else:
        nlat, nlon, embed = 32, 64, 16
    net = synthetic.build_network(10, 5, 2, nlat=nlat, nlon=nlon, embed=embed, layers=2, dropout_mlp=0.1, drop_path_rate=0.2,
                                  time_range=(1.0, 5.0))
    n = 3
    x = torch.randn(n, 10, nlat, nlon, generator=g).cuda()
    c = torch.randn(n, 2, nlat, nlon, generator=g).cuda()
    t = torch.tensor([2.0] * n + [1.0] * n).cuda()
    net.enable_inference_dropout()
    net.batch_offset, net._call = 4, 10
    with sdy_amd.ops.stage_timer() as tm:
        a = net(torch.cat([x, x]), time=t, static_condition=torch.cat([c, c]), rows_per_call=n)
    net._call = 10
    with sdy_amd.ops.stage_timer() as ts:
        b = net(x, time=t, static_condition=c, rows_per_call=n, shared_inputs=True)
    assert net._call == 12 and a.shape == b.shape == (2 * n, 5, nlat, nlon)
    assert torch.equal(a, b), f"shared inputs differ from stacked copies: {float((a - b).abs().max()):.3e}"
    assert not torch.equal(b[:n], b[n:])                       # the two calls are different samples at different times
    if shape == "production":
        assert tm.rows["encoder (fused pair)"] == 2 * n and ts.rows["encoder (fused pair)"] == n
    if shape == "production":     # a forward that encoded the shared rows only cannot serve a later reuse_encoder forward
        with pytest.raises(sdy_amd.SdyError):
            net(torch.cat([x, x]), time=t, static_condition=torch.cat([c, c]), rows_per_call=n, reuse_encoder=True)
    net.disable_inference_dropout()


def test_stepper_autoregressive_init_handoff():
    """use_cold_sampling_for_last_step = False (dyffusion.py:503-510): the last prediction of a window is the plain forecast,
    while the cold-sampled state is handed to the next window as `preds_autoregressive_init` (stepper_multistep.py:412-418),
    prescribed like the prediction.  Two windows, so the handed-over state matters; vs the oracle stepper + sampler."""
    import sdy_amd
    from oracle.stepper import run_on_batch

    C, n_forc, hz = 6, 2, 6
    fcfg = SFNOConfig(in_chans=C + n_forc, out_chans=C, nlat=32, nlon=64, embed_dim=16, num_layers=2, with_time_emb=True,
                      min_time=0.0, max_time=hz - 1.0)
    icfg = SFNOConfig(in_chans=2 * C + n_forc, out_chans=C, nlat=32, nlon=64, embed_dim=16, num_layers=2, with_time_emb=True,
                      min_time=1.0, max_time=hz - 1.0)
    fnet, fora, _ = make_pair(fcfg, C, n_forc, seed=11)
    inet, iora, _ = make_pair(icfg, 2 * C, n_forc, seed=22)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=hz), horizon=hz,
        diffusion_config=dict(use_cold_sampling_for_last_step=False, use_cold_sampling_for_init_of_ar_step=True,
                              enable_interpolator_dropout=False))
    oracle = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(x, time=time, condition=condition,
                                                                                          static_condition=static_condition),
                             lambda x, time, condition=None, static_condition=None: iora(x, time=time, condition=condition,
                                                                                          static_condition=static_condition),
                             timesteps=hz, use_cold_sampling_for_last_step=False, use_cold_sampling_for_init_of_ar_step=True)
    names = [f"v{i}" for i in range(C)]
    forcing = ["f0", "f1"]
    g = torch.Generator(device="cpu").manual_seed(8)
    B, T1 = 2, 13
    means = {n: 0.2 * i for i, n in enumerate(names + forcing)}
    stds = {n: 1.0 + 0.1 * i for i, n in enumerate(names + forcing)}
    data = {n: torch.randn(B, T1, 32, 64, generator=g) * stds[n] + means[n] for n in names + forcing}
    data["frac"] = torch.rand(B, T1, 32, 64, generator=g)
    pres = dict(prescribed_name="v2", mask_name="frac", mask_value=1, interpolate=True)

    class OMod:   # the reference module's surface around the oracle sampler: forcings ride as the dynamical condition
        true_horizon = hz
        from contextlib import nullcontext
        ema_scope = inference_dropout_scope = staticmethod(nullcontext)

        def __init__(self):
            self.cache = None

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            if horizon == 1:
                self.cache = oracle.sample(batch["dynamics"], static_condition=batch.get("static_condition"))
            out = {f"t{horizon}_preds_normed": self.cache[f"t{horizon}_preds"]}
            if horizon == hz:
                out["preds_autoregressive_init_normed"] = self.cache["preds_autoregressive_init"]
            return out

    # without the input-only HGTsfc channel the stepper passes no forcings (stepper_multistep.py:383-384): a network without
    # conditioning would be needed; keep the forcings by driving both sides through a thin wrapper that adds them
    class WithForcing:
        def __init__(self, inner):
            self.inner, self.true_horizon = inner, inner.true_horizon
            self.model = inner.model if hasattr(inner, "model") else None
            self.ema_scope, self.inference_dropout_scope = inner.ema_scope, inner.inference_dropout_scope
            self.forc = None

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            batch = dict(batch, static_condition=self.forc.to(batch["dynamics"].device))
            return self.inner.get_preds_at_t_for_batch(batch, horizon=horizon, **kw)

    forc0 = torch.stack([(data[n][:, 0] - means[n]) / stds[n] for n in forcing], dim=1)
    om, pm = WithForcing(OMod()), WithForcing(exp)
    om.forc = pm.forc = forc0
    tm = {k: torch.tensor(v) for k, v in means.items()}
    ts = {k: torch.tensor(v) for k, v in stds.items()}
    metrics, gen, _ = run_on_batch(data, om, names, names, [], tm, ts, T1 - 1, pres, hack=False)
    stepper = sdy_amd.MultiStepStepper(pm, names, names, [], means, stds, sdy_amd.Prescriber(**pres))
    out = stepper.run_on_batch({k: v.cuda() for k, v in data.items()}, None, n_forward_steps=T1 - 1)
    for n in names:
        assert rel_l2(out.gen_data[n], gen[n]) < TOL, n
    # the handed-over state is not the prediction: feeding the prediction back instead gives another second window
    class NoHandoff(WithForcing):
        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            r = super().get_preds_at_t_for_batch(batch, horizon, **kw)
            r.pop("preds_autoregressive_init_normed", None)
            return r
    nh = NoHandoff(exp)
    nh.forc = forc0
    out2 = sdy_amd.MultiStepStepper(nh, names, names, [], means, stds, sdy_amd.Prescriber(**pres)).run_on_batch(
        {k: v.cuda() for k, v in data.items()}, None, n_forward_steps=T1 - 1)
    assert rel_l2(out2.gen_data["v0"][:, :7], gen["v0"][:, :7]) < TOL
    assert rel_l2(out2.gen_data["v0"][:, 7:], gen["v0"][:, 7:]) > 1e-3


def test_ensemble_statistics_match_torch_drawn_dropout():
    """Distributional parity of the stochastic interpolator: the product draws its dropout / drop-path
    decisions from its own Philox stream, the reference from torch's generator (src/models/sfno/layers.py:76-78,
    src/models/modules/drop_path.py:15-22, enabled at inference by src/diffusion/dyffusion.py:226-235) -- the two can only
    agree in distribution.  One horizon-6 sampling pass, 25 members of ONE initial condition:
      product : 25 rows, dropout stream of trajectory b = row b (batch_offset 0), the HIP path;
      oracle  : the same pass with Bernoulli(1 - p) masks drawn by torch's CPU generator, independently per member.
    Per (lead time, channel, pixel) the two 25-member ensembles give a mean and a spread; if both sample the same
    distribution,  z = (mean_p - mean_o) / sqrt((s_p^2 + s_o^2) / 25)  is ~ N(0, 1) and log(s_p^2 / s_o^2) has mean ~ 0 and
    standard deviation ~ sqrt(4 / 24) = 0.41.  Bounds (pixels are spatially correlated, so they are loose, but a keep rate of
    0.8 instead of 0.9, a missing 1 / (1 - p), members sharing a stream, or a dead drop path all violate them -- checked
    below by running the statistics on deliberately wrong ensembles)."""
    M, horizon = 25, 6
    exp, oracle, cs, n_forc = _build(hack=True, dropout=True)
    g = torch.Generator(device="cpu").manual_seed(321)
    x0 = torch.randn(1, cs, 32, 64, generator=g).expand(M, -1, -1, -1).contiguous()
    forc = torch.randn(1, n_forc, 32, 64, generator=g).expand(M, -1, -1, -1).contiguous()
    exp.set_batch_offset(0)
    got = exp.model.sample(x0.cuda(), static_condition=forc.cuda())

    icfg = oracle._masks.cfg
    tg = torch.Generator(device="cpu").manual_seed(2024)

    def torch_masks(kind, layer, shape):          # what nn.Dropout / DropPath draw: independent Bernoulli keeps
        if kind == "drop_path":
            rate = icfg.drop_path_rates[layer]
            return (torch.rand(shape[0], generator=tg) >= rate).float().reshape(-1, 1, 1, 1)
        return (torch.rand(shape, generator=tg) >= icfg.dropout_mlp).float()

    _, fora, _ = make_pair(SFNOConfig(in_chans=cs + n_forc, out_chans=6, nlat=32, nlon=64, embed_dim=16, num_layers=2,
                                      with_time_emb=True, min_time=0.0, max_time=5.0), cs, n_forc, seed=11)
    _, iora, _ = make_pair(icfg, 2 * cs, n_forc, seed=22)

    def sample_with(mask_fn):
        o = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(
            x, time=time, condition=condition, static_condition=static_condition),
            lambda x, time, condition=None, static_condition=None: iora(
                x, time=time, condition=condition, static_condition=static_condition, mask_fn=mask_fn),
            timesteps=horizon, hack_for_imprecise_interpolation=True)
        return o.sample(x0, static_condition=forc)

    ref = sample_with(torch_masks)

    def stats(a, b):
        """a, b: (M, C, H, W) ensembles -> (mean z, std z, share of |z| > 3, mean log variance ratio)"""
        a, b = a.double().cpu(), b.double().cpu()
        va, vb = a.var(0, unbiased=True), b.var(0, unbiased=True)
        z = (a.mean(0) - b.mean(0)) / ((va + vb) / M).sqrt().clamp_min(1e-30)
        lr = (va.clamp_min(1e-30) / vb.clamp_min(1e-30)).log()
        return float(z.mean()), float(z.std()), float((z.abs() > 3).double().mean()), float(lr.mean())

    for k in ("t1_preds", "t3_preds", "t6_preds"):
        zm, zs, tail, lr = stats(got[k], ref[k])
        spread = float(got[k].double().std(0).mean())
        assert spread > 1e-3, f"{k}: the members do not diverge (spread {spread:.2e})"
        assert abs(zm) < 0.15, f"{k}: ensemble means differ, mean z {zm:.3f}"
        assert 0.75 < zs < 1.3, f"{k}: z-scores not unit-width, std {zs:.3f}"
        assert tail < 0.03, f"{k}: {tail:.3%} of the z-scores beyond 3 sigma"
        assert abs(lr) < 0.15, f"{k}: ensemble spreads differ, mean log variance ratio {lr:.3f}"

    # the statistic has teeth: ensembles that are wrong in the ways an implementation can be wrong fail the same bounds
    def wrong_keep(kind, layer, shape):           # keep rate 0.8 instead of 0.9 (element dropout only)
        if kind == "drop_path":
            return torch_masks(kind, layer, shape)
        return (torch.rand(shape, generator=tg) >= 0.2).float()

    def shared_stream(kind, layer, shape):        # every member draws the SAME element masks
        if kind == "drop_path":
            return torch_masks(kind, layer, shape)
        m = (torch.rand((1,) + tuple(shape[1:]), generator=tg) >= icfg.dropout_mlp).float()
        return m.expand(shape).contiguous()

    zm, zs, tail, lr = stats(sample_with(wrong_keep)["t6_preds"], ref["t6_preds"])
    assert abs(lr) > 0.15 or abs(zm) > 0.15 or not (0.75 < zs < 1.3), ("a wrong keep rate passes the bounds", zm, zs, lr)
    zm, zs, tail, lr = stats(sample_with(shared_stream)["t6_preds"], ref["t6_preds"])
    assert abs(lr) > 0.15, ("members sharing their element masks pass the spread bound", lr)


def test_predict_step_and_interface_run_inference():
    """The Lightning-driven entry without Lightning (reference: src/interface.py:302-313 -> trainer.predict ->
    _base_experiment.py:1083-1102 -> forecasting_multi_horizon.py:139-320): `predict_step` over two autoregressive windows
    equals the oracle's sampler chained by hand; with `num_predictions` = 3 the members are batched on the device, come back
    as (N, B, ...) and differ through their dropout streams; `interface.run_inference` drives a datamodule's batches and
    concatenates them along the batch axis."""
    import numpy as np

    import sdy_amd

    exp, oracle, cs, n_forc = _build(hack=False, dropout=False)
    g = torch.Generator(device="cpu").manual_seed(77)
    B, H, PH = 2, 6, 12
    dynamics = torch.randn(B, 1 + PH, cs, 32, 64, generator=g)
    dcond = torch.randn(B, PH + 1, n_forc, 32, 64, generator=g)
    out = exp.predict_step({"dynamics": dynamics.cuda(), "dynamical_condition": dcond.cuda()}, 0, prediction_horizon=PH)
    assert sorted(k for k in out if "preds" in k) == sorted(f"t{k}_preds_normed" for k in range(1, PH + 1))
    first = oracle.sample(dynamics[:, 0], dynamical_condition=dcond[:, :H + 1])
    second = oracle.sample(first[f"t{H}_preds"], dynamical_condition=dcond[:, H:2 * H + 1])
    for k in range(1, PH + 1):
        ref = first[f"t{k}_preds"] if k <= H else second[f"t{k - H}_preds"]
        assert isinstance(out[f"t{k}_preds_normed"], np.ndarray) and out[f"t{k}_preds_normed"].shape == (B, 6, 32, 64)
        assert rel_l2(torch.from_numpy(out[f"t{k}_preds_normed"]), ref) < 2e-5, f"t{k}"
        assert np.array_equal(out[f"t{k}_targets_normed"], dynamics[:, k].numpy())
    exp.on_predict_epoch_end()

    # ensemble: members batched, (N, B, ...) results, dropout streams per trajectory
    exp, _, cs, n_forc = _build(hack=False, dropout=True)
    exp.num_predictions = 3
    res = exp.predict_step({"dynamics": dynamics.cuda(), "dynamical_condition": dcond.cuda()}, 0, prediction_horizon=PH)
    p = res["t12_preds_normed"]
    assert p.shape == (3, B, 6, 32, 64) and np.isfinite(p).all()
    assert np.abs(p[0] - p[1]).max() > 1e-3 and res["t12_targets_normed"].shape == (B, 6, 32, 64)
    exp.on_predict_epoch_end()

    class DM:          # what run_inference needs of a LightningDataModule
        def setup(self, stage):
            self.stage = stage

        def predict_dataloader(self):
            for i in range(2):
                yield {"dynamics": dynamics[i:i + 1, :1 + H], "dynamical_condition": dcond[i:i + 1, :H + 1]}

    exp.set_dropout_calls((0, 0))
    merged = sdy_amd.interface.run_inference(exp, DM())
    assert merged["t6_preds_normed"].shape == (3, 2, 6, 32, 64)          # batches concatenated on axis 1 (members lead)
    assert merged["t6_targets_normed"].shape == (2, 6, 32, 64)
    assert exp._predict_step_outputs == []


@pytest.mark.parametrize("hack", [False, True])
def test_interpolator_pair_with_shared_encoder_equals_two_full_forwards(hack):
    """Beyond `fuse_interpolator_pair_max_batch` the two interpolations of a cold-sampling step (reference
    src/diffusion/dyffusion.py:497,515: same x_0 and forecast, times s' and s) run as two forwards, the second restarting
    from the first one's encoder output (`sdy_sfno_fwd_args.reuse_encoder`): bit-identical to two full forwards, 10
    interpolator calls either way; a time-dependent condition switches the sharing off; restarting without a previous
    forward is an error."""
    import sdy_amd

    exp, oracle, cs, n_forc = _build(hack=hack, dropout=True)
    smp = exp.model
    smp.fuse_interpolator_pair_max_batch = 0          # force the two-forward path at this small batch
    g = torch.Generator(device="cpu").manual_seed(5)
    B = 3
    x0 = torch.randn(B, cs, 32, 64, generator=g).cuda()
    kw = {"static_condition": torch.randn(B, n_forc, 32, 64, generator=g).cuda()}
    inet = smp.interpolator.model
    seen = []
    orig = inet._native_call

    def spy(*a, reuse_encoder=False, **k):
        seen.append(bool(reuse_encoder))
        return orig(*a, reuse_encoder=reuse_encoder, **k)

    inet._native_call = spy
    smp.reuse_interpolator_encoder = True
    shared = smp.sample(x0, **kw)
    assert len(seen) == 10
    n_shared = sum(seen)                              # every cold step with s > 0 shares: the second call of each pair
    assert all(not a or not b for a, b in zip(seen, seen[1:])), "a restart follows a full forward"
    exp.set_dropout_calls((0, 0))
    seen.clear()
    smp.reuse_interpolator_encoder = False
    plain = smp.sample(x0, **kw)
    assert len(seen) == 10 and not any(seen)
    assert n_shared >= 4
    for k in plain:
        assert torch.equal(shared[k], plain[k]), k
    ref = oracle.sample(x0.cpu(), **{k: v.cpu() for k, v in kw.items()})
    for k in ref:
        assert rel_l2(shared[k], ref[k]) < 2e-5
    # a time-dependent condition differs between the two calls: no sharing
    if not hack:
        exp.set_dropout_calls((0, 0))
        seen.clear()
        smp.reuse_interpolator_encoder = True
        dyn = torch.randn(B, 7, n_forc, 32, 64, generator=g).cuda()
        smp.sample(x0, dynamical_condition=dyn)
        assert len(seen) == 10 and not any(seen)
    # the native call refuses to restart when there is nothing to restart from
    net2, _, _ = make_pair(SFNOConfig(in_chans=4, out_chans=4, nlat=32, nlon=64, embed_dim=16, num_layers=1,
                                      with_time_emb=False), 4, 0)
    with pytest.raises(sdy_amd.SdyError):
        net2(torch.zeros(1, 4, 32, 64).cuda(), reuse_encoder=True)


class _MailboxComm:
    """RelayComm stand-in for ranks played one after the other in ONE process (the hosts of a single relay trajectory follow
    the rank order, so every state is in the box before its receiver runs); the transport itself -- torch.distributed send /
    recv with the store handshake -- is covered by tests/test_distributed_cpu.py."""

    def __init__(self, box):
        self.box = box

    def send(self, task, state):
        self.box[(task.unit, task.w_end)] = state.clone()

    def ready(self, task, like):
        return (task.unit, task.w_begin) in self.box

    def recv(self, task, like):
        return self.box.pop((task.unit, task.w_begin))

    def finish(self):
        pass


def test_relayed_remainder_member_equals_the_unsharded_ensemble():
    """ensemble.relay_plan / run_relay with the REAL sampler (dropout and drop path on): 7 members x 6 windows as one batch of
    seven, and as three ranks' plans -- two resident members each and member 6 relayed through the ranks in slices of two
    windows, every pass keyed by (global trajectory, window) through set_batch_offset / set_dropout_calls (what bench.py's
    strong-scaling leg does on N GPUs).  The ranks are played one after the other in this process with a mailbox for the
    hand-overs (the host order of a single relay trajectory is the rank order, so nothing waits); the transport itself is
    covered with real send / recv by tests/test_distributed_cpu.py.  Every trajectory must equal its row of the batch of
    seven at every window (2e-5: the InstanceNorm sums are accumulated in an order that depends on the batch)."""
    from sdy_amd import ensemble

    exp, _, cs, n_forc = _build(hack=False, dropout=True)
    g = torch.Generator(device="cpu").manual_seed(99)
    n_units, n_windows, world = 7, 6, 3
    x0 = torch.randn(1, cs, 32, 64, generator=g).expand(n_units, -1, -1, -1).contiguous().cuda()
    forc = torch.randn(1, n_forc, 32, 64, generator=g).cuda()

    def one_pass(x, first_unit, w):
        exp.set_batch_offset(first_unit)
        exp.set_dropout_calls((6 * w, 10 * w))
        out = None
        for h in range(1, 7):
            batch = {"dynamics": x, "static_condition": forc.expand(x.shape[0], -1, -1, -1).contiguous()}
            with exp.ema_scope(), exp.inference_dropout_scope():
                out = exp.get_preds_at_t_for_batch(batch, horizon=h, split="predict", is_autoregressive=False,
                                                   prepare_inputs=False, ensemble=False, num_predictions=1)
        assert exp.dropout_calls() == (6 * (w + 1), 10 * (w + 1))
        return out["t6_preds_normed"]

    full, x = [], x0
    for w in range(n_windows):
        x = one_pass(x, 0, w)
        full.append(x.clone())
    assert float((full[-1][0] - full[-1][1]).abs().max()) > 1e-3          # members are different samples

    mailbox, seen = {}, {}
    for rank in range(world):
        plan = ensemble.relay_plan(n_units, world, n_windows, rank)
        assert plan.count == 2 and len(plan.tasks) == 1 and plan.tasks[0].unit == 6
        state = {"res": x0[plan.start:plan.start + plan.count].clone()}

        def resident_step(w):
            state["res"] = one_pass(state["res"], plan.start, w)
            for r in range(plan.count):
                seen[(plan.start + r, w)] = state["res"][r:r + 1].clone()

        def relay_step(task, w, xs):
            xs = one_pass(xs, task.unit, w)
            seen[(task.unit, w)] = xs.clone()
            return xs

        finals = ensemble.run_relay(plan, n_windows, resident_step, relay_step, lambda u: x0[u:u + 1].clone(),
                                    _MailboxComm(mailbox), like=lambda task: x0[:1])
        assert (rank == world - 1) == (6 in finals)
    assert not mailbox and sorted(seen) == [(u, w) for u in range(n_units) for w in range(n_windows)]
    for (u, w), v in seen.items():
        e = rel_l2(v[0], full[w][u])
        assert e < 2e-5 * (4 ** w), f"trajectory {u}, window {w}: rel L2 {e:.3e}"


def test_window_driver_relays_the_remainder_trajectory():
    """The relay in the PRODUCT driver: `run_inference(relay=ensemble.relay_plan(...))` -> stepper -> sampler (dropout and drop
    path on) -> SFNO, 7 members of one initial condition over 6 windows of 6 steps, as the unsharded job and as three ranks'
    plans (two resident members each, member 6 relayed through the ranks in slices of two windows: the 25-over-8 schedule in
    miniature).  Every prediction any rank's writer receives -- resident rows and relay rows, window by window -- must be the
    unsharded job's field of that (trajectory, time step); every (trajectory, window) is written exactly once; and the ranks'
    time-mean sums add up to the unsharded aggregator's.  Ranks are played one after the other with a mailbox for the
    hand-overs; tools/c4_rollout.py --gpus N runs the same call over torch.distributed (tests/test_gpu_fullsize.py)."""
    import sdy_amd
    from sdy_amd import ensemble, synthetic

    dev = torch.device("cuda", 0)
    n_out, n_forc, nlat, nlon, window, n_windows, members, world = 4, 2, 32, 64, 6, 6, 7, 3
    exp, _, _ = synthetic.build_sampler(dev, state_chans=n_out, forcing_chans=n_forc, nlat=nlat, nlon=nlon, embed=16, layers=2,
                                        horizon=6, carried_input_only_channel=True)
    stepper, names, out_names = synthetic.build_stepper(exp, n_out, n_forc, carried_input_only_channel=True)
    steps = window * n_windows
    area = sdy_amd.metrics.spherical_area_weights(torch.linspace(-89.5, 89.5, nlat), nlon)

    class Writer:
        def __init__(self):
            self.fields, self.targets = {}, {}

        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            v = prediction[out_names[1]]
            if v.dim() == 5:                       # (members, n_sample, time, H, W): the unsharded presentation
                v = v[:, 0]
            for r in range(v.shape[0]):
                for t in range(v.shape[1]):
                    key = (start_sample + r, start_timestep + t)
                    assert key not in self.fields, f"trajectory / time step {key} written twice"
                    self.fields[key] = v[r, t].detach().cpu().clone()

    keyed = {}       # (first global trajectory of a device batch, rows) -> dropout call counters it started from, per job

    run_on_batch = stepper.run_on_batch

    def logged(data, *a, **k):
        rows = next(iter(data.values())).shape[0]
        keyed.setdefault("log", []).append((exp.model.interpolator.model.batch_offset, rows, exp.dropout_calls()))
        return run_on_batch(data, *a, **k)

    stepper.run_on_batch = logged

    def job(**kw):
        exp.set_dropout_calls((0, 0))
        wr = Writer()
        agg = sdy_amd.metrics.TimeMeanAggregator(area, is_ensemble=True)
        t = sdy_amd.run_inference(agg, stepper, synthetic.windows(names, n_windows, window, nlat, nlon, seed=5), steps, window,
                                  n_ensemble_members=members, eval_device=dev, writer=wr, **kw)
        return wr, agg, t

    full, agg_full, _ = job()
    assert len(full.fields) == members * (steps + 1)
    per_window = {c for _, _, c in keyed.pop("log")}
    assert len(per_window) == n_windows
    c0, c1 = sorted(per_window)[:2]
    delta = (c1[0] - c0[0], c1[1] - c0[1])                   # dropout calls of one window (forecaster, interpolator)
    assert c0 == (0, 0) and delta[1] > 0
    box, seen, gen_sum, gen_rows = {}, {}, None, 0.0
    for rank in range(world):
        plan = ensemble.relay_plan(members, world, n_windows, rank)
        assert plan.count == 2 and [t.unit for t in plan.tasks] == [6]
        # (the driver's other options ride along: device batches of one row, the synchronous loop with pinned-host outputs)
        opts = [{}, {"max_batch": 1}, {"prefetch": 0, "host_outputs": True}][rank]
        wr, agg, timers = job(relay=plan, relay_comm=_MailboxComm(box), **opts)
        assert timers["forecast_steps_per_second"] > 0 and timers["trajectory_steps"] == (2 * n_windows + 2) * window
        assert not (set(wr.fields) & set(seen))
        seen.update(wr.fields)
        # The stream position of every device batch, bit for bit (not through chain-amplified states): a batch of window w
        # starts from the unsharded job's call numbers of window w, whichever rank runs it and in whatever order -- resident
        # rows keyed by the block's first trajectory, the relay trajectory by its own index.
        log = keyed.pop("log")
        res = [e for e in log if e[0] != 6]
        rel = [e for e in log if e[0] == 6]
        assert [c for _, _, c in res if True][::(2 if opts.get("max_batch") else 1)] == \
            [(w * delta[0], w * delta[1]) for w in range(n_windows)]
        assert [c for _, _, c in rel] == [(w * delta[0], w * delta[1]) for w in range(plan.tasks[0].w_begin, plan.tasks[0].w_end)]
        assert all(rows == 1 for _, rows, _ in rel) and {off for off, _, _ in res} <= {plan.start, plan.start + 1}
        gen_sum = agg._gen_data[out_names[1]].double() if gen_sum is None else gen_sum + agg._gen_data[out_names[1]].double()
        gen_rows += agg._gen_rows
    assert not box
    assert sorted(seen) == sorted(full.fields)                       # every (trajectory, time step) exactly once
    worst = 0.0
    for (u, t), v in seen.items():
        e = rel_l2(v, full.fields[(u, t)])
        worst = max(worst, e)
        assert e < 2e-5 * (4 ** (t // window)), f"trajectory {u}, time step {t}: rel L2 {e:.3e}"
    assert gen_rows == agg_full._gen_rows
    assert rel_l2(gen_sum, agg_full._gen_data[out_names[1]].double()) < 1e-4
