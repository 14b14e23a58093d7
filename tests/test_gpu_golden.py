"""HIP path vs golden vectors produced by the REFERENCE'S OWN classes (tests/golden/*.npz, tools/gen_golden.py).

This is the direct parity claim: same weights and inputs -> the native library reproduces the reference fields to
<= 1e-4 relative L2 (north_star bound); measured errors are ~1e-6."""
import json

import pytest
import torch

import golden_utils as gu
from conftest import rel_l2
from oracle.sfno import SFNOConfig

pytestmark = pytest.mark.gpu
TOL = 1e-4
TOL_TIGHT = 2e-5


def _t(z, k):
    return torch.from_numpy(z[k]) if k in z.files else None


def _cu(t):
    return None if t is None else t.cuda()


def _net(cfg, n_in, n_cond, sd, seed=0):
    import sdy_amd

    net = sdy_amd.SphericalFourierNeuralOperatorNet(
        num_input_channels=n_in, num_output_channels=cfg.out_chans, num_conditional_channels=n_cond,
        spatial_shape_in=(cfg.nlat, cfg.nlon), embed_dim=cfg.embed_dim, num_layers=cfg.num_layers,
        mlp_ratio=cfg.mlp_ratio, dropout_mlp=cfg.dropout_mlp, drop_path_rate=cfg.drop_path_rate,
        with_time_emb=cfg.with_time_emb, data_grid=cfg.data_grid, big_skip=cfg.big_skip, pos_embed=cfg.pos_embed,
        seed=seed)
    net.load_state_dict(sd, strict=True)
    if cfg.with_time_emb:
        net.set_min_max_time(cfg.min_time, cfg.max_time)
    return net


def _injector(per_fwd, cfg, first_call=0):
    """mask_injector for the product network from the reference's recorded masks."""
    L = cfg.num_layers

    def inj(call):
        d = per_fwd[call - first_call]
        B = d[("mlp_hidden", 0)].shape[0]
        km = []
        for i in range(L):
            km += [d[("mlp_hidden", i)], d[("mlp_out", i)]]
        dpk = torch.ones(L, B)
        for i in range(L):
            if ("drop_path", i) in d:
                dpk[i] = d[("drop_path", i)].reshape(-1)
        return km, dpk
    return inj


@pytest.mark.parametrize("name", ["fx_block_c1", "fx_sfno_tiny", "fx_sfno_tiny_lg"])
def test_network_vs_reference(name):
    z = gu.load(name)
    cfg, n_in, n_cond = gu.cfg_from(z)
    net = _net(cfg, n_in, n_cond, gu.state_dict(z))
    y = net(_cu(_t(z, "x")), time=_cu(_t(z, "time")), condition=_cu(_t(z, "cond")))
    err = rel_l2(y, _t(z, "y"))
    assert err < TOL_TIGHT, f"{name}: rel L2 {err:.3e}"
    if "t_repr" in z.files:
        trep, _ = net.time_embedding(_cu(_t(z, "time")))
        assert rel_l2(trep, _t(z, "t_repr")) < 2e-6


def test_network_vs_reference_with_recorded_dropout():
    z = gu.load("fx_sfno_tiny")
    cfg, n_in, n_cond = gu.cfg_from(z)
    net = _net(cfg, n_in, n_cond, gu.state_dict(z))
    net.mask_injector = _injector(gu.masks_per_forward(gu.recorded_masks(z), cfg), cfg)
    net.enable_inference_dropout()
    y = net(_cu(_t(z, "x")), time=_cu(_t(z, "time")), condition=_cu(_t(z, "cond")))
    err = rel_l2(y, _t(z, "y_dropout"))
    assert err < TOL_TIGHT, f"rel L2 {err:.3e}"


def test_full_size_network_vs_reference():
    """The HIP path against the REFERENCE network's own output at production size -- 180 x 360, E = 256, hidden 512, all 8
    blocks, 68 + 2 -> 34 channels (BASELINE.json configs[1]; tests/golden/fx_sfno_full.npz, written by tools/gen_golden.py
    from src/models/sfno/sfnonet.py:797-841 run on CPU).  No oracle in between: reference vectors on `mlp_h3`, `conv_h3`,
    `dh_h3`, `leg_par`, `fft360` and `pair_h3`."""
    z = gu.load("fx_sfno_full")
    cfg, n_in, n_cond, sd, x, cond, t = gu.seeded_case(z)
    net = _net(cfg, n_in, n_cond, sd)
    y = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    assert torch.isfinite(y).all()
    err = rel_l2(y, _t(z, "y"))
    assert err < TOL, f"full size vs reference: rel L2 {err:.3e} (north_star bound 1e-4)"
    assert err < TOL_TIGHT, f"full size vs reference: rel L2 {err:.3e} (fp32 expectation)"
    # and per output channel: no field hides behind the others' norm
    ref = _t(z, "y")
    worst = max(rel_l2(y[:, c], ref[:, c]) for c in range(ref.shape[1]))
    assert worst < TOL, f"worst output channel rel L2 {worst:.3e}"


def test_wide_network_vs_reference_with_recorded_dropout():
    """E = 256 / hidden 512 (the fused MLP kernel's shape; small grid) with the masks the REFERENCE's nn.Dropout / DropPath
    layers drew (src/models/sfno/layers.py:76-78, src/models/modules/drop_path.py:15-22) injected: they drive the fused
    `mlp_h3` kernel (its INJECT instantiation: same code, mask tensors instead of the Philox stream) -- reference-generated
    vectors on the dropout path of the production MLP kernel, no builder-defined stream on either side."""
    z = gu.load("fx_sfno_wide_masks")
    cfg, n_in, n_cond, sd, x, cond, t = gu.seeded_case(z)
    net = _net(cfg, n_in, n_cond, sd)
    y = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    assert rel_l2(y, _t(z, "y")) < TOL_TIGHT
    net.mask_injector = _injector(gu.masks_per_forward(gu.recorded_masks(z), cfg), cfg, first_call=net._call)
    net.enable_inference_dropout()
    yd = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    err = rel_l2(yd, _t(z, "y_dropout"))
    assert err < TOL_TIGHT, f"recorded masks through the fused MLP: rel L2 {err:.3e}"
    assert rel_l2(_t(z, "y_dropout"), _t(z, "y")) > 1e-2      # the masks matter


@pytest.mark.parametrize("name", ["fx_sample_tiny", "fx_sample_tiny_hack", "fx_sample_tiny_masks", "fx_sample_tiny_refine"])
def test_sampler_vs_reference(name):
    import sdy_amd

    z = gu.load(name)
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    hack, dropout = bool(int(z["hack"])), bool(int(z["dropout"]))
    n_forc = 2
    fnet = _net(fcfg, fcfg.in_chans - n_forc, n_forc, gu.state_dict(z, "f::"))
    inet = _net(icfg, icfg.in_chans - n_forc, n_forc, gu.state_dict(z, "i::"))
    if dropout:
        inet.mask_injector = _injector(gu.masks_per_forward(gu.recorded_masks(z), icfg), icfg)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=6), horizon=6,
        diffusion_config=dict(hack_for_imprecise_interpolation=hack, enable_interpolator_dropout=dropout,
                              **(json.loads(str(z["diffusion_extra"])) if "diffusion_extra" in z.files else {})))
    kw = {k: _cu(_t(z, k)) for k in ("dynamical_condition", "static_condition") if k in z.files}
    out = exp.model.sample(_cu(_t(z, "x0")), **kw)
    assert fnet._call + inet._call == len(json.loads(str(z["trace"])))       # 16 network calls (21 with the refining sweep)
    ref = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    assert sorted(out) == sorted(ref)
    for k in ref:
        err = rel_l2(out[k], ref[k])
        assert err < TOL, f"{name}/{k}: rel L2 {err:.3e}"
        assert err < TOL_TIGHT, f"{name}/{k}: rel L2 {err:.3e}"


@pytest.mark.parametrize("name", ["fx_sample_tiny_k2", "fx_sample_tiny_k2_every2nd", "fx_sample_tiny_naive"])
def test_sampler_outside_k0_vs_reference(name):
    """The product sampler OUTSIDE the shipped k = 0 configuration against the reference's own `DYffusion.sample`
    (src/diffusion/dyffusion.py:134-188 step map, :226-235 per-call dropout rule, :367-455 named schedules, :467-520 loop):
    additional_interpolation_steps = 2 with the full schedule and with "every2nd", `enable_interpolator_dropout=
    "except_dynamical_steps"` (the reference's recorded masks are injected into exactly the calls whose dropout it had on),
    and `sampling_type="naive"` with one artificial step.  The call trace (network, time, dropout on) equals the reference's."""
    import sdy_amd

    z = gu.load(name)
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    extra = json.loads(str(z["diffusion_extra"]))
    per_call = extra.get("enable_interpolator_dropout") == "except_dynamical_steps"
    n_forc = 2
    fnet = _net(fcfg, fcfg.in_chans - n_forc, n_forc, gu.state_dict(z, "f::"))
    inet = _net(icfg, icfg.in_chans - n_forc, n_forc, gu.state_dict(z, "i::"))
    per_fwd = gu.masks_per_forward(gu.recorded_masks(z), icfg)
    used = {"n": 0}

    def inject(call):      # called only by forwards whose dropout is on: the recorded sets are consumed in that order
        d = per_fwd[used["n"]]
        used["n"] += 1
        return _injector([d], icfg, first_call=call)(call)

    inet.mask_injector = inject
    trace = []
    for tag, net in (("F", fnet), ("I", inet)):
        def hook(fwd, tag=tag, net=net):
            def run(inputs, time=None, **kw):
                trace.append([tag, float(time[0])] + ([bool(net.inference_dropout)] if (tag == "I" and per_call) else []))
                return fwd(inputs, time=time, **kw)
            return run
        net.forward = hook(net.forward)
    ipol = sdy_amd.InterpolationExperiment(inet, horizon=6)
    # (the experiment pins the interpolator's valid times to the data steps [1, 5] like the reference's; the fixture's
    #  interpolator had its range opened to [0, 5] for the artificial steps -- tools/gen_golden.py, build_experiments)
    inet.set_min_max_time(icfg.min_time, icfg.max_time)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, ipol, horizon=6,
        diffusion_config=dict(hack_for_imprecise_interpolation=True, **{"enable_interpolator_dropout": True, **extra}))
    out = exp.model.sample(_cu(_t(z, "x0")), static_condition=_cu(_t(z, "static_condition")))
    assert trace == json.loads(str(z["trace"])), "network call order / times / dropout flags differ from the reference's"
    assert used["n"] == len(per_fwd)
    ref = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    assert sorted(out) == sorted(ref)
    for k in ref:
        err = rel_l2(out[k], ref[k])
        assert err < TOL_TIGHT, f"{name}/{k}: rel L2 {err:.3e}"


def test_sampler_refuses_artificial_times_outside_the_interpolators_range():
    """An interpolator whose time range is the data steps [1, horizon - 1] (what `InterpolationExperiment` sets,
    src/experiment_types/interpolation.py:24-31) fails the network's range assert (sfnonet.py:780-782) when a sampler with
    artificial steps asks for time 1/3; here the same AssertionError comes from the host scalar, without a device sync."""
    import sdy_amd

    z = gu.load("fx_sample_tiny_k2")
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    fnet = _net(fcfg, fcfg.in_chans - 2, 2, gu.state_dict(z, "f::"))
    inet = _net(icfg, icfg.in_chans - 2, 2, gu.state_dict(z, "i::"))
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=6), horizon=6,       # (pins the range to [1, 5])
        diffusion_config=dict(hack_for_imprecise_interpolation=True, enable_interpolator_dropout=False,
                              additional_interpolation_steps=2))
    with pytest.raises(AssertionError, match="time must be in"):
        exp.model.sample(_cu(_t(z, "x0")), static_condition=_cu(_t(z, "static_condition")))


def test_stepper_vs_reference():
    """MultiStepStepper.run_on_batch vs the reference's own run_on_batch_multistep (normalise, pack, 8 autoregressive
    steps across a window boundary, prescriber, HGTsfc carry-over, denormalise, LpLoss metrics)."""
    import sdy_amd

    z = gu.load("fx_stepper_tiny")
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    names = {k: json.loads(str(z[k])) for k in ("in_names", "out_names", "forcing_names")}
    n_forc = len(names["forcing_names"])
    fnet = _net(fcfg, fcfg.in_chans - n_forc, n_forc, gu.state_dict(z, "f::"))
    inet = _net(icfg, icfg.in_chans - n_forc, n_forc, gu.state_dict(z, "i::"))
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=6), horizon=6,
        diffusion_config=dict(hack_for_imprecise_interpolation=True, enable_interpolator_dropout=False))
    pr = json.loads(str(z["prescriber"]))
    stepper = sdy_amd.MultiStepStepper(
        exp, names["in_names"] + names["forcing_names"], names["out_names"], names["forcing_names"],
        means={k[6:]: float(z[k]) for k in z.files if k.startswith("mean::")},
        stds={k[5:]: float(z[k]) for k in z.files if k.startswith("std::")},
        prescriber=sdy_amd.Prescriber(pr["prescribed_name"], pr["mask_name"], pr["mask_value"], pr["interpolate"]))
    data = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("data::")}
    n_steps = int(z["n_steps"])
    out = stepper.run_on_batch(data, None, n_forward_steps=n_steps)
    assert sorted(out.gen_data) == sorted(names["out_names"])
    for n in names["out_names"]:
        assert out.gen_data[n].shape == (2, n_steps + 1, 32, 64)
        e1 = rel_l2(out.gen_data_norm[n], torch.from_numpy(z["gen_norm::" + n]))
        e2 = rel_l2(out.gen_data[n], torch.from_numpy(z["gen::" + n]))
        assert e1 < TOL_TIGHT and e2 < TOL_TIGHT, f"{n}: {e1:.3e} {e2:.3e}"
    for k in z.files:
        if k.startswith("metric::"):
            assert abs(float(out.metrics[k[8:]]) - float(z[k])) < 1e-4 * max(1.0, abs(float(z[k]))), k
    # normalised targets are exact (same two fp32 operations as the reference)
    v = "v1"
    ref = (torch.from_numpy(z["data::" + v]) - float(z["mean::" + v])) / float(z["std::" + v])
    assert rel_l2(out.target_data_norm[v], ref) < 1e-6


def test_window_driver_vs_reference():
    """sdy_amd.run_inference (members batched on the device, carried state resident on the device) vs the reference's
    own run_inference + WindowStitcher: 2 windows x 6 steps, 2 samples, 2 members (fixture fx_loop_tiny)."""
    import types

    import sdy_amd

    z, zw = gu.load("fx_loop_tiny"), gu.load("fx_stepper_tiny")
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    names = {k: json.loads(str(z[k])) for k in ("in_names", "out_names", "forcing_names")}
    n_forc = len(names["forcing_names"])
    fnet = _net(fcfg, fcfg.in_chans - n_forc, n_forc, gu.state_dict(zw, "f::"))
    inet = _net(icfg, icfg.in_chans - n_forc, n_forc, gu.state_dict(zw, "i::"))
    exp = sdy_amd.MultiHorizonForecastingDYffusion(
        fnet, sdy_amd.InterpolationExperiment(inet, horizon=6), horizon=6,
        diffusion_config=dict(hack_for_imprecise_interpolation=True, enable_interpolator_dropout=False))
    pr = json.loads(str(z["prescriber"]))
    stepper = sdy_amd.MultiStepStepper(
        exp, names["in_names"] + names["forcing_names"], names["out_names"], names["forcing_names"],
        means={k[6:]: float(z[k]) for k in z.files if k.startswith("mean::")},
        stds={k[5:]: float(z[k]) for k in z.files if k.startswith("std::")},
        prescriber=sdy_amd.Prescriber(pr["prescribed_name"], pr["mask_name"], pr["mask_value"], pr["interpolate"]))
    series = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("series::")}
    n_total, n_mem, members = int(z["n_total"]), int(z["n_mem_steps"]), int(z["members"])
    windows = [types.SimpleNamespace(data={k: v[:, i * n_mem:(i + 1) * n_mem + 1] for k, v in series.items()}, times=None)
               for i in range(n_total // n_mem)]

    for host in (False, True):
        wcalls, acalls = [], []

        class W:
            def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
                wcalls.append((start_timestep, {k: v.clone() for k, v in prediction.items()},
                               {k: v.shape for k, v in target.items()}, next(iter(prediction.values())).device.type))

        class A:
            def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start=0):
                acalls.append((loss, i_time_start))

        timers = sdy_amd.run_inference(A(), stepper, types.SimpleNamespace(loader=windows), n_total, n_mem, members,
                                       writer=W(), host_outputs=host)
        assert {"data_loading", "run_on_batch", "writer_and_aggregator"} <= set(timers)
        assert [c[0] for c in wcalls] == [int(v) for v in z["starts"]]
        assert [c[1] for c in acalls] == [int(v) for v in z["i_time_starts"]]
        assert all(c[3] == ("cpu" if host else "cuda") for c in wcalls)
        for w, (_, pred, tshapes, _) in enumerate(wcalls):
            for n in names["out_names"]:
                want = torch.from_numpy(z[f"pred{w}::{n}"])
                assert pred[n].shape == want.shape
                assert rel_l2(pred[n], want) < TOL_TIGHT, (w, n)
            assert all(s[1] == want.shape[2] for s in tshapes.values())
        for (loss, _), want in zip(acalls, z["losses"]):
            assert abs(loss - float(want)) < 1e-4 * max(1.0, abs(float(want)))


def test_checkpoint_ingestion_reproduces_reference_stepper():
    """checkpoint.stepper_from_state on Lightning-shaped checkpoints (key layout + hyper-parameters recorded from the
    reference's experiments, fx_ckpt_layout.json; weights of fx_stepper_tiny): the loaded stepper reproduces the
    reference's run_on_batch_multistep outputs, from the raw weights and from the EMA shadows (raw weights corrupted)."""
    import os

    import sdy_amd

    z = gu.load("fx_stepper_tiny")
    with open(os.path.join(os.path.dirname(__file__), "golden", "fx_ckpt_layout.json")) as f:
        lay = json.load(f)
    fsd, isd = gu.state_dict(z, "f::"), gu.state_dict(z, "i::")
    names = {k: json.loads(str(z[k])) for k in ("in_names", "out_names", "forcing_names")}
    pr = json.loads(str(z["prescriber"]))

    def ckpt(kind, weights, prefix, has_ema, sample_with_ema, interpolator_use_ema=False):
        """A Lightning-shaped checkpoint.  `has_ema`: trained with use_ema (model_ema.* shadows present);
        `sample_with_ema`: which copy holds the fixture's weights - the other copy is noise and must not be used."""
        hp = json.loads(json.dumps(lay[kind]["hyper_parameters"]))
        hp["use_ema"] = has_ema
        if kind == "forecaster":     # the data module of the stepper fixture (HGTsfc input-only, prescriber)
            hp["datamodule_config"].update(in_names=names["in_names"], out_names=names["out_names"],
                                           forcing_names=names["forcing_names"],
                                           prescriber=dict(_target_="x.Prescriber", **pr))
            hp["diffusion_config"]["enable_interpolator_dropout"] = False
            hp["diffusion_config"]["interpolator_use_ema"] = interpolator_use_ema
        else:
            hp["model_config"].update(dropout_mlp=0.0, drop_path_rate=0.0)
        keys = lay[kind]["state_dict_keys"]
        sd = {}
        for k in keys:
            w = weights.get(k[len(prefix):])
            if w is None:    # fc2 is index 2 (no dropout layer) in the stepper fixture, index 3 in the layout fixture
                w = weights[k[len(prefix):].replace("mlp.fwd.3", "mlp.fwd.2")]
                k = k.replace("mlp.fwd.3", "mlp.fwd.2")
            sd[k] = w.clone()
        if has_ema:
            handle = "model." if kind == "forecaster" else ""
            for k in list(sd):
                shadow = "model_ema." + (handle + k[len(prefix):]).replace(".", "")
                if sample_with_ema:
                    sd[shadow] = sd[k].clone()
                    sd[k] = torch.randn_like(sd[k])          # the raw weights must not be used
                else:
                    sd[shadow] = torch.randn_like(sd[k])     # the shadows must not be used
            sd["model_ema.decay"] = torch.tensor(0.9999)
        return {"hyper_parameters": hp, "state_dict": sd}

    means = {k[6:]: float(z[k]) for k in z.files if k.startswith("mean::")}
    stds = {k[5:]: float(z[k]) for k in z.files if k.startswith("std::")}
    data = {k[6:]: torch.from_numpy(z[k]).cuda() for k in z.files if k.startswith("data::")}
    n_steps = int(z["n_steps"])
    # (forecaster use_ema, interpolator trained with EMA, diffusion_config.interpolator_use_ema): the reference samples
    # the interpolator from its EMA shadows only under the LAST flag (dyffusion.py:236-237); the middle case is the
    # shipped configuration (fv3gfs.yaml use_ema: True for both, dyffusion.yaml interpolator_use_ema: False)
    for f_ema, i_has_ema, i_use_ema in ((False, False, False), (True, True, False), (True, True, True)):
        stepper = sdy_amd.checkpoint.stepper_from_state(
            ckpt("forecaster", fsd, "model.model.", f_ema, f_ema, interpolator_use_ema=i_use_ema),
            ckpt("interpolator", isd, "model.", i_has_ema, i_use_ema), means, stds, (32, 64))
        assert stepper.prescriber is not None and stepper.prescriber.prescribed_name == pr["prescribed_name"]
        out = stepper.run_on_batch(data, None, n_forward_steps=n_steps)
        for n in names["out_names"]:
            e = rel_l2(out.gen_data[n], torch.from_numpy(z["gen::" + n]))
            assert e < TOL_TIGHT, f"ema={(f_ema, i_has_ema, i_use_ema)} {n}: {e:.3e}"


def test_ensemble_metrics_vs_reference():
    """sdy_amd.metrics.ensemble_metrics (one HIP pass over the ensemble) vs the reference's core/metrics.py outputs."""
    import sdy_amd

    z = gu.load("fx_metrics")
    w = sdy_amd.metrics.spherical_area_weights(torch.from_numpy(z["lats"]), z["truth"].shape[-1])
    assert torch.allclose(w, torch.from_numpy(z["weights"]), rtol=1e-6, atol=0)
    got = sdy_amd.metrics.ensemble_metrics(torch.from_numpy(z["truth"]).cuda(), torch.from_numpy(z["pred"]).cuda(), w)
    for k in ("rmse", "spread", "spread_skill_ratio", "crps", "bias"):
        want = torch.from_numpy(z[k]).double()
        assert got[k].shape == want.shape
        assert torch.allclose(got[k].cpu(), want, rtol=2e-5, atol=2e-6), (k, got[k].cpu(), want)
    # 25 members on the model grid against the oracle
    from oracle.metrics import ensemble_metrics as oracle_metrics
    g = torch.Generator(device="cpu").manual_seed(3)
    truth = torch.randn(2, 180, 360, generator=g)
    pred = truth[None] + 0.5 * torch.randn(25, 2, 180, 360, generator=g)
    w = sdy_amd.metrics.spherical_area_weights(torch.linspace(-89.5, 89.5, 180), 360)
    got = sdy_amd.metrics.ensemble_metrics(truth.cuda(), pred.cuda(), w)
    ref = oracle_metrics(truth, pred, w)
    for k in ref:
        assert torch.allclose(got[k].cpu(), ref[k], rtol=1e-5, atol=1e-7), k
    with pytest.raises(RuntimeError):
        sdy_amd.metrics.ensemble_metrics(truth, pred, w)     # CPU tensors: no fallback


def test_time_mean_aggregator_vs_reference():
    """sdy_amd.metrics.TimeMeanAggregator (device-resident maps, one HIP launch per variable and window, strided member
    views) vs the reference's own TimeMeanAggregator fed the same two windows (fx_time_mean.npz)."""
    import sdy_amd

    z = gu.load("fx_time_mean")
    names = json.loads(str(z["names"]))
    W = z["ens::gen_map::a"].shape[-1]
    w = sdy_amd.metrics.spherical_area_weights(torch.from_numpy(z["lats"]), W)
    for key, ens in (("ens", True), ("det", False)):
        agg = sdy_amd.metrics.TimeMeanAggregator(w, is_ensemble=ens)
        for i in range(2):
            tgt = {n: torch.from_numpy(z[f"{key}::tgt{i}::{n}"]).cuda() for n in names}
            gen = {n: torch.from_numpy(z[f"{key}::gen{i}::{n}"]).cuda() for n in names}
            if ens:   # the window driver hands over (members, samples, ...) as a TRANSPOSED view of its IC-major batch
                gen = {n: v.transpose(0, 1).contiguous().transpose(0, 1) for n, v in gen.items()}
                assert not gen[names[0]].is_contiguous()
            agg.record_batch(0.0, tgt, gen, tgt, gen, i_time_start=int(z[f"{key}::i_time_start{i}"]))
        maps = agg.time_mean_maps()
        logs = agg.get_logs("inference")
        for n in names:
            assert rel_l2(maps["gen"][n], torch.from_numpy(z[f"{key}::gen_map::{n}"])) < 1e-6
            assert rel_l2(maps["target"][n], torch.from_numpy(z[f"{key}::target_map::{n}"])) < 1e-6
            assert abs(logs[f"inference/rmse/{n}"] - float(z[f"{key}::rmse::{n}"])) < 1e-5
            assert abs(logs[f"inference/bias/{n}"] - float(z[f"{key}::bias::{n}"])) < 1e-5
        want = sum(float(z[f"{key}::rmse::{n}"]) for n in names) / len(names)
        assert abs(logs["inference/rmse/channel_mean"] - want) < 1e-5
    with pytest.raises(ValueError):
        sdy_amd.metrics.TimeMeanAggregator(w).get_logs("x")
    with pytest.raises(RuntimeError):
        sdy_amd.metrics.TimeMeanAggregator(w).record_batch(0.0, {"a": torch.zeros(1, 2, 16, 32)}, {"a": torch.zeros(1, 2, 16, 32)},
                                                        {}, {})



def test_mean_aggregator_series_vs_reference():
    """sdy_amd.metrics.MeanAggregator (one `sdy_ensemble_series` launch per variable and window on the member-stacked VIEW,
    device accumulators indexed by i_time_start) vs the reference's own MeanAggregator fed the same three windows
    (fx_mean_series.npz; reduced.py:144-266)."""
    import sdy_amd

    z = gu.load("fx_mean_series")
    names = json.loads(str(z["names"]))
    W = z["ens::tgt0::a"].shape[-1]
    w = sdy_amd.metrics.spherical_area_weights(torch.from_numpy(z["lats"]), W)
    n_t = int(z["n_timesteps"])
    for key, ens in (("ens", True), ("det", False)):
        agg = sdy_amd.metrics.MeanAggregator(w, target="denorm", n_timesteps=n_t, is_ensemble=ens)
        for i in range(3):
            tgt = {n: torch.from_numpy(z[f"{key}::tgt{i}::{n}"]).cuda() for n in names}
            gen = {n: torch.from_numpy(z[f"{key}::gen{i}::{n}"]).cuda() for n in names}
            if ens:   # as the window driver presents it: a transposed view of the IC-major batch
                gen = {n: v.transpose(0, 1).contiguous().transpose(0, 1) for n, v in gen.items()}
                assert not gen[names[0]].is_contiguous()
            agg.record_batch(0.0, tgt, gen, tgt, gen, i_time_start=int(z[f"{key}::i_time_start{i}"]))
        series = agg.get_series()
        metrics = json.loads(str(z[f"{key}::metrics"]))
        assert sorted(agg.metric_names) == metrics
        for m in metrics:
            for n in names:
                want = torch.from_numpy(z[f"{key}::series::{m}/{n}"])
                got = series[f"{m}/{n}"].cpu()
                assert got.shape == want.shape == (n_t,)
                assert torch.allclose(got, want, rtol=5e-5, atol=5e-6), (key, m, n, got, want)
        logs = agg.get_logs("inference")
        assert set(logs["inference/series"]) == set(series)
        # the reference's composite (aggregator/inference/main.py): the same windows through ONE record_batch; its per-step
        # logs carry the series above, its last step the time-mean scalars, mean_step_20-style one-step means equal the series
        comp = sdy_amd.metrics.InferenceAggregator(w, n_timesteps=n_t, n_ensemble_members=3 if ens else 1)
        comp._aggregators["mean_step_2"] = sdy_amd.metrics.OneStepMeanAggregator(w, target_time=2, is_ensemble=ens)
        tm = sdy_amd.metrics.TimeMeanAggregator(w, is_ensemble=ens)
        for i in range(3):
            tgt = {n: torch.from_numpy(z[f"{key}::tgt{i}::{n}"]).cuda() for n in names}
            gen = {n: torch.from_numpy(z[f"{key}::gen{i}::{n}"]).cuda() for n in names}
            t0 = int(z[f"{key}::i_time_start{i}"])
            comp.record_batch(0.25 * (i + 1), tgt, gen, {n: 2 * v for n, v in tgt.items()}, {n: 2 * v for n, v in gen.items()},
                              i_time_start=t0)
            tm.record_batch(0.0, tgt, gen, tgt, gen, i_time_start=t0)
        steps = comp.get_inference_logs("inference")
        assert len(steps) == n_t and all(st["inference/mean/forecast_step"] == i for i, st in enumerate(steps))
        for m in metrics:
            for n in names:
                got = torch.tensor([st[f"inference/mean/{m}/{n}"] for st in steps], dtype=torch.float64)
                assert torch.allclose(got, series[f"{m}/{n}"].cpu(), rtol=1e-12, atol=0)
        # normalised copies were 2 x the fields: RMSE, bias, means and CRPS double, the spread-skill ratio does not move
        n0 = names[0]
        for st in steps:
            assert st[f"inference/mean_norm/weighted_rmse/{n0}"] == pytest.approx(2 * st[f"inference/mean/weighted_rmse/{n0}"], rel=1e-5)
        for k, v in tm.get_logs("time_mean").items():
            assert steps[-1][f"inference/{k}"] == pytest.approx(v, rel=1e-12)
            assert f"inference/{k}" not in steps[0]
        # the window holding forecast step 2 is the only one the one-step aggregator uses: its means are the series' entry
        # (the fixture's windows do not overlap), its loss the mean of ALL windows' losses over the recorded ones
        one = steps[-1]
        for m in ("weighted_rmse", "weighted_bias", "weighted_mean_gen") + (("weighted_crps", "weighted_ssr") if ens else ()):
            for n in names:
                assert one[f"inference/mean_step_2/{m}/{n}"] == pytest.approx(float(series[f"{m}/{n}"][2]), rel=2e-5, abs=1e-7)
        assert one["inference/mean_step_2/loss"] == pytest.approx(0.25 + 0.5 + 0.75)
        maps = comp.get_time_mean_maps()
        assert set(maps["gen"]) == set(names) and maps["gen"][n0].shape == w.shape
    with pytest.raises(ValueError):
        sdy_amd.metrics.MeanAggregator(w, n_timesteps=4).get_series()
    with pytest.raises(ValueError):      # a ragged share hands flat rows: ensemble metrics need whole initial conditions
        a = sdy_amd.metrics.MeanAggregator(w, n_timesteps=4, is_ensemble=True)
        x = torch.zeros(3, 2, 16, 32).cuda()
        a.record_batch(0.0, {"a": x[:1]}, {"a": x}, {}, {})
