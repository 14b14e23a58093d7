"""Host-side logic of the drop-in surface that needs no GPU: schedule maths, the stateful prediction cache, state_dict
contract, sharding."""
import numpy as np
import pytest
import torch

import golden_utils as gu


@pytest.fixture(scope="module")
def sdy():
    import sdy_amd

    return sdy_amd


class _FakeIpol:
    window, true_horizon = 1, 6


def _sampler(sdy, **kw):
    return sdy.DYffusion(model=None, interpolator=_FakeIpol(), timesteps=6, **kw)


def test_schedule_maths_matches_reference_semantics(sdy):
    d = _sampler(sdy)
    assert d.num_timesteps == 6 and d.sampling_schedule == [0, 1, 2, 3, 4, 5]
    assert [d.diffusion_step_to_interpolation_step(i) for i in range(6)] == [0, 1, 2, 3, 4, 5]
    assert d.valid_time_range_for_backbone_model == [0, 1, 2, 3, 4, 5]
    # k = 2 artificial steps before t1 (dyffusion.py:155-169): d1 -> 1/3, d2 -> 2/3, d3 -> 1, ...
    d2 = _sampler(sdy, additional_interpolation_steps=2)
    assert d2.num_timesteps == 8
    got = [d2.diffusion_step_to_interpolation_step(i) for i in range(8)]
    assert got[0] == 0 and abs(got[1] - 1 / 3) < 1e-12 and abs(got[2] - 2 / 3) < 1e-12 and got[3:] == [1, 2, 3, 4, 5]
    assert d2.dynamical_steps == {3: 1, 4: 2, 5: 3, 6: 4, 7: 5}
    d2.sampling_schedule = "only_dynamics"
    assert d2.sampling_schedule == [0, 3, 4, 5, 6, 7]
    d2.sampling_schedule = [3, 4.0, 5, 6, 7]          # explicit lists: 0 is prepended, integral floats become ints
    assert d2.sampling_schedule == [0, 3, 4, 5, 6, 7]
    # the reference's schedule NAMES (dyffusion.py:369-455): known answers produced by the reference's own setter
    # (BaseDYffusion with timesteps = 6 and k = 2 / k = 5 artificial steps), recorded in the build container
    known = {
        2: {"only_dynamics": [0, 3, 4, 5, 6, 7], "only_dynamics_plus1": [0, 1.5, 3, 4, 5, 6, 7],
            "only_dynamics_plus3": [0, 0.75, 1.5, 2.25, 3, 4, 5, 6, 7], "only_dynamics_plus_discrete2": [0, 1, 2, 3, 4, 5, 6, 7],
            "every2": [0, 1, 3, 4, 5, 6, 7], "every3rd": [0, 1, 3, 4, 5, 6, 7], "first1": [0, 1, 3, 4, 5, 6, 7],
            "first2": [0, 1, 2, 3, 4, 5, 6, 7], "first0.5": [0, 1, 3, 4, 5, 6, 7], "first0.3": [0, 1, 3, 4, 5, 6, 7]},
        5: {"only_dynamics": [0, 6, 7, 8, 9, 10], "only_dynamics_plus1": [0, 3, 6, 7, 8, 9, 10],
            "only_dynamics_plus3": [0, 1.5, 3, 4.5, 6, 7, 8, 9, 10], "only_dynamics_plus_discrete2": [0, 2, 4, 6, 7, 8, 9, 10],
            "every2": [0, 1, 3, 5, 6, 7, 8, 9, 10], "every3rd": [0, 1, 4, 6, 7, 8, 9, 10], "first1": [0, 1, 6, 7, 8, 9, 10],
            "first2": [0, 1, 2, 6, 7, 8, 9, 10], "first0.5": [0, 1, 2, 3, 6, 7, 8, 9, 10], "first0.3": [0, 1, 2, 6, 7, 8, 9, 10],
            "first9": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]},
    }
    for k, table in known.items():
        dk = _sampler(sdy, additional_interpolation_steps=k)
        for name, want in table.items():
            dk.sampling_schedule = name
            assert dk.sampling_schedule == want, (k, name, dk.sampling_schedule)
    with pytest.raises(AssertionError):
        d2.sampling_schedule = "first9"               # more steps than the schedule has (the reference asserts too)
    with pytest.raises(ValueError):
        d2.sampling_schedule = "bogus"
    with pytest.raises(AssertionError):
        d2.sampling_schedule = [0, 2, 2]
    with pytest.raises(AssertionError):
        d2.sampling_schedule = [0, 9]
    # capabilities a checkpoint's diffusion_config may carry: constructing must not refuse them
    assert _sampler(sdy, refine_intermediate_predictions=True).hparams.refine_intermediate_predictions
    assert _sampler(sdy, log_every_t="auto").hparams.log_every_t == "auto"
    with pytest.raises(AssertionError):
        d.diffusion_step_to_interpolation_step(6)
    with pytest.raises(ValueError):
        sdy.DYffusion(model=None, interpolator=_FakeIpol(), timesteps=5)     # horizon mismatch (dyffusion.py:634-640)
    with pytest.raises(NotImplementedError):
        _sampler(sdy, schedule="linear")


def test_prediction_cache_surface(sdy):
    """get_preds_at_t_for_batch: horizon 1 computes, later horizons pop the cache, the last one clears it
    (forecasting_multi_horizon.py:347-380)."""
    class FakeNet:
        def set_min_max_time(self, **kw): self.rng = kw
        def enable_inference_dropout(self): pass
        def disable_inference_dropout(self): pass

    net = FakeNet()
    ipol = sdy.InterpolationExperiment(FakeNet(), horizon=6)
    assert ipol.horizon_range == [1, 2, 3, 4, 5] and ipol.model.rng == {"min_time": 1, "max_time": 5}
    exp = sdy.MultiHorizonForecastingDYffusion(net, ipol, horizon=6)
    assert net.rng == {"min_time": 0, "max_time": 5} and exp.true_horizon == 6
    calls = []

    def fake_predict_forward(x, **kw):
        calls.append(sorted(kw))
        return {f"t{h}_preds": x + h for h in range(1, 7)}

    exp.model.predict_forward = fake_predict_forward
    x = torch.zeros(2, 3, 4, 8)
    for h in range(1, 7):
        out = exp.get_preds_at_t_for_batch({"dynamics": x, "static_condition": x}, horizon=h, split="predict",
                                           prepare_inputs=False, num_predictions=1)
        assert list(out) == [f"t{h}_preds_normed"] and float(out[f"t{h}_preds_normed"].mean()) == h
    assert calls == [["static_condition"]] and exp._current_preds is None
    with pytest.raises(AssertionError):
        exp.get_preds_at_t_for_batch({"dynamics": x}, horizon=0, split="predict", prepare_inputs=False)
    with pytest.raises(AssertionError):   # horizon 2 before horizon 1
        exp.get_preds_at_t_for_batch({"dynamics": x}, horizon=2, split="predict", prepare_inputs=False)


def test_state_dict_contract_matches_reference_fixture(sdy):
    """Names and shapes of the product network == the reference network's state_dict (fixture from the real class)."""
    z = gu.load("fx_sfno_tiny")
    cfg, n_in, n_cond = gu.cfg_from(z)
    ref_sd = gu.state_dict(z)
    net = sdy.SphericalFourierNeuralOperatorNet(
        num_input_channels=n_in, num_output_channels=cfg.out_chans, num_conditional_channels=n_cond,
        spatial_shape_in=(cfg.nlat, cfg.nlon), embed_dim=cfg.embed_dim, num_layers=cfg.num_layers,
        dropout_mlp=cfg.dropout_mlp, drop_path_rate=cfg.drop_path_rate, with_time_emb=True)
    mine = net.state_dict()
    assert set(mine) == set(ref_sd)
    for k in ref_sd:
        assert tuple(mine[k].shape) == tuple(ref_sd[k].shape), k
    net.load_state_dict(ref_sd, strict=True)
    with pytest.raises(RuntimeError):
        net.load_state_dict({**ref_sd, "bogus": torch.zeros(1)}, strict=True)
    # dropout = 0 moves fc2 to index 2 of the Sequential (layers.py:76-80)
    net0 = sdy.SphericalFourierNeuralOperatorNet(4, 4, spatial_shape_in=(32, 64), embed_dim=8, num_layers=1)
    assert "blocks.0.mlp.fwd.2.weight" in net0.state_dict() and "blocks.0.mlp.fwd.3.weight" not in net0.state_dict()
    with pytest.raises(NotImplementedError):
        sdy.SphericalFourierNeuralOperatorNet(4, 4, operator_type="diagonal")
    # keywords that would change the (stochastic) model are refused, the inert ones of the shipped sfno.yaml are accepted
    with pytest.raises(NotImplementedError):
        sdy.SphericalFourierNeuralOperatorNet(4, 4, pos_emb_dropout=0.1)
    with pytest.raises(NotImplementedError):
        sdy.SphericalFourierNeuralOperatorNet(4, 4, debug_mode=True)
    with pytest.raises(TypeError):
        sdy.SphericalFourierNeuralOperatorNet(4, 4, no_such_option=1)
    sdy.SphericalFourierNeuralOperatorNet(4, 4, spatial_shape_in=(32, 64), embed_dim=8, num_layers=1, dropout_filter=0.0,
                                          pos_emb_dropout=0.0, spectral_layers=3, sparsity_threshold=0.0, num_blocks=8,
                                          checkpointing=0, loss_function="mse", verbose=False, name="")


def test_partition(sdy):
    from sdy_amd import ensemble

    assert [c for _, c in ensemble.partition(25, 8)] == [4, 3, 3, 3, 3, 3, 3, 3]
    parts = ensemble.partition(100, 8)
    assert sum(c for _, c in parts) == 100 and parts[0] == (0, 13) and parts[-1] == (88, 12)
    units = [u for r in range(8) for u in ensemble.rank_units(4, 25, r, 8)]
    assert units == [(ic, m) for ic in range(4) for m in range(25)]
    assert [len(b) for b in ensemble.batches(ensemble.rank_units(4, 25, 0, 8), 5)] == [5, 5, 3]
    assert ensemble.partition(3, 8)[5] == (3, 0)


def test_stepper_host_validation(sdy):
    with pytest.raises(ValueError):
        sdy.Prescriber("a", "m", mask_value=0, interpolate=True)                  # prescriber.py:33-35
    with pytest.raises(ValueError):                                                # prescribed var must be in & out
        sdy.MultiStepStepper(None, ["a", "b", "f"], ["b"], ["f"], {}, {}, sdy.Prescriber("a", "m", 1))
    st = sdy.MultiStepStepper(None, ["HGTsfc", "a", "b", "f"], ["a", "b"], ["f"], {"a": 1.0}, {"a": 2.0})
    assert st.in_names == ["HGTsfc", "a", "b"] and st._entries == ["HGTsfc", "a", "b"]
    with pytest.raises(RuntimeError, match="GPU only"):
        class M:
            true_horizon = 6
            model = None
        sdy.MultiStepStepper(M(), ["a", "f"], ["a"], ["f"], {}, {}).run_on_batch({"a": torch.zeros(1, 2, 4, 8)}, None, 1)


def test_checkpoint_weight_selection_follows_lightning_layout():
    """checkpoint.select_weights against the key layout of the reference's own experiments (fx_ckpt_layout.json: state_dict
    keys and LitEma buffer names recorded from the reference): raw vs EMA selection, interpolator keys skipped."""
    import json
    import os

    import torch

    from sdy_amd.checkpoint import select_weights

    with open(os.path.join(os.path.dirname(__file__), "golden", "fx_ckpt_layout.json")) as f:
        lay = json.load(f)
    fk = lay["forecaster"]["state_dict_keys"]
    assert all(k.startswith("model.model.") for k in fk)
    sd = {k: torch.full((2,), float(i)) for i, k in enumerate(fk)}
    sd["model.interpolator.model.pos_embed"] = torch.zeros(2)          # stray interpolator key: ignored under model.model.
    for name, buf in lay["forecaster"]["ema_names"].items():             # name is relative to experiment.model (DYffusion)
        sd["model_ema." + buf] = sd["model." + name] + 1000.0
    sd["model_ema.decay"] = torch.tensor(0.9999)
    sd["model_ema.num_updates"] = torch.tensor(7)
    raw = select_weights(sd, "model.model.", "model.", use_ema=False)
    ema = select_weights(sd, "model.model.", "model.", use_ema=True)
    assert sorted(raw) == sorted(k[len("model.model."):] for k in fk) == sorted(ema)
    assert all(float(ema[k][0]) == float(raw[k][0]) + 1000.0 for k in raw)
    # interpolation experiment: the network is experiment.model, EMA names relative to it
    ik = lay["interpolator"]["state_dict_keys"]
    isd = {k: torch.full((1,), float(i)) for i, k in enumerate(ik)}
    for k in ik:
        isd["model_ema." + k[len("model."):].replace(".", "")] = isd[k] - 500.0
    iema = select_weights(isd, "model.", "", use_ema=True)
    assert all(float(iema[k[len("model."):]][0]) == float(isd[k][0]) - 500.0 for k in ik)
    import pytest
    with pytest.raises(KeyError):
        select_weights({"foo.bar": torch.zeros(1)}, "model.model.", "model.", False)
    with pytest.raises(KeyError):
        select_weights({k: v for k, v in sd.items() if "model_ema" not in k}, "model.model.", "model.", True)


def test_interpolator_ema_follows_forecaster_diffusion_config():
    """checkpoint.module_weights: the interpolator's EMA shadows are used iff the FORECASTER's
    diffusion_config.interpolator_use_ema is set (reference dyffusion.py:236-237), never because the interpolator's own
    checkpoint says use_ema (shipped: fv3gfs.yaml use_ema True everywhere, dyffusion.yaml interpolator_use_ema False)."""
    import pytest
    import torch

    from sdy_amd.checkpoint import module_weights

    def ck(prefix, handle, use_ema, diffusion=None):
        sd = {prefix + "encoder.0.weight": torch.full((1,), 1.0), prefix + "decoder.2.weight": torch.full((1,), 2.0)}
        if use_ema:
            for k in list(sd):
                sd["model_ema." + (handle + k[len(prefix):]).replace(".", "")] = sd[k] + 100.0
        hp = {"use_ema": use_ema}
        if diffusion is not None:
            hp["diffusion_config"] = diffusion
        return {"hyper_parameters": hp, "state_dict": sd}

    # shipped: both trained with EMA, interpolator_use_ema False -> forecaster EMA, interpolator RAW
    fw, iw = module_weights(ck("model.model.", "model.", True, {"interpolator_use_ema": False}), ck("model.", "", True))
    assert float(fw["encoder.0.weight"]) == 101.0 and float(iw["encoder.0.weight"]) == 1.0
    # key absent from the diffusion config: the reference's default is False (dyffusion.py:52)
    fw, iw = module_weights(ck("model.model.", "model.", True, {}), ck("model.", "", True))
    assert float(iw["decoder.2.weight"]) == 2.0
    # interpolator_use_ema True -> shadows
    fw, iw = module_weights(ck("model.model.", "model.", False, {"interpolator_use_ema": True}), ck("model.", "", True))
    assert float(fw["encoder.0.weight"]) == 1.0 and float(iw["encoder.0.weight"]) == 101.0
    # ... and an interpolator checkpoint without shadows cannot serve that request (the reference has no model_ema then)
    with pytest.raises(KeyError):
        module_weights(ck("model.model.", "model.", False, {"interpolator_use_ema": True}), ck("model.", "", False))
    # explicit overrides win
    fw, iw = module_weights(ck("model.model.", "model.", True, {}), ck("model.", "", True), use_ema=False,
                            interpolator_use_ema=True)
    assert float(fw["encoder.0.weight"]) == 1.0 and float(iw["encoder.0.weight"]) == 101.0



def test_persisted_sht_buffers_are_checked(sdy):
    """checkpoints written with an old torch-harmonics carry `*.weights` / `*.pct`; those are the
    tables the network was trained with (reference sfnonet.py:551-554).  Matching ones load, different ones raise."""
    from oracle.sht import sht_tables
    from sdy_amd._lib import SdyError

    H, W, E, Lr = 16, 32, 8, 2
    net = sdy.SphericalFourierNeuralOperatorNet(4, 4, spatial_shape_in=(H, W), embed_dim=E, num_layers=Lr)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    tabs = {g: sht_tables(H, W, H, W // 2 + 1, g) for g in ("equiangular", "legendre-gauss")}
    f32 = lambda a: torch.from_numpy(a).float()  # noqa: E731
    eq, lg = tabs["equiangular"], tabs["legendre-gauss"]
    persisted = {
        "trans_down.weights": f32(eq[1]), "itrans_up.pct": f32(eq[0]), "trans.weights": f32(lg[1]), "itrans.pct": f32(lg[0]),
        "blocks.0.filter.filter.forward_transform.weights": f32(eq[1]),      # first block reads the data grid
        "blocks.0.filter.filter.inverse_transform.pct": f32(lg[0]),
        "blocks.1.filter.filter.forward_transform.weights": f32(lg[1]),
        "blocks.1.filter.filter.inverse_transform.pct": f32(eq[0]),          # last block writes the data grid
    }
    missing, unexpected = net.load_state_dict({**sd, **persisted}, strict=True)
    assert not missing and not unexpected
    # a table from another convention (no Condon-Shortley phase) is refused, naming the key
    bad = dict(persisted)
    nocs = eq[0].copy()
    nocs[1::2] *= -1
    bad["itrans_up.pct"] = f32(nocs)
    with pytest.raises(SdyError, match="itrans_up.pct"):
        net.load_state_dict({**sd, **bad}, strict=True)
    # the Gauss table where the equiangular one belongs (wrong grid for the first block)
    bad = dict(persisted)
    bad["blocks.0.filter.filter.forward_transform.weights"] = f32(lg[1])
    with pytest.raises(SdyError, match="blocks.0.filter"):
        net.load_state_dict({**sd, **bad}, strict=False)
    # another layout
    bad = dict(persisted)
    bad["trans.weights"] = f32(lg[1]).permute(1, 0, 2).contiguous()
    with pytest.raises(SdyError, match="shape"):
        net.load_state_dict({**sd, **bad}, strict=True)
    # keys that merely end in .weights elsewhere are still unexpected
    with pytest.raises(RuntimeError):
        net.load_state_dict({**sd, "something.else.weights": torch.zeros(1)}, strict=True)


def test_window_stitcher_carries_state_per_trajectory(sdy):
    """loop.WindowStitcher on the IC-major flat layout: generated variables are carried per trajectory, everything else per
    initial condition (reference loop.py:85-117), the writer sees the running time index."""
    from sdy_amd.loop import WindowStitcher

    calls = []

    class W:
        def append_batch(self, **kw):
            calls.append((kw["start_timestep"], kw["start_sample"]))

    n_sample, members, T1, H, Wd = 2, 3, 4, 2, 2
    st = WindowStitcher(n_forward_steps=6, writer=W(), is_ensemble=True)
    target = {"a": torch.arange(n_sample * T1 * H * Wd, dtype=torch.float32).view(n_sample, T1, H, Wd),
              "forc": torch.ones(n_sample, T1, H, Wd)}
    rows = n_sample * members
    last = {"a": torch.arange(rows, dtype=torch.float32).view(rows, 1, 1).expand(rows, H, Wd) + 100.0}
    gen = {"a": torch.zeros(members, n_sample, T1, H, Wd)}
    st.append(target, gen, None, last_state=last)
    assert calls == [(0, 0)] and st.i_time == T1
    ic_rows = torch.tensor([0, 0, 0, 1, 1, 1])
    batch = {"a": torch.zeros(rows, T1, H, Wd), "forc": torch.zeros(rows, T1, H, Wd)}
    st.apply_initial_condition(batch, ic_rows)
    assert torch.equal(batch["a"][:, 0, 0, 0], torch.arange(rows, dtype=torch.float32) + 100.0)   # per trajectory
    assert torch.equal(batch["forc"][:, 0], target["forc"][:, -1].index_select(0, ic_rows))         # per IC
    st.append({k: v[:, 1:] for k, v in target.items()}, gen, None, last_state=last)
    assert calls[-1] == (T1, 0) and st.i_time == 2 * T1 - 1
    with pytest.raises(ValueError):
        st.apply_initial_condition(batch, ic_rows)


def test_synthetic_weights_equal_the_oracle_generator(sdy):
    """`sdy_amd.synthetic.trained_like_state_dict` (what bench.py / tools build their networks from, product code) and
    `oracle.sfno.make_state_dict` (what the parity tests load into both sides) are the same generator, value for value: a
    benchmark network and a parity-test network of one seed are the same network."""
    import torch

    from oracle.sfno import SFNOConfig, make_state_dict

    for drop, cond, big_skip, temb in ((0.1, 2, True, True), (0.0, 0, False, False)):
        cfg = SFNOConfig(in_chans=6 + cond, out_chans=4, nlat=16, nlon=32, embed_dim=8, num_layers=2, dropout_mlp=drop,
                         drop_path_rate=drop, with_time_emb=temb, big_skip=big_skip, pos_embed=big_skip)
        net = sdy.SphericalFourierNeuralOperatorNet(6, 4, num_conditional_channels=cond, spatial_shape_in=(16, 32), embed_dim=8,
                                                    num_layers=2, dropout_mlp=drop, drop_path_rate=drop, with_time_emb=temb,
                                                    big_skip=big_skip, pos_embed=big_skip)
        a, b = sdy.synthetic.trained_like_state_dict(net, seed=77), make_state_dict(cfg, seed=77)
        assert list(a) == list(b)
        assert all(torch.equal(a[k], b[k]) for k in a)
        net.load_state_dict(a, strict=True)


# The variable lists of the shipped data module (`src/configs/datamodule/fv3gfs_prescriptive_only.yaml:22-60`).
_FV3GFS_STATE = (["PRESsfc", "surface_temperature"] + [f"{v}_{k}" for v in
                 ("air_temperature", "specific_total_water", "eastward_wind", "northward_wind") for k in range(8)])
_FV3GFS_FORCING = ["DSWRFtoa", "HGTsfc"]


def test_packaged_statistics_hold_the_reference_scalars(sdy, tmp_path):
    """`data_statistics/{centering,scaling}.nc` of the reference, shipped HDF5-free (`tools/convert_statistics.py`): known
    answers read off the reference's files (float32, printed to round-trip), the shipped variable lists are covered, the
    reference's search order is kept (`stepper_multistep.py:112-127`), a NaN scaling is refused."""
    import json
    import math
    import os
    nz = sdy.normalizer
    mean_p, std_p = nz.find_statistics()
    assert mean_p.parent == nz.PACKAGED_STATISTICS and mean_p.name == "centering.json" and std_p.name == "scaling.json"
    means, stds = nz.load_dict(mean_p), nz.load_dict(std_p)
    assert len(means) == len(stds) == 55 and set(means) == set(stds)
    known_mean = {"PRESsfc": 96794.3828125, "surface_temperature": 277.7657775878906, "air_temperature_7": 276.7269287109375,
                  "specific_total_water_0": 2.01499710783537e-06, "DSWRFtoa": 298.72540283203125, "HGTsfc": 377.55096435546875}
    known_std = {"DLWRFsfc": 61.89848709106445, "HGTsfc": 836.8834228515625, "PRESsfc": 791.6809692382812,
                 "specific_total_water_0": 1.6999416985186144e-08, "northward_wind_3": 17.700998306274414}
    for k, v in known_mean.items():
        assert means[k] == v and np.float32(v) == v        # bit-exact float32
    for k, v in known_std.items():
        assert stds[k] == v and np.float32(v) == v
    norm = nz.get_normalizer(mean_p, std_p, _FV3GFS_STATE + _FV3GFS_FORCING)
    assert all(math.isfinite(norm.means[n]) and norm.stds[n] > 0 for n in _FV3GFS_STATE + _FV3GFS_FORCING)
    assert nz.StandardNormalizer.from_state(norm.get_state()).stds == norm.stds
    x = {"PRESsfc": torch.full((2, 2), 96794.3828125 + 791.6809692382812), "other": torch.ones(1)}
    y = norm.normalize(x)
    assert torch.allclose(y["PRESsfc"], torch.ones(2, 2)) and y["other"] is x["other"]           # normalizer.py:98-102
    assert torch.allclose(norm.denormalize(y)["PRESsfc"], x["PRESsfc"])
    assert math.isnan(stds["soil_moisture"])
    with pytest.raises(ValueError, match="soil_moisture"):
        nz.get_normalizer(mean_p, std_p, ["soil_moisture"])
    with pytest.raises(KeyError):
        nz.load_dict(mean_p, ["no_such_variable"])
    # a directory given by the caller wins over the packaged copy
    for stem, val in (("centering", 1.5), ("scaling", 2.5)):
        with open(tmp_path / (stem + ".json"), "w") as fh:
            json.dump({"variables": {"a": val}}, fh)
    m2, s2 = nz.find_statistics(data_dir_stats=tmp_path)
    assert m2.parent == tmp_path and nz.load_dict(m2) == {"a": 1.5} and nz.load_dict(s2) == {"a": 2.5}
    # the stepper's constructor path: forcings = input-only names, scalars by name
    st = sdy.MultiStepStepper.from_statistics(None, _FV3GFS_STATE + _FV3GFS_FORCING, _FV3GFS_STATE)
    assert sorted(st.forcing_names) == sorted(_FV3GFS_FORCING) and st.in_names == _FV3GFS_STATE
    assert st.means["HGTsfc"] == known_mean["HGTsfc"] and st.stds["HGTsfc"] == known_std["HGTsfc"]
    # where the reference's files and a libhdf5 are at hand (the build container), the converter reproduces the shipped JSON
    src = "/root/reference/data_statistics"
    if os.path.exists(src):
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "convert_statistics", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "convert_statistics.py"))
        conv = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(conv)
        try:
            fresh = conv.read_scalars(os.path.join(src, "scaling.nc"))
        except SystemExit:
            return
        assert {k: v for k, v in fresh.items() if v == v} == {k: v for k, v in stds.items() if v == v}


def test_inference_log_plumbing_follows_the_reference(sdy):
    """`data_to_table` (`aggregator/inference/reduced.py:282-293`) and `to_inference_logs` (`main.py:189-211`): the series of
    an aggregator become one table with a `forecast_step` column and sorted keys; tables become one dict per row with the
    table's own name (the key's last component) dropped; scalars land in the last row's dict."""
    m = sdy.metrics
    t = m.data_to_table({"weighted_rmse/b": [1.0, 2.0, 3.0], "weighted_bias/a": [0.1, 0.2, 0.3]})
    assert t.columns == ["forecast_step", "weighted_bias/a", "weighted_rmse/b"]
    assert t.data == [[0, 0.1, 1.0], [1, 0.2, 2.0], [2, 0.3, 3.0]]
    with pytest.raises(ValueError):
        t.add_data(1, 2)
    logs = m.to_inference_logs({"inference/mean/series": t, "inference/time_mean/rmse/a": 0.5})
    assert logs == [
        {"inference/mean/forecast_step": 0, "inference/mean/weighted_bias/a": 0.1, "inference/mean/weighted_rmse/b": 1.0},
        {"inference/mean/forecast_step": 1, "inference/mean/weighted_bias/a": 0.2, "inference/mean/weighted_rmse/b": 2.0},
        {"inference/mean/forecast_step": 2, "inference/mean/weighted_bias/a": 0.3, "inference/mean/weighted_rmse/b": 3.0,
         "inference/time_mean/rmse/a": 0.5}]
    # the composite refuses the image products it does not make, and needs the series length
    w = torch.ones(4, 8)
    with pytest.raises(NotImplementedError):
        m.InferenceAggregator(w, n_timesteps=5, log_video=True)
    with pytest.raises(NotImplementedError):
        m.InferenceAggregator(w, n_timesteps=5, log_zonal_mean_images=True)
    with pytest.raises(ValueError):
        m.InferenceAggregator(w)
    agg = m.InferenceAggregator(w, sigma_coordinates=object(), n_timesteps=25, n_ensemble_members=3, record_step_20=True)
    assert list(agg._aggregators) == ["mean", "mean_norm", "time_mean", "mean_step_20"]
    with pytest.raises(ValueError, match="target_data"):
        agg.record_batch(0.0, {}, {"a": w}, {}, {"a": w})


def test_bench_reads_the_counter_traffic_of_the_newest_profile():
    """`roofline.traffic` of the bench line is not a pasted constant: bench.measured_traffic() parses the newest
    profiles/*/pmc_summary.txt for the dominant kernel's row (FETCH_SIZE x 2 + WRITE_SIZE per 25-row launch) and cites the file."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    traffic, src, why = bench.measured_traffic()
    prof = os.path.join(root, "profiles")
    key = lambda d: bench.profile_round_key(os.path.join(prof, d, "x"))  # noqa: E731
    assert key("r10a") > key("r5d") > key("r5c") > key("r4f") and key("bench_r1_first.json") == (-1, "")   # rounds, not strings
    newest_pmc = sorted((d for d in os.listdir(prof) if os.path.exists(os.path.join(prof, d, "pmc_summary.txt"))), key=key)[-1]
    newest_trace = sorted((d for d in os.listdir(prof) if os.path.exists(os.path.join(prof, d, "kernel_stats.csv"))), key=key)[-1]
    if key(newest_trace) > key(newest_pmc):       # a round that re-profiled without the counter passes: null, with the reason
        assert traffic is None and src is None and "stale" in why
        return
    assert why is None and src == os.path.join("profiles", newest_pmc, "pmc_summary.txt")
    assert os.path.exists(os.path.join(root, src))
    algorithmic = 3 * 4.0 * 256 * 180 * 360 * 25            # x, residual, output: fp32 (25, 256, 180, 360) tensors
    assert algorithmic <= traffic <= 1.15 * algorithmic, (traffic, algorithmic)
    # the per-launch work of a stage scales with the rows its launches covered (drop-path skip)
    w25, w20 = bench.stage_work(25)["mlp fused (dropout)"], bench.stage_work(20)["mlp fused (dropout)"]
    assert abs(w20[0] / w25[0] - 0.8) < 1e-12 and abs(w20[1] / w25[1] - 0.8) < 1e-12
