"""Pins the CPU oracle against golden vectors produced by the REFERENCE'S OWN classes (tools/gen_golden.py)."""
import json
import os

import pytest
import torch

import golden_utils as gu
from conftest import rel_l2
from oracle.dyffusion import OracleDYffusion
from oracle.sfno import OracleSFNO, SFNOConfig

TOL = 2e-6   # oracle and reference run the same torch CPU ops; differences are summation-order level


def _t(z, k):
    return torch.from_numpy(z[k]) if k in z.files else None


@pytest.mark.parametrize("name", ["fx_block_c1", "fx_sfno_tiny", "fx_sfno_tiny_lg"])
def test_network_matches_reference(name):
    z = gu.load(name)
    cfg, n_in, n_cond = gu.cfg_from(z)
    net = OracleSFNO(cfg, gu.state_dict(z))
    y = net(_t(z, "x"), time=_t(z, "time"), condition=_t(z, "cond"))
    err = rel_l2(y, _t(z, "y"))
    assert err < TOL, f"{name}: rel L2 {err:.3e}"
    if "t_repr" in z.files:
        assert rel_l2(net.time_repr(_t(z, "time")), _t(z, "t_repr")) < TOL


def test_block_matches_reference():
    """FourierNeuralOperatorBlock.forward in isolation (sfnonet.py:289-337): input/output captured by a hook."""
    z = gu.load("fx_block_c1")
    cfg, _, _ = gu.cfg_from(z)
    net = OracleSFNO(cfg, gu.state_dict(z))
    t_repr = net.time_repr(_t(z, "time"))
    y = net.block(0, _t(z, "block0_in"), t_repr, None)
    err = rel_l2(y, _t(z, "block0_out"))
    assert err < TOL, f"block: rel L2 {err:.3e}"


def test_network_with_recorded_dropout_masks():
    z = gu.load("fx_sfno_tiny")
    cfg, _, _ = gu.cfg_from(z)
    net = OracleSFNO(cfg, gu.state_dict(z))
    fwd = gu.masks_per_forward(gu.recorded_masks(z), cfg)
    assert len(fwd) == 1
    # block 0 has no DropPath (rate 0 -> nn.Identity, sfnonet.py:252)
    assert ("drop_path", 0) not in fwd[0] and ("drop_path", 1) in fwd[0]
    y = net(_t(z, "x"), time=_t(z, "time"), condition=_t(z, "cond"), mask_fn=gu.mask_fn_from(fwd[0]))
    err = rel_l2(y, _t(z, "y_dropout"))
    assert err < TOL, f"dropout: rel L2 {err:.3e}"
    assert rel_l2(_t(z, "y_dropout"), _t(z, "y")) > 1e-2   # the masks matter


def test_full_size_network_matches_reference():
    """The oracle against the REFERENCE network's own output at production size: 180 x 360, E = 256, all 8 blocks,
    68 + 2 -> 34 channels (tests/golden/fx_sfno_full.npz; weights and inputs from seeds, checksummed)."""
    z = gu.load("fx_sfno_full")
    cfg, n_in, n_cond, sd, x, cond, t = gu.seeded_case(z)
    assert (cfg.nlat, cfg.nlon, cfg.embed_dim, cfg.num_layers) == (180, 360, 256, 8)
    y = OracleSFNO(cfg, sd)(x, time=t, condition=cond)
    err = rel_l2(y, _t(z, "y"))
    assert err < 5e-6, f"full size: rel L2 {err:.3e}"


def test_wide_network_with_recorded_dropout_masks():
    """E = 256 / hidden 512 on the small grid with the masks the reference's nn.Dropout / DropPath layers drew."""
    z = gu.load("fx_sfno_wide_masks")
    cfg, n_in, n_cond, sd, x, cond, t = gu.seeded_case(z)
    net = OracleSFNO(cfg, sd)
    assert rel_l2(net(x, time=t, condition=cond), _t(z, "y")) < 5e-6
    fwd = gu.masks_per_forward(gu.recorded_masks(z), cfg)
    assert len(fwd) == 1 and ("drop_path", 2) in fwd[0]
    y = net(x, time=t, condition=cond, mask_fn=gu.mask_fn_from(fwd[0]))
    err = rel_l2(y, _t(z, "y_dropout"))
    assert err < 5e-6, f"dropout: rel L2 {err:.3e}"


def _sampler(z, masks=None):
    fcfg = SFNOConfig(**json.loads(str(z["fcfg"])))
    icfg = SFNOConfig(**json.loads(str(z["icfg"])))
    fnet = OracleSFNO(fcfg, gu.state_dict(z, "f::"))
    inet = OracleSFNO(icfg, gu.state_dict(z, "i::"))
    per_fwd = gu.masks_per_forward(masks, icfg) if masks is not None else None
    n = {"i": 0}
    trace = []

    def f(x, time, condition=None, static_condition=None):
        trace.append(["F", float(time[0])])
        return fnet(x, time=time, condition=condition, static_condition=static_condition)

    extra = json.loads(str(z["diffusion_extra"])) if "diffusion_extra" in z.files else {}
    per_call = extra.get("enable_interpolator_dropout") == "except_dynamical_steps"
    smp = None

    def i(x, time, condition=None, static_condition=None):
        on = per_fwd is not None and smp.dropout_on        # a mask set exists for exactly the calls whose dropout was on
        trace.append(["I", float(time[0])] + ([bool(on)] if per_call else []))
        mf = gu.mask_fn_from(per_fwd[n["i"]]) if on else None
        n["i"] += int(on)
        return inet(x, time=time, condition=condition, static_condition=static_condition, mask_fn=mf)

    smp = OracleDYffusion(f, i, timesteps=6, hack_for_imprecise_interpolation=bool(int(z["hack"])), **extra)
    return smp, trace


@pytest.mark.parametrize("name", ["fx_sample_tiny", "fx_sample_tiny_hack", "fx_sample_tiny_masks"])
def test_sampler_matches_reference(name):
    z = gu.load(name)
    masks = gu.recorded_masks(z) if int(z["dropout"]) else None
    smp, trace = _sampler(z, masks)
    kw = {k: _t(z, k) for k in ("dynamical_condition", "static_condition") if k in z.files}
    out = smp.sample(_t(z, "x0"), **kw)
    ref = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    assert sorted(out) == sorted(ref)
    for k in ref:
        err = rel_l2(out[k], ref[k])
        assert err < 5e-6, f"{name}/{k}: rel L2 {err:.3e}"
    assert trace == json.loads(str(z["trace"]))
    if masks is not None:
        assert len(gu.masks_per_forward(masks, SFNOConfig(**json.loads(str(z["icfg"]))))) == 10


@pytest.mark.parametrize("name,n_dropout_calls", [("fx_sample_tiny_k2", 3), ("fx_sample_tiny_k2_every2nd", 1),
                                                   ("fx_sample_tiny_naive", 6)])
def test_sampler_outside_k0_matches_reference(name, n_dropout_calls):
    """The sampler with artificial diffusion steps (additional_interpolation_steps = 2: interpolation times 1/3, 2/3 before the
    first data step), the named schedule "every2nd", per-call dropout "except_dynamical_steps", and naive sampling -- fixtures
    from the reference's own DYffusion.sample (src/diffusion/dyffusion.py:134-188,226-235,367-455,467-520), masks recorded."""
    z = gu.load(name)
    masks = gu.recorded_masks(z)
    smp, trace = _sampler(z, masks)
    out = smp.sample(_t(z, "x0"), static_condition=_t(z, "static_condition"))
    ref = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out::")}
    assert sorted(out) == sorted(ref)
    assert trace == json.loads(str(z["trace"]))
    for k in ref:
        err = rel_l2(out[k], ref[k])
        assert err < 5e-6, f"{name}/{k}: rel L2 {err:.3e}"
    assert len(gu.masks_per_forward(masks, SFNOConfig(**json.loads(str(z["icfg"]))))) == n_dropout_calls


def test_call_trace_fixture():
    with open(os.path.join(gu.GOLDEN, "fx_trace.json")) as f:
        tr = json.load(f)
    assert [c for c, _ in tr].count("F") == 6 and [c for c, _ in tr].count("I") == 10
    assert tr[:5] == [["F", 0.0], ["I", 1.0], ["F", 1.0], ["I", 2.0], ["I", 1.0]]


def test_stepper_matches_reference():
    """oracle/stepper.py vs the reference's own run_on_batch_multistep (fixture fx_stepper_tiny)."""
    from contextlib import nullcontext

    from oracle.stepper import run_on_batch

    z = gu.load("fx_stepper_tiny")
    smp, _ = _sampler(_ZWrap(z, hack=1))

    class Mod:   # the stateful prediction cache of forecasting_multi_horizon.py:347-380, on the oracle sampler
        true_horizon = 6
        ema_scope = inference_dropout_scope = staticmethod(nullcontext)

        def __init__(self):
            self.cache = None

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            if horizon == 1:
                x = batch["dynamics"]
                self.cache = smp.sample(x, static_condition=batch["static_condition"])
            return {f"t{horizon}_preds_normed": self.cache[f"t{horizon}_preds"]}

    names = {k: json.loads(str(z[k])) for k in ("in_names", "out_names", "forcing_names")}
    data = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("data::")}
    means = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mean::")}
    stds = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("std::")}
    metrics, gen, gen_norm = run_on_batch(data, Mod(), names["in_names"], names["out_names"], names["forcing_names"],
                                          means, stds, int(z["n_steps"]), json.loads(str(z["prescriber"])), hack=True)
    for n in names["out_names"]:
        assert rel_l2(gen_norm[n], torch.from_numpy(z["gen_norm::" + n])) < 5e-6, n
        assert rel_l2(gen[n], torch.from_numpy(z["gen::" + n])) < 5e-6, n
    for k in z.files:
        if k.startswith("metric::"):
            assert abs(metrics[k[8:]] - float(z[k])) < 1e-4 * max(1.0, abs(float(z[k]))), k


def _loop_fixture():
    z, zw = gu.load("fx_loop_tiny"), gu.load("fx_stepper_tiny")    # the loop fixture reuses the stepper fixture's weights
    names = {k: json.loads(str(z[k])) for k in ("in_names", "out_names", "forcing_names")}
    series = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("series::")}
    means = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mean::")}
    stds = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("std::")}
    n_total, n_mem, members = int(z["n_total"]), int(z["n_mem_steps"]), int(z["members"])
    windows = [{k: v[:, i * n_mem:(i + 1) * n_mem + 1] for k, v in series.items()} for i in range(n_total // n_mem)]
    return z, zw, names, series, means, stds, n_total, n_mem, members, windows


def test_window_driver_matches_reference():
    """oracle/loop.py vs the reference's own run_inference + WindowStitcher (fixture fx_loop_tiny: 2 windows x 6 steps,
    2 samples, 2 members)."""
    from contextlib import nullcontext

    from oracle.loop import run_inference
    from oracle.stepper import run_on_batch

    z, zw, names, series, means, stds, n_total, n_mem, members, windows = _loop_fixture()
    smp, _ = _sampler(_ZWrap(zw, hack=1))

    class Mod:
        true_horizon = 6
        ema_scope = inference_dropout_scope = staticmethod(nullcontext)

        def __init__(self):
            self.cache = None

        def get_preds_at_t_for_batch(self, batch, horizon, **kw):
            if horizon == 1:
                self.cache = smp.sample(batch["dynamics"], static_condition=batch["static_condition"])
            return {f"t{horizon}_preds_normed": self.cache[f"t{horizon}_preds"]}

    pres = json.loads(str(z["prescriber"]))
    rob = lambda data, m: run_on_batch(data, Mod(), names["in_names"], names["out_names"], names["forcing_names"],  # noqa: E731
                                       means, stds, n_mem, pres, hack=True)
    wcalls, acalls = run_inference(windows, rob, n_total, n_mem, members)
    assert [c[0] for c in wcalls] == [int(v) for v in z["starts"]]
    assert [c[1] for c in acalls] == [int(v) for v in z["i_time_starts"]]
    for w, (_, pred) in enumerate(wcalls):
        for n in names["out_names"]:
            want = torch.from_numpy(z[f"pred{w}::{n}"])
            assert pred[n].shape == want.shape
            assert rel_l2(pred[n], want) < 5e-6, (w, n)
    for (loss, _), want in zip(acalls, z["losses"]):
        assert abs(loss - float(want)) < 1e-4 * max(1.0, abs(float(want)))


class _ZWrap:
    """npz view with an overridable scalar (the stepper fixture has no 'hack'/'dropout' entries)."""

    def __init__(self, z, **over):
        self.z, self.over = z, over
        self.files = list(z.files) + list(over)

    def __getitem__(self, k):
        return self.over[k] if k in self.over else self.z[k]


def test_ensemble_metrics_match_reference():
    """oracle/metrics.py vs the reference's own core/metrics.py functions (fixture fx_metrics)."""
    from oracle.metrics import ensemble_metrics

    z = gu.load("fx_metrics")
    got = ensemble_metrics(torch.from_numpy(z["truth"]), torch.from_numpy(z["pred"]), torch.from_numpy(z["weights"]))
    for k in ("rmse", "spread", "spread_skill_ratio", "crps", "bias"):
        want = torch.from_numpy(z[k]).double()
        assert got[k].shape == want.shape
        assert torch.allclose(got[k], want, rtol=2e-5, atol=1e-6), k


def test_oracle_time_mean_vs_reference_aggregator():
    """oracle.metrics.time_mean_maps vs the reference's own TimeMeanAggregator (fx_time_mean.npz: two windows, ensemble and
    deterministic), and the RMSE / bias the reference derives from the maps."""
    import json

    import numpy as np
    import torch

    from oracle.metrics import time_mean_maps, weighted_mean

    z = gu.load("fx_time_mean")
    names = json.loads(str(z["names"]))
    lats = torch.from_numpy(z["lats"])
    W = z["ens::gen_map::a"].shape[-1]
    w = torch.cos(torch.deg2rad(lats)).repeat(W, 1).t()
    w = w / w.sum()
    for key, ens in (("ens", True), ("det", False)):
        wins = [(int(z[f"{key}::i_time_start{i}"]), {n: torch.from_numpy(z[f"{key}::tgt{i}::{n}"]) for n in names},
                 {n: torch.from_numpy(z[f"{key}::gen{i}::{n}"]) for n in names}) for i in range(2)]
        gen, tgt = time_mean_maps(wins, ens)
        for n in names:
            assert np.allclose(gen[n].numpy(), z[f"{key}::gen_map::{n}"], rtol=1e-6, atol=1e-6)
            assert np.allclose(tgt[n].numpy(), z[f"{key}::target_map::{n}"], rtol=1e-6, atol=1e-6)
            rmse = float(weighted_mean((gen[n] - tgt[n]).double() ** 2, w.double()).sqrt())
            bias = float(weighted_mean((gen[n] - tgt[n]).double(), w.double()))
            assert abs(rmse - float(z[f"{key}::rmse::{n}"])) < 1e-6 and abs(bias - float(z[f"{key}::bias::{n}"])) < 1e-6



def test_oracle_mean_series_vs_reference_aggregator():
    """oracle.metrics.mean_series vs the reference's own MeanAggregator (fx_mean_series.npz: three windows, ensemble and
    deterministic; every metric of reduced.py:182-197 but the gradient magnitude)."""
    import json

    import numpy as np
    import torch

    from oracle.metrics import mean_series

    z = gu.load("fx_mean_series")
    names = json.loads(str(z["names"]))
    lats = torch.from_numpy(z["lats"])
    W = z["ens::tgt0::a"].shape[-1]
    w = torch.cos(torch.deg2rad(lats)).repeat(W, 1).t()
    w = w / w.sum()
    n_t = int(z["n_timesteps"])
    for key, ens in (("ens", True), ("det", False)):
        wins = [(int(z[f"{key}::i_time_start{i}"]), {n: torch.from_numpy(z[f"{key}::tgt{i}::{n}"]) for n in names},
                 {n: torch.from_numpy(z[f"{key}::gen{i}::{n}"]) for n in names}, w) for i in range(3)]
        got = mean_series(wins, ens, n_t)
        metrics = json.loads(str(z[f"{key}::metrics"]))
        assert ("weighted_crps" in metrics) == ens and "weighted_rmse" in metrics and len(metrics) == (8 if ens else 6)
        for m in metrics:
            for n in names:
                want = z[f"{key}::series::{m}/{n}"]
                assert want.shape == (n_t,)
                assert np.allclose(got[f"{m}/{n}"].numpy(), want, rtol=2e-5, atol=2e-6), (key, m, n)
