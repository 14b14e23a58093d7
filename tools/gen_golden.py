#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN CLASSES on CPU (build container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz (+ fx_trace.json)

The reference is imported from /root/reference through tools/ref_shims.py (stubs for the packages this image lacks;
`torch_harmonics` is supplied by oracle/sht.py -- see the shim header).  Nothing here is needed at test time: the
fixtures are plain data (inputs, weights, expected outputs, recorded dropout masks).

Weights are "trained-like" (oracle.sfno.make_state_dict) and are loaded into the reference network with
strict=True, which also pins the state_dict contract (names + shapes) of SURVEY.md Appendix B.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shims  # noqa: E402

ref_shims.install()
from ref_shims import AttrDict  # noqa: E402

from oracle.sfno import SFNOConfig, make_state_dict  # noqa: E402
from src.models.modules.drop_path import DropPath  # noqa: E402
from src.models.sfno.sfnonet import SphericalFourierNeuralOperatorNet as RefSFNO  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def np_sd(sd):
    return {"sd::" + k: v.detach().cpu().numpy() for k, v in sd.items()}


def pack(mask: torch.Tensor) -> np.ndarray:
    return np.packbits(mask.detach().cpu().numpy().astype(bool).reshape(-1))


def ref_net(cfg: SFNOConfig, n_in: int, n_cond: int, seed: int):
    net = RefSFNO(
        num_input_channels=n_in, num_output_channels=cfg.out_chans, num_conditional_channels=n_cond,
        spatial_shape_in=(cfg.nlat, cfg.nlon), spatial_shape_out=(cfg.nlat, cfg.nlon), loss_function=None,
        embed_dim=cfg.embed_dim, num_layers=cfg.num_layers, operator_type="dhconv", filter_type="linear",
        scale_factor=1, use_mlp=True, mlp_ratio=cfg.mlp_ratio, dropout_mlp=cfg.dropout_mlp,
        drop_path_rate=cfg.drop_path_rate, normalization_layer="instance_norm", with_time_emb=cfg.with_time_emb,
        data_grid=cfg.data_grid, big_skip=cfg.big_skip, pos_embed=cfg.pos_embed, verbose=False,
    )
    sd = make_state_dict(cfg, seed=seed)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    # non-persistent SHT buffers are the only thing allowed to be absent
    assert not unexpected, unexpected
    assert all(".weights" in k or ".pct" in k for k in missing), missing
    ref_keys = {k for k in net.state_dict().keys() if ".weights" not in k and ".pct" not in k}
    assert ref_keys == set(sd.keys()), (sorted(ref_keys ^ set(sd.keys())))
    if cfg.with_time_emb:
        net.set_min_max_time(cfg.min_time, cfg.max_time)
    net.eval()
    return net, sd


class MaskRecorder:
    """Records the keep-masks the reference's nn.Dropout / DropPath layers actually drew (out != 0)."""

    def __init__(self, net):
        self.records = []
        self.handles = []
        for name, m in net.named_modules():
            if isinstance(m, torch.nn.Dropout):
                self.handles.append(m.register_forward_hook(self._hook(name, "elem")))
            elif isinstance(m, DropPath):
                self.handles.append(m.register_forward_hook(self._hook(name, "path")))

    def _hook(self, name, kind):
        def fn(mod, inp, out):
            if not mod.training:
                return
            if kind == "elem":
                self.records.append((name, (out != 0) | (inp[0] == 0)))
            else:
                keep = (out.flatten(1) != 0).any(dim=1) | (inp[0].flatten(1) == 0).all(dim=1)
                self.records.append((name, keep))
        return fn

    def remove(self):
        for h in self.handles:
            h.remove()


def gen_sfno(tag, cfg, n_in, n_cond, B, times, seed, with_masks):
    net, sd = ref_net(cfg, n_in, n_cond, seed)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(B, n_in, cfg.nlat, cfg.nlon, generator=g)
    cond = torch.randn(B, n_cond, cfg.nlat, cfg.nlon, generator=g) if n_cond else None
    t = torch.tensor(times, dtype=torch.float32) if cfg.with_time_emb else None
    blk_io = {}
    h = net.blocks[0].register_forward_hook(
        lambda m, i, o: blk_io.update(x=i[0].detach().clone(), y=o.detach().clone()))
    with torch.no_grad():
        y, trepr = net(x, time=t, condition=cond, return_time_emb=True)
    h.remove()
    out = dict(np_sd(sd), x=x.numpy(), y=y.numpy(), block0_in=blk_io["x"].numpy(), block0_out=blk_io["y"].numpy(),
               cfg=json.dumps({**cfg.__dict__, "n_in": n_in, "n_cond": n_cond}))
    if cond is not None:
        out["cond"] = cond.numpy()
    if t is not None:
        out["time"] = t.numpy()
        out["t_repr"] = trepr.numpy()
    if with_masks:
        net.enable_inference_dropout()       # _base_model.py:288-290 -> utils.enable_inference_dropout
        rec = MaskRecorder(net)
        torch.manual_seed(777)
        with torch.no_grad():
            yd = net(x, time=t, condition=cond)
        rec.remove()
        net.disable_inference_dropout()
        out["y_dropout"] = yd.numpy()
        out["mask_names"] = json.dumps([n for n, _ in rec.records])
        for i, (_, m) in enumerate(rec.records):
            out[f"mask{i}"] = pack(m)
            out[f"mask{i}_shape"] = np.array(m.shape)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: y std {float(y.std()):.4f}, saved")


def _weights_digest(sd):
    """Checksum of a seeded state_dict (the full-width fixtures store the SEED, not 100+ MB of weights): the test refuses
    to compare if its own make_state_dict(seed) does not reproduce these numbers."""
    return np.array([float(sum(v.double().abs().sum() for v in sd.values())),
                     float(sum((v.double() ** 2).sum() for v in sd.values()))])


def gen_sfno_full(tag="fx_sfno_full"):
    """The REFERENCE network at production width on the production grid: 180 x 360, E = 256, hidden 512, 68 + 2 -> 34
    channels, ALL 8 blocks (BASELINE.json configs[1] = the interpolator, whole): the first (equiangular -> Legendre-Gauss), six inner
    (LG -> LG) and the last (LG -> equiangular), time embedding, B = 1, dropout off.  These
    sizes reach every production kernel (`mlp_h3`, `conv_h3`, `dh_h3`, `leg_par`, `fft360`, `pair_h3`); the small fixtures
    above only reach the generic tile kernels.  Stored: seeds, checksums of the seeded weights / inputs, and the
    34 x 180 x 360 output of the reference (8.8 MB)."""
    cfg = SFNOConfig(in_chans=70, out_chans=34, nlat=180, nlon=360, embed_dim=256, num_layers=8, with_time_emb=True,
                     min_time=1.0, max_time=5.0)
    seed_w, seed_x = 4321, 1234
    net, sd = ref_net(cfg, 68, 2, seed_w)
    g = torch.Generator(device="cpu").manual_seed(seed_x)
    x = torch.randn(1, 68, cfg.nlat, cfg.nlon, generator=g)
    cond = torch.randn(1, 2, cfg.nlat, cfg.nlon, generator=g)
    t = torch.tensor([3.0])
    with torch.no_grad():
        y = net(x, time=t, condition=cond)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), y=y.numpy(), time=t.numpy(), seed_w=np.array(seed_w),
                        seed_x=np.array(seed_x), weights_digest=_weights_digest(sd),
                        inputs_digest=np.array([float(x.double().abs().sum()), float(cond.double().abs().sum())]),
                        cfg=json.dumps({**cfg.__dict__, "n_in": 68, "n_cond": 2}))
    print(f"{tag}: y std {float(y.std()):.4f}, |y| max {float(y.abs().max()):.3f}, saved")


def gen_sfno_wide_masks(tag="fx_sfno_wide_masks"):
    """The reference network at production WIDTH (E = 256, hidden 512: the fused MLP kernel's shape) on the small grid, with
    dropout and drop path ON and the masks its nn.Dropout / DropPath layers drew recorded -- injected into the product, they
    drive the fused `mlp_h3` kernel with the reference's own random decisions (no Philox on either side).  3 blocks, B = 2."""
    cfg = SFNOConfig(in_chans=10, out_chans=6, nlat=32, nlon=64, embed_dim=256, num_layers=3, with_time_emb=True,
                     dropout_mlp=0.1, drop_path_rate=0.3, min_time=0.0, max_time=5.0)
    seed_w, seed_x = 4321, 1234
    net, sd = ref_net(cfg, 8, 2, seed_w)
    g = torch.Generator(device="cpu").manual_seed(seed_x)
    x = torch.randn(2, 8, cfg.nlat, cfg.nlon, generator=g)
    cond = torch.randn(2, 2, cfg.nlat, cfg.nlon, generator=g)
    t = torch.tensor([1.0, 4.0])
    with torch.no_grad():
        y = net(x, time=t, condition=cond)
    net.enable_inference_dropout()
    rec = MaskRecorder(net)
    torch.manual_seed(777)
    with torch.no_grad():
        yd = net(x, time=t, condition=cond)
    rec.remove()
    out = dict(y=y.numpy(), y_dropout=yd.numpy(), time=t.numpy(), seed_w=np.array(seed_w), seed_x=np.array(seed_x),
               weights_digest=_weights_digest(sd),
               inputs_digest=np.array([float(x.double().abs().sum()), float(cond.double().abs().sum())]),
               cfg=json.dumps({**cfg.__dict__, "n_in": 8, "n_cond": 2}), mask_names=json.dumps([n for n, _ in rec.records]))
    for i, (_, m) in enumerate(rec.records):
        out[f"mask{i}"] = pack(m)
        out[f"mask{i}_shape"] = np.array(m.shape)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: y std {float(y.std()):.4f}, dropout changes y by {float((yd - y).norm() / y.norm()):.3f}, "
          f"{len(rec.records)} masks, saved")


def build_experiments(C, n_forc, H, W, E, L, hack, dropout, seed_f, seed_i, extra=None, ipol_min_time=1.0):
    import src.experiment_types._base_experiment as be
    from src.experiment_types.forecasting_multi_horizon import MultiHorizonForecastingDYffusion
    from src.experiment_types.interpolation import InterpolationExperiment

    be.get_dims_of_dataset = lambda dc: {"input": len(dc.in_names), "output": len(dc.out_names), "spatial_in": (H, W),
                                          "spatial_out": (H, W), "conditional": len(dc.forcing_names)}
    cs = C + (1 if hack else 0)
    dm = AttrDict(_target_="src.datamodules.fv3gfs_ensemble.FV3GFSEnsembleDataModule",
                  in_names=[f"v{i}" for i in range(cs)], out_names=[f"v{i}" for i in range(cs - C, cs)],
                  forcing_names=[f"f{i}" for i in range(n_forc)], window=1, horizon=6)

    def mcfg(**kw):
        return AttrDict(_target_="src.models.sfno.sfnonet.SphericalFourierNeuralOperatorNet", embed_dim=E, num_layers=L,
                        operator_type="dhconv", filter_type="linear", scale_factor=1, use_mlp=True, mlp_ratio=2.0,
                        normalization_layer="instance_norm", with_time_emb=True, data_grid="equiangular",
                        loss_function=None, verbose=False, **kw)

    ipol = InterpolationExperiment(model_config=mcfg(dropout_mlp=0.1 if dropout else 0.0,
                                                     drop_path_rate=0.1 if dropout else 0.0),
                                   datamodule_config=dm, enable_inference_dropout=True, verbose=False)
    icfg = SFNOConfig(in_chans=2 * cs + n_forc, out_chans=C, nlat=H, nlon=W, embed_dim=E, num_layers=L,
                      with_time_emb=True, dropout_mlp=0.1 if dropout else 0.0, drop_path_rate=0.1 if dropout else 0.0,
                      min_time=ipol_min_time, max_time=5.0)
    if ipol_min_time != 1.0:
        # InterpolationExperiment.__init__ pins the network's valid time range to the data time steps [1, horizon - 1]
        # (src/experiment_types/interpolation.py:24-25,27-31) and the network asserts it (sfnonet.py:780-782): sampling with
        # artificial steps (interpolation times in (0, 1)) needs an interpolator whose range was opened, as a user with a
        # continuous-time interpolator would do
        ipol.model.set_min_max_time(min_time=ipol_min_time, max_time=5.0)
    assert ipol.model.in_chans == icfg.in_chans and ipol.model.out_chans == C
    isd = make_state_dict(icfg, seed=seed_i)
    ipol.model.load_state_dict(isd, strict=False)
    dcfg = AttrDict(_target_="src.diffusion.dyffusion.DYffusion", timesteps=6, forward_conditioning="none",
                    interpolator=ipol, interpolator_local_checkpoint_path=None, time_encoding="dynamics",
                    hack_for_imprecise_interpolation=hack,
                    **{"enable_interpolator_dropout": bool(dropout), **(extra or {})})
    fc = MultiHorizonForecastingDYffusion(model_config=mcfg(), datamodule_config=dm, diffusion_config=dcfg,
                                          verbose=False)
    fcfg = SFNOConfig(in_chans=cs + n_forc, out_chans=C, nlat=H, nlon=W, embed_dim=E, num_layers=L,
                      with_time_emb=True, min_time=0.0, max_time=5.0)
    assert fc.model.model.in_chans == fcfg.in_chans
    fsd = make_state_dict(fcfg, seed=seed_f)
    fc.model.model.load_state_dict(fsd, strict=False)
    fc.eval()
    ipol.eval()
    return fc, ipol, fcfg, icfg, fsd, isd, cs


def gen_sample(tag, hack, dropout, extra=None, ipol_min_time=1.0):
    C, n_forc, H, W, E, L = 6, 2, 32, 64, 16, 2
    fc, ipol, fcfg, icfg, fsd, isd, cs = build_experiments(C, n_forc, H, W, E, L, hack, dropout, 11, 22, extra,
                                                           ipol_min_time=ipol_min_time)
    g = torch.Generator(device="cpu").manual_seed(1234)
    B = 2
    x0 = torch.randn(B, cs, H, W, generator=g)
    if hack:
        kw = {"static_condition": torch.randn(B, n_forc, H, W, generator=g)}
    else:
        kw = {"dynamical_condition": torch.randn(B, 7, n_forc, H, W, generator=g)}
    trace = []
    f_net, i_net = fc.model.model, ipol.model
    hf = f_net.register_forward_pre_hook(lambda m, a, k: trace.append(["F", float(k["time"][0])]), with_kwargs=True)
    # (the third entry -- is the interpolator's dropout on for this call -- only where a fixture switches it per call)
    per_call_dropout = (extra or {}).get("enable_interpolator_dropout") == "except_dynamical_steps"
    hi = i_net.register_forward_pre_hook(
        lambda m, a, k: trace.append(["I", float(k["time"][0])] + ([bool(m.blocks[0].mlp.fwd[2].training)] if per_call_dropout else [])),
        with_kwargs=True)
    rec = MaskRecorder(i_net) if dropout else None
    torch.manual_seed(4242)
    res = fc.model.sample(x0, **kw)     # DYffusion.sample (dyffusion.py:569-572)
    hf.remove()
    hi.remove()
    out = dict(x0=x0.numpy(), hack=np.array(int(hack)), dropout=np.array(int(dropout)),
               fcfg=json.dumps(fcfg.__dict__), icfg=json.dumps(icfg.__dict__), trace=json.dumps(trace))
    if extra:
        out["diffusion_extra"] = json.dumps(extra)
    out.update({"f::" + k: v.numpy() for k, v in fsd.items()})
    out.update({"i::" + k: v.numpy() for k, v in isd.items()})
    for k, v in kw.items():
        out[k] = v.numpy()
    for k, v in res.items():
        out["out::" + k] = v.numpy()
    if rec is not None:
        rec.remove()
        out["mask_names"] = json.dumps([n for n, _ in rec.records])
        for i, (_, m) in enumerate(rec.records):
            out[f"mask{i}"] = pack(m)
            out[f"mask{i}_shape"] = np.array(m.shape)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: keys {sorted(res.keys())}, {len(trace)} network calls, saved")
    return trace

    # also exercise the stepper-facing surface once (get_preds_at_t_for_batch), results must equal sample()


def gen_sample_refine():
    """`refine_intermediate_predictions=True` (src/diffusion/dyffusion.py:551-563: a second interpolator sweep from the last
    forecast) with the carried input-only channel; 6 + 10 + 5 network calls."""
    return gen_sample("fx_sample_tiny_refine", hack=True, dropout=False, extra=dict(refine_intermediate_predictions=True))


def gen_sample_artificial():
    """The sampler OUTSIDE the shipped k = 0 configuration (src/diffusion/dyffusion.py:134-188 step map, :226-235 per-call
    dropout rule, :363-455 named schedules, :467-520 loop): additional_interpolation_steps = 2 puts two artificial diffusion
    steps (interpolation times 1/3, 2/3) in front of the first data step.
      fx_sample_tiny_k2          full schedule [0 .. 7], dropout "except_dynamical_steps" (on only for the calls of a step that
                                 lands on an artificial time), masks recorded
      fx_sample_tiny_k2_every2nd the same with sampling_schedule="every2nd" (artificial step 1 of [1, 2] joins)
      fx_sample_tiny_naive       sampling_type="naive", one artificial step, dropout always on, masks recorded"""
    k2 = dict(additional_interpolation_steps=2, enable_interpolator_dropout="except_dynamical_steps")
    t1 = gen_sample("fx_sample_tiny_k2", hack=True, dropout=True, extra=k2, ipol_min_time=0.0)
    t2 = gen_sample("fx_sample_tiny_k2_every2nd", hack=True, dropout=True, extra=dict(k2, sampling_schedule="every2nd"),
                    ipol_min_time=0.0)
    t3 = gen_sample("fx_sample_tiny_naive", hack=True, dropout=True,
                    extra=dict(additional_interpolation_steps=1, sampling_type="naive"), ipol_min_time=0.0)
    return t1, t2, t3


def gen_stepper(tag="fx_stepper_tiny"):
    """The reference's own `run_on_batch_multistep` (src/ace_inference/core/stepper_multistep.py:298-466): normalise, pack,
    autoregressive loop over n_forward_steps with the DYffusion module, prescriber, denormalise."""
    from src.ace_inference.core.normalizer import StandardNormalizer
    from src.ace_inference.core.optimization import NullOptimization
    from src.ace_inference.core.prescriber import Prescriber
    from src.ace_inference.core.stepper_multistep import run_on_batch_multistep
    from src.ace_inference.training.utils.darcy_loss import LpLoss
    from src.utilities.packer import Packer

    C, n_forc, H, W, E, L = 6, 2, 32, 64, 16, 2
    fc, ipol, fcfg, icfg, fsd, isd, cs = build_experiments(C, n_forc, H, W, E, L, True, False, 11, 22)
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]     # HGTsfc is input-only (the "imprecise" case)
    out_names = in_names[1:]
    forcing_names = [f"f{i}" for i in range(n_forc)]
    mask_name = "ocean_fraction"
    n_steps, B = 8, 2
    g = torch.Generator(device="cpu").manual_seed(2024)
    names = in_names + forcing_names
    means = {n: torch.randn((), generator=g) * 3.0 for n in names}
    stds = {n: torch.rand((), generator=g) * 2.0 + 0.5 for n in names}
    data = {n: torch.randn(B, n_steps + 1, H, W, generator=g) * stds[n] + means[n] for n in names}
    data[mask_name] = torch.rand(B, n_steps + 1, H, W, generator=g)
    pres = Prescriber(prescribed_name="v2", mask_name=mask_name, mask_value=1, interpolate=False)

    class _NullAgg:
        def record_batch(self, *a, **k):
            pass

    axis = -3
    stepped = run_on_batch_multistep(
        data={k: v.clone() for k, v in data.items()}, module=fc, normalizer=StandardNormalizer(means, stds),
        in_packer=Packer(in_names, axis=axis), out_packer=Packer(out_names, axis=axis),
        forcings_packer=Packer(forcing_names, axis=axis), optimization=NullOptimization(), loss_obj=LpLoss(),
        prescriber=pres, aggregator=_NullAgg(), n_forward_steps=n_steps)
    out = dict(in_names=json.dumps(in_names), out_names=json.dumps(out_names), forcing_names=json.dumps(forcing_names),
               prescriber=json.dumps(pres.get_state()), n_steps=np.array(n_steps),
               fcfg=json.dumps(fcfg.__dict__), icfg=json.dumps(icfg.__dict__))
    out.update({"f::" + k: v.numpy() for k, v in fsd.items()})
    out.update({"i::" + k: v.numpy() for k, v in isd.items()})
    out.update({"data::" + k: v.numpy() for k, v in data.items()})
    out.update({"mean::" + k: v.numpy() for k, v in means.items()})
    out.update({"std::" + k: v.numpy() for k, v in stds.items()})
    out.update({"gen::" + k: v.numpy() for k, v in stepped.gen_data.items()})
    out.update({"gen_norm::" + k: v.numpy() for k, v in stepped.gen_data_norm.items()})
    out.update({"metric::" + k: np.asarray(float(v)) for k, v in stepped.metrics.items()})
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: loss {float(stepped.metrics['loss']):.5f}, gen vars {sorted(stepped.gen_data)[:3]}..., saved")


def gen_loop(tag="fx_loop_tiny"):
    """The reference's own window driver `run_inference` + `WindowStitcher` (src/ace_inference/inference/loop.py:26-264):
    two windows of 6 steps, 2 samples, 2 ensemble members (dropout off: the fixture pins the stitching / carry-over /
    stacking logic, the stochastic part is covered by the device-vs-oracle tests)."""
    import types

    import src.ace_inference.inference.loop as L
    from src.ace_inference.core.aggregator.null import NullAggregator
    from src.ace_inference.core.normalizer import StandardNormalizer
    from src.ace_inference.core.prescriber import Prescriber
    from src.ace_inference.core.stepper_multistep import run_on_batch_multistep
    from src.ace_inference.training.utils.darcy_loss import LpLoss
    from src.utilities.packer import Packer

    C, n_forc, H, W, E, Lr = 6, 2, 32, 64, 16, 2
    fc, ipol, fcfg, icfg, fsd, isd, cs = build_experiments(C, n_forc, H, W, E, Lr, True, False, 11, 22)
    in_names = ["HGTsfc"] + [f"v{i}" for i in range(1, cs)]
    out_names = in_names[1:]
    forcing_names = [f"f{i}" for i in range(n_forc)]
    mask_name = "ocean_fraction"
    n_total, n_mem_steps, B, members = 12, 6, 2, 2
    g = torch.Generator(device="cpu").manual_seed(777)
    names = in_names + forcing_names
    means = {n: torch.randn((), generator=g) * 3.0 for n in names}
    stds = {n: torch.rand((), generator=g) * 2.0 + 0.5 for n in names}
    series = {n: torch.randn(B, n_total + 1, H, W, generator=g) * stds[n] + means[n] for n in names}
    series[mask_name] = torch.rand(B, n_total + 1, H, W, generator=g)
    pres = Prescriber(prescribed_name="v2", mask_name=mask_name, mask_value=1, interpolate=False)
    axis = -3

    class _Stepper:   # what run_inference needs of MultiStepStepper: .module and .run_on_batch
        module = fc

        def run_on_batch(self, data, optimization, n_forward_steps=1, aggregator=None):
            return run_on_batch_multistep(
                data=data, module=fc, normalizer=StandardNormalizer(means, stds), in_packer=Packer(in_names, axis=axis),
                out_packer=Packer(out_names, axis=axis), forcings_packer=Packer(forcing_names, axis=axis),
                optimization=optimization, loss_obj=LpLoss(), prescriber=pres,
                aggregator=aggregator if aggregator is not None else NullAggregator(), n_forward_steps=n_forward_steps)

    class _Times:   # the only thing the loop does with xr.DataArray times: .isel(time=slice(1, None))
        def __init__(self, idx):
            self.idx = list(idx)

        def isel(self, time):
            return _Times(self.idx[time])

    windows = [types.SimpleNamespace(data={k: v[:, i * n_mem_steps:(i + 1) * n_mem_steps + 1].clone() for k, v in series.items()},
                                     times=_Times(range(i * n_mem_steps, (i + 1) * n_mem_steps + 1)))
               for i in range(n_total // n_mem_steps)]
    data = types.SimpleNamespace(loader=windows, sigma_coordinates=None)
    rec = []

    class _Writer:
        def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
            rec.append((int(start_timestep), {k: v.clone() for k, v in prediction.items()},
                        {k: v.clone() for k, v in target.items()}))

    losses = []

    class _Agg:   # (with the reference's NullAggregator an ensemble run dies on `stepped.metrics["loss"]`, loop.py:145,209)
        def record_batch(self, loss, target_data, gen_data, target_data_norm, gen_data_norm, i_time_start=0):
            losses.append((float(loss), int(i_time_start)))

    L.compute_derived_quantities = lambda d, s: d     # derived variables are outside the fixture's scope
    L.run_inference(_Agg(), _Stepper(), data, n_total, n_mem_steps, members, "cpu", writer=_Writer())
    out = dict(in_names=json.dumps(in_names), out_names=json.dumps(out_names), forcing_names=json.dumps(forcing_names),
               prescriber=json.dumps(pres.get_state()), n_total=np.array(n_total), n_mem_steps=np.array(n_mem_steps),
               members=np.array(members), fcfg=json.dumps(fcfg.__dict__), icfg=json.dumps(icfg.__dict__),
               starts=np.array([r[0] for r in rec]), losses=np.array([l[0] for l in losses]),
               i_time_starts=np.array([l[1] for l in losses]))
    # network weights: identical to fx_stepper_tiny.npz (same build_experiments seeds), not stored twice
    out.update({"series::" + k: v.numpy() for k, v in series.items()})
    out.update({"mean::" + k: v.numpy() for k, v in means.items()})
    out.update({"std::" + k: v.numpy() for k, v in stds.items()})
    for w, (st, pred, tgt) in enumerate(rec):
        out.update({f"pred{w}::" + k: v.numpy() for k, v in pred.items()})
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: writer calls at {[r[0] for r in rec]}, prediction shapes "
          f"{[tuple(next(iter(r[1].values())).shape) for r in rec]}, saved")


def gen_ckpt_layout(tag="fx_ckpt_layout"):
    """What a Lightning checkpoint of the two reference experiments looks like (no tensors: key names and
    hyper-parameters only): `state_dict` keys of `MultiHorizonForecastingDYffusion` / `InterpolationExperiment`, the
    `LitEma` buffer name of every parameter (src/models/modules/ema.py:20-27) and the `hyper_parameters` dictionaries
    (`_base_experiment.py:75-100`).  The weights of the matching fixture are those of fx_stepper_tiny."""
    from src.models.modules.ema import LitEma

    fc, ipol, fcfg, icfg, fsd, isd, cs = build_experiments(6, 2, 32, 64, 16, 2, True, True, 11, 22)

    def clean(o):
        if isinstance(o, dict):
            return {k: clean(v) for k, v in o.items() if k != "interpolator"}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (int, float, str, bool)) or o is None:
            return o
        return None

    def ema_map(handle):
        e = LitEma(handle, decay=0.9999)
        return dict(e.m_name2s_name)

    out = {
        "forecaster": {"hyper_parameters": clean(dict(fc.hparams)),
                       "state_dict_keys": [k for k in fc.state_dict().keys() if not k.startswith("model.interpolator")],
                       "ema_names": ema_map(fc.model), "ema_extra": ["decay", "num_updates"]},
        "interpolator": {"hyper_parameters": clean(dict(ipol.hparams)), "state_dict_keys": list(ipol.state_dict().keys()),
                         "ema_names": ema_map(ipol.model), "ema_extra": ["decay", "num_updates"]},
    }
    with open(os.path.join(OUT, f"{tag}.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(f"{tag}: {len(out['forecaster']['state_dict_keys'])} + {len(out['interpolator']['state_dict_keys'])} keys, saved")


def gen_metrics(tag="fx_metrics"):
    """The reference's own ensemble diagnostics (src/ace_inference/core/metrics.py) on a small ensemble."""
    from src.ace_inference.core import metrics as M

    g = torch.Generator(device="cpu").manual_seed(31)
    E, S, T, H, W = 5, 2, 3, 16, 32
    lats = torch.linspace(-84.375, 84.375, H)
    w = M.spherical_area_weights(lats, W)
    truth = torch.randn(S, T, H, W, generator=g) * 2.0 + 1.0
    pred = truth[None] + torch.randn(E, S, T, H, W, generator=g) * 0.7 + 0.1
    dim = (-2, -1)
    out = dict(lats=lats.numpy(), weights=w.numpy(), truth=truth.numpy(), pred=pred.numpy(),
               rmse=M.root_mean_squared_error(truth, pred.mean(0), w, dim=dim).numpy(),
               spread=M.ensemble_spread(pred, w, dim=dim).numpy(),
               spread_skill_ratio=M.spread_skill_ratio(truth, pred, w, dim=dim).numpy(),
               crps=M.weighted_crps(truth, pred, w, dim=dim).numpy(),
               bias=M.weighted_mean_bias(truth, pred.mean(0), w, dim=dim).numpy())
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: crps[0] {out['crps'][0]}, saved")


def gen_time_mean(tag="fx_time_mean"):
    """The reference's own TimeMeanAggregator (src/ace_inference/core/aggregator/inference/time_mean.py) fed two windows of
    an ensemble run the way run_inference feeds it (loop.py:133-149): time-mean maps, their RMSE and bias."""
    from src.ace_inference.core import metrics as M
    from src.ace_inference.core.aggregator.inference.time_mean import TimeMeanAggregator

    class NoDist:            # single process: reduce_mean is the identity (core/distributed.py)
        def reduce_mean(self, t):
            return t

    g = torch.Generator(device="cpu").manual_seed(41)
    E, S, T, H, W = 3, 2, 4, 16, 32
    names = ["a", "b"]
    lats = torch.linspace(-84.375, 84.375, H)
    w = M.spherical_area_weights(lats, W)
    out = dict(lats=lats.numpy(), names=json.dumps(names))
    for is_ens in (True, False):
        agg = TimeMeanAggregator(w, dist=NoDist(), is_ensemble=is_ens)
        key = "ens" if is_ens else "det"
        for win, (i_time_start, nt) in enumerate(((0, T + 1), (T + 1, T))):
            tgt = {n: torch.randn(S, nt, H, W, generator=g) * 2.0 + 1.0 for n in names}
            shp = (E, S, nt, H, W) if is_ens else (S, nt, H, W)
            gen = {n: (tgt[n][None] if is_ens else tgt[n]) + torch.randn(*shp, generator=g) * 0.5 + 0.2 for n in names}
            agg.record_batch(loss=0.0, target_data=tgt, gen_data=gen, target_data_norm=tgt, gen_data_norm=gen,
                             i_time_start=i_time_start)
            for n in names:
                out[f"{key}::tgt{win}::{n}"] = tgt[n].numpy()
                out[f"{key}::gen{win}::{n}"] = gen[n].numpy()
            out[f"{key}::i_time_start{win}"] = i_time_start
        for pr in agg._get_target_gen_pairs():
            out[f"{key}::gen_map::{pr.name}"] = pr.gen.numpy()
            out[f"{key}::target_map::{pr.name}"] = pr.target.numpy()
            out[f"{key}::rmse::{pr.name}"] = pr.rmse(weights=w)
            out[f"{key}::bias::{pr.name}"] = pr.weighted_mean_bias(weights=w)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: ens rmse a {out['ens::rmse::a']}, saved")


def gen_mean_series(tag="fx_mean_series"):
    """The reference's own MeanAggregator (src/ace_inference/core/aggregator/inference/reduced.py:144-266: the per-timestep
    series of area-weighted metrics) fed three windows the way run_inference feeds it (loop.py:133-149), ensemble and
    deterministic.  The gradient-magnitude metric is not stored (out of the build's scope)."""
    from src.ace_inference.core import metrics as M
    from src.ace_inference.core.aggregator.inference.reduced import MeanAggregator

    class NoDist:            # single process: reduce_mean is the identity (core/distributed.py)
        def reduce_mean(self, t):
            return t

    g = torch.Generator(device="cpu").manual_seed(43)
    E, S, T, H, W = 4, 2, 3, 16, 32
    names = ["a", "b"]
    lats = torch.linspace(-84.375, 84.375, H)
    w = M.spherical_area_weights(lats, W)
    n_timesteps = 1 + 3 * T
    out = dict(lats=lats.numpy(), names=json.dumps(names), n_timesteps=n_timesteps)
    for is_ens in (True, False):
        agg = MeanAggregator(w, target="denorm", n_timesteps=n_timesteps, is_ensemble=is_ens, dist=NoDist(),
                             device=torch.device("cpu"))
        key = "ens" if is_ens else "det"
        i_time = 0
        for win in range(3):
            nt = T + 1 if win == 0 else T           # the first window keeps its initial condition (loop.py:133-141)
            tgt = {n: torch.randn(S, nt, H, W, generator=g) * 2.0 + 1.0 for n in names}
            shp = (E, S, nt, H, W) if is_ens else (S, nt, H, W)
            gen = {n: (tgt[n][None] if is_ens else tgt[n]) + torch.randn(*shp, generator=g) * 0.5 + 0.2 for n in names}
            agg.record_batch(loss=0.0, target_data=tgt, gen_data=gen, target_data_norm=tgt, gen_data_norm=gen,
                             i_time_start=i_time)
            for n in names:
                out[f"{key}::tgt{win}::{n}"] = tgt[n].numpy()
                out[f"{key}::gen{win}::{n}"] = gen[n].numpy()
            out[f"{key}::i_time_start{win}"] = i_time
            i_time += nt
        metrics = []
        for d in agg._get_series_data():
            if "grad_mag" in d.metric_name:
                continue
            out[f"{key}::series::{d.metric_name}/{d.var_name}"] = np.asarray(d.data, dtype=np.float64)
            metrics.append(d.metric_name)
        out[f"{key}::metrics"] = json.dumps(sorted(set(metrics)))
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}: ens weighted_crps/a {out['ens::series::weighted_crps/a'][:3]}, saved")


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:      # regenerate selected fixtures only: python tools/gen_golden.py gen_time_mean
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    # C1: one block, 32x64, 8 channels (BASELINE.json configs[0])
    gen_sfno("fx_block_c1", SFNOConfig(in_chans=8, out_chans=8, nlat=32, nlon=64, embed_dim=8, num_layers=1,
                                       with_time_emb=True, min_time=0.0, max_time=5.0), 8, 0, 2, [1.0, 4.0], 4321, False)
    # tiny full network with conditioning, time embedding, dropout + drop path (interpolator-like)
    gen_sfno("fx_sfno_tiny", SFNOConfig(in_chans=10, out_chans=6, nlat=32, nlon=64, embed_dim=16, num_layers=3,
                                        with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.3, min_time=0.0,
                                        max_time=5.0), 8, 2, 3, [0.0, 2.0, 5.0], 4321, True)
    gen_sfno("fx_sfno_tiny_lg", SFNOConfig(in_chans=4, out_chans=4, nlat=32, nlon=64, embed_dim=8, num_layers=2,
                                           with_time_emb=False, data_grid="legendre-gauss", big_skip=False,
                                           pos_embed=False), 4, 0, 2, None, 99, False)
    t1 = gen_sample("fx_sample_tiny", hack=False, dropout=False)
    gen_sample("fx_sample_tiny_hack", hack=True, dropout=False)
    gen_sample("fx_sample_tiny_masks", hack=True, dropout=True)
    gen_sample_refine()
    gen_sample_artificial()
    with open(os.path.join(OUT, "fx_trace.json"), "w") as f:
        json.dump(t1, f)
    gen_stepper()
    gen_loop()
    gen_ckpt_layout()
    gen_metrics()
    gen_time_mean()
    gen_mean_series()
    gen_sfno_wide_masks()
    gen_sfno_full()
    sizes = {n: os.path.getsize(os.path.join(OUT, n)) for n in sorted(os.listdir(OUT))}
    print(sizes)
