"""Synthetic workloads for measurement: networks with trained-like random weights and standardised random fields.

There is no network access for the published checkpoints or the FV3GFS data, so `bench.py` and `tools/c4_rollout.py` run
the architecture of the shipped YAML (`src/configs/model/sfno.yaml`: embed 256, 8 blocks, dhconv, instance norm, MLP
ratio 2, time embedding) with random weights of trained-like magnitude on N(0, 1) fields (the data are per-variable
standardised, `src/ace_inference/core/normalizer.py:96-102`).  Nothing here imports `oracle/` or `tests/`: the parity tests
hold `trained_like_state_dict` to the oracle's generator value for value (`tests/test_host_logic.py`), so a benchmark
network and a parity-test network of the same seed are the same network.
"""
from __future__ import annotations

import math
import types
from typing import Dict, Iterator, List, Optional, Tuple

import torch

from .experiment import InterpolationExperiment, MultiHorizonForecastingDYffusion
from .sfno import SphericalFourierNeuralOperatorNet
from .stepper import MultiStepStepper


def trained_like_state_dict(net: SphericalFourierNeuralOperatorNet, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """Random weights under the reference's `state_dict` names with trained-like magnitudes, so that
    every branch of the block is numerically visible: the reference initialises the dhconv weights at scale 1 / E^2
    (`src/models/sfno/s2convolutions.py:70-71,146`), next to which the skip path hides the spectral branch; here dhconv ~
    N(0, 1 / E) per component, non-zero biases, gamma != 1, beta != 0, pos_embed sigma = 0.5."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    E, L, T, hid, cin = net.embed_dim, net.modes_lat, net.time_dim, net.mlp_hidden, net.in_chans
    nlat, nlon = net.img_shape

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd = {}
    sd["encoder.0.weight"] = rn(E, cin, 1, 1, std=1.0 / math.sqrt(cin))
    sd["encoder.0.bias"] = rn(E, std=0.1)
    sd["encoder.2.weight"] = rn(E, E, 1, 1, std=1.0 / math.sqrt(E))
    if net.use_pos_embed:
        sd["pos_embed"] = rn(1, E, nlat, nlon, std=0.5)
    if net.with_time_emb:
        sd["time_emb_mlp.1.weight"] = rn(T, E, std=1.0 / math.sqrt(E))
        sd["time_emb_mlp.1.bias"] = rn(T, std=0.1)
        sd["time_emb_mlp.3.weight"] = rn(T, T, std=1.0 / math.sqrt(T))
        sd["time_emb_mlp.3.bias"] = rn(T, std=0.1)
    fc2 = "mlp.fwd.3" if net.dropout_mlp > 0.0 else "mlp.fwd.2"          # layers.py:76-80
    for i in range(net.num_layers):
        p = f"blocks.{i}."
        for n in ("norm0", "norm1"):
            sd[p + n + ".weight"] = 1.0 + rn(E, std=0.2)
            sd[p + n + ".bias"] = rn(E, std=0.2)
        if net.with_time_emb:
            sd[p + "time_mlp.1.weight"] = rn(2 * E, T, std=0.5 / math.sqrt(T))
            sd[p + "time_mlp.1.bias"] = rn(2 * E, std=0.1)
        sd[p + "filter.filter.weight"] = rn(E, E, L, 2, std=1.0 / math.sqrt(E))
        sd[p + "filter.filter.bias"] = rn(1, E, 1, 1, std=0.1)
        sd[p + "inner_skip.weight"] = rn(E, E, 1, 1, std=1.0 / math.sqrt(E))
        sd[p + "inner_skip.bias"] = rn(E, std=0.1)
        sd[p + "mlp.fwd.0.weight"] = rn(hid, E, 1, 1, std=1.0 / math.sqrt(E))
        sd[p + "mlp.fwd.0.bias"] = rn(hid, std=0.1)
        sd[p + fc2 + ".weight"] = rn(E, hid, 1, 1, std=1.0 / math.sqrt(hid))
        sd[p + fc2 + ".bias"] = rn(E, std=0.1)
    dec_in = E + (cin if net.big_skip else 0)
    sd["decoder.0.weight"] = rn(E, dec_in, 1, 1, std=1.0 / math.sqrt(dec_in))
    sd["decoder.0.bias"] = rn(E, std=0.1)
    sd["decoder.2.weight"] = rn(net.out_chans, E, 1, 1, std=1.0 / math.sqrt(E))
    return sd


def build_network(n_in: int, n_out: int, n_cond: int, *, nlat: int = 180, nlon: int = 360, embed: int = 256, layers: int = 8,
                  dropout_mlp: float = 0.0, drop_path_rate: float = 0.0, time_range: Optional[Tuple[float, float]] = None,
                  weight_seed: int = 4321, dropout_seed: int = 99, **kw) -> SphericalFourierNeuralOperatorNet:
    """One SFNO on the current device with `trained_like_state_dict(weight_seed)` loaded."""
    net = SphericalFourierNeuralOperatorNet(
        num_input_channels=n_in, num_output_channels=n_out, num_conditional_channels=n_cond, spatial_shape_in=(nlat, nlon),
        embed_dim=embed, num_layers=layers, dropout_mlp=dropout_mlp, drop_path_rate=drop_path_rate,
        with_time_emb=time_range is not None, seed=dropout_seed, **kw)
    net.load_state_dict(trained_like_state_dict(net, seed=weight_seed), strict=True)
    if time_range is not None:
        net.set_min_max_time(*time_range)
    return net


def build_sampler(device, *, state_chans: int = 63, forcing_chans: int = 2, nlat: int = 180, nlon: int = 360, embed: int = 256,
                  layers: int = 8, horizon: int = 6, carried_input_only_channel: bool = False,
                  forecaster_seed: int = 4321, interpolator_seed: int = 4322, dropout_seed: int = 1000,
                  gemm_mode: Optional[str] = None):
    """Forecaster + interpolator + DYffusion sampler of the shipped configuration (`src/configs/diffusion/dyffusion.yaml`,
    `experiment/fv3gfs_interpolation.yaml:18-23`: interpolator dropout 0.1 / drop path 0.1 ON at inference, forecaster
    without).  `carried_input_only_channel`: the published checkpoints' layout, one input-only variable (HGTsfc) in front of
    the state (`hack_for_imprecise_interpolation`, `src/diffusion/dyffusion.py:41-44`)."""
    cs = state_chans + (1 if carried_input_only_channel else 0)
    with torch.cuda.device(device):
        fnet = build_network(cs, state_chans, forcing_chans, nlat=nlat, nlon=nlon, embed=embed, layers=layers,
                             time_range=(0.0, horizon - 1.0), weight_seed=forecaster_seed, gemm_mode=gemm_mode)
        inet = build_network(2 * cs, state_chans, forcing_chans, nlat=nlat, nlon=nlon, embed=embed, layers=layers,
                             dropout_mlp=0.1, drop_path_rate=0.1, time_range=(1.0, horizon - 1.0),
                             weight_seed=interpolator_seed, dropout_seed=dropout_seed, gemm_mode=gemm_mode)
    cfg = dict(hack_for_imprecise_interpolation=True) if carried_input_only_channel else None
    exp = MultiHorizonForecastingDYffusion(fnet, InterpolationExperiment(inet, horizon=horizon), horizon=horizon,
                                           diffusion_config=cfg)
    return exp, fnet, inet


def build_stepper(exp, state_chans: int = 63, forcing_chans: int = 2, carried_input_only_channel: bool = True):
    """`MultiStepStepper` around `exp` for standardised synthetic variables (mean 0, std 1): returns (stepper, all names,
    output names)."""
    out_names = [f"v{i}" for i in range(state_chans)]
    in_names = (["HGTsfc"] if carried_input_only_channel else []) + out_names
    forcing = [f"f{i}" for i in range(forcing_chans)]
    names = in_names + forcing
    stepper = MultiStepStepper(exp, names, out_names, forcing, {n: 0.0 for n in names}, {n: 1.0 for n in names}, None)
    return stepper, names, out_names


def windows(names: List[str], n_windows: int, window: int, nlat: int, nlon: int, n_ics: int = 1,
            seed: int = 1234) -> Iterator[types.SimpleNamespace]:
    """Synthetic standardised series of `n_ics` initial conditions, generated window by window on the host like a data
    loader would deliver them: `.data[name]` is (n_ics, window + 1, nlat, nlon), the first time of a window repeats the last
    time of the previous one (targets only feed the loss terms and the aggregators).  One draw per window for all
    variables (a window of 66 variables is 26 M normals: drawn variable by variable it cost the host more than the GPU
    needs for the window)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    nv = len(names)
    last = torch.randn(nv, n_ics, 1, nlat, nlon, generator=g)
    for _ in range(n_windows):
        buf = torch.cat([last, torch.randn(nv, n_ics, window, nlat, nlon, generator=g)], dim=2)
        last = buf[:, :, -1:]
        yield types.SimpleNamespace(data={n: buf[i] for i, n in enumerate(names)}, times=None)
