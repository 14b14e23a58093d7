"""Oracle: PyTorch-CPU restatement of the reference SFNO forward pass.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, op for op (same torch primitives as the reference, fp32):
  * `SphericalFourierNeuralOperatorNet.forward / forward_features`
    (`src/models/sfno/sfnonet.py:775-841`)
  * `FourierNeuralOperatorBlock.forward / time_scale_shift`
    (`src/models/sfno/sfnonet.py:280-337`)
  * `SpectralConvS2.forward` (`src/models/sfno/s2convolutions.py:158-193`)
    with `_contract_dhconv` (`src/models/sfno/contractions.py:159-169`)
  * `MLP` (`src/models/sfno/layers.py:53-93`), `drop_path`
    (`src/models/modules/drop_path.py:5-22`)
  * `SinusoidalPosEmb`, `get_time_embedder` (`src/models/modules/misc.py:21-33,132-148`)
  * `BaseModel.concat_condition_if_needed` (`src/models/_base_model.py:166-192`)

Parameters are taken from a state_dict with the reference's key names
(SURVEY.md Appendix B), so golden fixtures generated from the real reference
load unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .sht import InverseRealSHT, RealSHT


@dataclass
class SFNOConfig:
    in_chans: int              # input channels incl. conditional channels
    out_chans: int
    nlat: int = 180
    nlon: int = 360
    embed_dim: int = 256
    num_layers: int = 8
    mlp_ratio: float = 2.0
    dropout_mlp: float = 0.0
    drop_path_rate: float = 0.0
    with_time_emb: bool = True
    time_dim_mult: int = 2
    data_grid: str = "equiangular"
    scale_factor: int = 1
    hard_thresholding_fraction: float = 1.0
    big_skip: bool = True
    pos_embed: bool = True
    min_time: Optional[float] = None
    max_time: Optional[float] = None

    @property
    def h(self):
        return self.nlat // self.scale_factor

    @property
    def w(self):
        return self.nlon // self.scale_factor

    @property
    def modes_lat(self):
        return int(self.h * self.hard_thresholding_fraction)

    @property
    def modes_lon(self):
        return int((self.w // 2 + 1) * self.hard_thresholding_fraction)

    @property
    def drop_path_rates(self):
        # sfnonet.py:622
        return [x.item() for x in torch.linspace(0, self.drop_path_rate, self.num_layers)]

    @property
    def mlp_fc2_key(self):
        # layers.py:76-80: with dropout the Sequential is (fc1, act, drop, fc2, drop)
        return "mlp.fwd.3" if self.dropout_mlp > 0.0 else "mlp.fwd.2"


# A mask provider returns a keep-mask (float 0/1, broadcastable to `shape`) or None (= no dropout).
MaskFn = Callable[[str, int, tuple], Optional[torch.Tensor]]


def sinusoidal_pos_emb(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=t.dtype, device=t.device) * -emb)
    emb = t[:, None] * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


class OracleSFNO:
    """Functional restatement; `sd` maps reference state_dict names to CPU fp32 tensors."""

    def __init__(self, cfg: SFNOConfig, sd: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float32, device="cpu"):
        """`dtype=torch.float64`: the same op sequence in double precision -- the yardstick that tells how much of a
        difference between two fp32 implementations is either one's rounding (tools/chain_error_probe.py); the reference,
        and every parity test, run fp32.
        `device`: where torch evaluates the op sequence.  "cpu" is the oracle of every parity test; "cuda" runs the SAME torch
        ops on the GPU (as the reference itself does in production) -- only the float64 yardstick of the chain-error test uses
        it, where sixteen chained full-depth forwards in double precision take minutes on the host cores."""
        self.cfg = cfg
        self.dtype = dtype
        self.device = torch.device(device)
        self.sd = {k: v.detach().to(device=self.device, dtype=dtype) for k, v in sd.items()}
        c = cfg
        kw = dict(lmax=c.modes_lat, mmax=c.modes_lon)
        self.trans_down = RealSHT(c.nlat, c.nlon, grid=c.data_grid, **kw).to(dtype)
        self.itrans_up = InverseRealSHT(c.nlat, c.nlon, grid=c.data_grid, **kw).to(dtype)
        self.trans = RealSHT(c.h, c.w, grid="legendre-gauss", **kw).to(dtype)
        self.itrans = InverseRealSHT(c.h, c.w, grid="legendre-gauss", **kw).to(dtype)
        for m in (self.trans_down, self.itrans_up, self.trans, self.itrans):
            m.to(self.device)

    # ---- pieces ---------------------------------------------------------------------------
    def time_repr(self, time: torch.Tensor) -> torch.Tensor:
        sd, c = self.sd, self.cfg
        if c.min_time is not None:
            assert (c.min_time <= time).all() and (time <= c.max_time).all(), f"time out of range: {time}"
        e = sinusoidal_pos_emb(time.to(device=self.device, dtype=self.dtype), c.embed_dim)
        h = F.linear(e, sd["time_emb_mlp.1.weight"], sd["time_emb_mlp.1.bias"])
        h = F.gelu(h)
        return F.linear(h, sd["time_emb_mlp.3.weight"], sd["time_emb_mlp.3.bias"])

    def spectral_conv(self, i: int, x: torch.Tensor):
        """SpectralConvS2.forward: returns (y, residual)."""
        c, sd = self.cfg, self.sd
        fwd = self.trans_down if i == 0 else self.trans
        inv = self.itrans_up if i == c.num_layers - 1 else self.itrans
        scale_residual = (fwd.nlat != inv.nlat) or (fwd.nlon != inv.nlon) or (fwd.grid != inv.grid)
        residual = x
        xs = fwd(x)
        if scale_residual:
            residual = inv(xs.contiguous())
        w = torch.view_as_complex(sd[f"blocks.{i}.filter.filter.weight"].contiguous())  # (in, out, l)
        xp = torch.zeros_like(xs)
        ml, mm = inv.lmax, inv.mmax
        xp[..., :ml, :mm] = torch.einsum("bixy,iox->boxy", xs[..., :ml, :mm], w)
        y = inv(xp.contiguous())
        y = y + sd[f"blocks.{i}.filter.filter.bias"]
        return y, residual

    def block(self, i: int, x: torch.Tensor, t_repr: Optional[torch.Tensor], mask_fn: Optional[MaskFn]):
        c, sd = self.cfg, self.sd
        p = f"blocks.{i}."
        xn = F.instance_norm(x, weight=sd[p + "norm0.weight"], bias=sd[p + "norm0.bias"], eps=1e-6)
        if t_repr is not None:
            te = F.linear(F.silu(t_repr), sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
            scale, shift = te[:, :, None, None].chunk(2, dim=1)
            xn = xn * (scale + 1) + shift
        y, residual = self.spectral_conv(i, xn)
        y = y + F.conv2d(residual, sd[p + "inner_skip.weight"], sd[p + "inner_skip.bias"])
        y = F.gelu(y)
        y = F.instance_norm(y, weight=sd[p + "norm1.weight"], bias=sd[p + "norm1.bias"], eps=1e-6)
        # MLP
        h = F.conv2d(y, sd[p + "mlp.fwd.0.weight"], sd[p + "mlp.fwd.0.bias"])
        h = F.gelu(h)
        pm = c.dropout_mlp
        if mask_fn is not None and pm > 0.0:
            m = mask_fn("mlp_hidden", i, tuple(h.shape))
            if m is not None:
                h = h * m.to(h.device) * (1.0 / (1.0 - pm))
        k2 = p + c.mlp_fc2_key
        h = F.conv2d(h, sd[k2 + ".weight"], sd[k2 + ".bias"])
        if mask_fn is not None and pm > 0.0:
            m = mask_fn("mlp_out", i, tuple(h.shape))
            if m is not None:
                h = h * m.to(h.device) * (1.0 / (1.0 - pm))
        dp = c.drop_path_rates[i]
        if mask_fn is not None and dp > 0.0:
            m = mask_fn("drop_path", i, (h.shape[0], 1, 1, 1))
            if m is not None:
                h = h.div(1.0 - dp) * m.to(h.device)
        return h + residual

    # ---- full network ---------------------------------------------------------------------
    def forward(self, inputs, time=None, condition=None, static_condition=None, mask_fn: Optional[MaskFn] = None):
        c, sd = self.cfg, self.sd
        parts = [t.to(device=self.device, dtype=self.dtype) for t in (inputs, condition, static_condition) if t is not None]
        x = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
        assert x.shape[1] == c.in_chans, f"expected {c.in_chans} channels, got {x.shape[1]}"
        residual = x
        x = F.conv2d(x, sd["encoder.0.weight"], sd["encoder.0.bias"])
        x = F.gelu(x)
        x = F.conv2d(x, sd["encoder.2.weight"])
        if c.pos_embed:
            x = x + sd["pos_embed"]
        t_repr = self.time_repr(time) if c.with_time_emb else None
        for i in range(c.num_layers):
            x = self.block(i, x, t_repr, mask_fn)
        if c.big_skip:
            x = torch.cat((x, residual), dim=1)
        x = F.conv2d(x, sd["decoder.0.weight"], sd["decoder.0.bias"])
        x = F.gelu(x)
        return F.conv2d(x, sd["decoder.2.weight"])

    __call__ = forward


# ---- synthetic "trained-like" weights ---------------------------------------
def make_state_dict(cfg: SFNOConfig, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """Random weights with trained-like magnitudes so every branch is numerically visible.

    The reference initialises dhconv weights at scale 1/E^2 (`s2convolutions.py:70-71,146`) which makes
    the spectral branch invisible next to the skip path; here dhconv ~ N(0, 1/E) per component,
    non-zero biases, gamma != 1, beta != 0, pos_embed sigma = 0.5.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    E, L = cfg.embed_dim, cfg.modes_lat
    T = E * cfg.time_dim_mult
    hid = int(E * cfg.mlp_ratio)

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd = {}
    sd["encoder.0.weight"] = rn(E, cfg.in_chans, 1, 1, std=1.0 / math.sqrt(cfg.in_chans))
    sd["encoder.0.bias"] = rn(E, std=0.1)
    sd["encoder.2.weight"] = rn(E, E, 1, 1, std=1.0 / math.sqrt(E))
    if cfg.pos_embed:
        sd["pos_embed"] = rn(1, E, cfg.nlat, cfg.nlon, std=0.5)
    if cfg.with_time_emb:
        sd["time_emb_mlp.1.weight"] = rn(T, E, std=1.0 / math.sqrt(E))
        sd["time_emb_mlp.1.bias"] = rn(T, std=0.1)
        sd["time_emb_mlp.3.weight"] = rn(T, T, std=1.0 / math.sqrt(T))
        sd["time_emb_mlp.3.bias"] = rn(T, std=0.1)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        for n in ("norm0", "norm1"):
            sd[p + n + ".weight"] = 1.0 + rn(E, std=0.2)
            sd[p + n + ".bias"] = rn(E, std=0.2)
        if cfg.with_time_emb:
            sd[p + "time_mlp.1.weight"] = rn(2 * E, T, std=0.5 / math.sqrt(T))
            sd[p + "time_mlp.1.bias"] = rn(2 * E, std=0.1)
        sd[p + "filter.filter.weight"] = rn(E, E, L, 2, std=1.0 / math.sqrt(E))
        sd[p + "filter.filter.bias"] = rn(1, E, 1, 1, std=0.1)
        sd[p + "inner_skip.weight"] = rn(E, E, 1, 1, std=1.0 / math.sqrt(E))
        sd[p + "inner_skip.bias"] = rn(E, std=0.1)
        sd[p + "mlp.fwd.0.weight"] = rn(hid, E, 1, 1, std=1.0 / math.sqrt(E))
        sd[p + "mlp.fwd.0.bias"] = rn(hid, std=0.1)
        sd[p + cfg.mlp_fc2_key + ".weight"] = rn(E, hid, 1, 1, std=1.0 / math.sqrt(hid))
        sd[p + cfg.mlp_fc2_key + ".bias"] = rn(E, std=0.1)
    dec_in = E + (cfg.in_chans if cfg.big_skip else 0)
    sd["decoder.0.weight"] = rn(E, dec_in, 1, 1, std=1.0 / math.sqrt(dec_in))
    sd["decoder.0.bias"] = rn(E, std=0.1)
    sd["decoder.2.weight"] = rn(cfg.out_chans, E, 1, 1, std=1.0 / math.sqrt(E))
    return sd
