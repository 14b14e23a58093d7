"""Shared builders for the parity tests: a product network and an oracle network with the same weights."""
import torch

from oracle.philox import drop_path_keep, element_keep_mask, element_keep_mask_torch
from oracle.sfno import OracleSFNO, SFNOConfig, make_state_dict


def make_pair(cfg: SFNOConfig, n_in: int, n_cond: int, seed: int = 4321, net_seed: int = 99):
    """(product net on cuda, oracle net) sharing a 'trained-like' state_dict."""
    import sdy_amd

    assert n_in + n_cond == cfg.in_chans
    sd = make_state_dict(cfg, seed=seed)
    net = sdy_amd.SphericalFourierNeuralOperatorNet(
        num_input_channels=n_in, num_output_channels=cfg.out_chans, num_conditional_channels=n_cond,
        spatial_shape_in=(cfg.nlat, cfg.nlon), embed_dim=cfg.embed_dim, num_layers=cfg.num_layers,
        mlp_ratio=cfg.mlp_ratio, dropout_mlp=cfg.dropout_mlp, drop_path_rate=cfg.drop_path_rate,
        with_time_emb=cfg.with_time_emb, data_grid=cfg.data_grid, big_skip=cfg.big_skip, pos_embed=cfg.pos_embed,
        seed=net_seed, time_dim_mult=cfg.time_dim_mult,
    )
    net.load_state_dict(sd, strict=True)
    if cfg.with_time_emb:
        net.set_min_max_time(cfg.min_time, cfg.max_time)
    return net, OracleSFNO(cfg, sd), sd


class PhiloxMasks:
    """Oracle-side mask provider reproducing the device dropout stream (include/sdy_amd.h, 'Dropout stream')."""

    def __init__(self, cfg: SFNOConfig, seed: int, batch_offset: int = 0):
        self.cfg, self.seed, self.batch_offset = cfg, seed, batch_offset
        self.call = 0
        self.rows = None      # optional: global trajectory index of every batch row (overrides batch_offset + b)
        self.device = None    # "cuda": evaluate the (torch restatement of the) stream on the GPU -- full-size, full-depth tests

    def __call__(self, kind, layer, shape):
        c = self.cfg
        if kind == "drop_path":
            keep = drop_path_keep(self.seed, self.call, layer, c.drop_path_rates[layer], shape[0], self.batch_offset,
                                  self.rows)
            return torch.from_numpy(keep).reshape(-1, 1, 1, 1)
        B, C, H, W = shape
        k = 0 if kind == "mlp_hidden" else 1
        if self.device is not None:
            return element_keep_mask_torch(self.seed, self.call, layer, k, c.dropout_mlp, B, C, H, W, self.batch_offset,
                                           self.rows, device=self.device)
        return torch.from_numpy(element_keep_mask(self.seed, self.call, layer, k, c.dropout_mlp, B, C, H, W,
                                                  self.batch_offset, self.rows))
