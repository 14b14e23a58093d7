"""Loading helpers for tests/golden/*.npz (vectors produced by tools/gen_golden.py from the real reference)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def state_dict(z, prefix="sd::"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def cfg_from(z, key="cfg"):
    from oracle.sfno import SFNOConfig

    d = json.loads(str(z[key]))
    n_in, n_cond = d.pop("n_in", None), d.pop("n_cond", None)
    return SFNOConfig(**d), n_in, n_cond


def recorded_masks(z):
    """-> list of (module_name, bool tensor) in the order the reference drew them."""
    names = json.loads(str(z["mask_names"]))
    out = []
    for i, n in enumerate(names):
        shape = tuple(int(s) for s in z[f"mask{i}_shape"])
        bits = np.unpackbits(z[f"mask{i}"])[: int(np.prod(shape))]
        out.append((n, torch.from_numpy(bits.astype(np.float32)).reshape(shape)))
    return out


def masks_per_forward(records, cfg):
    """Split the flat record list into per-forward dicts {(kind, layer): mask}; every forward of a network with dropout
    draws, per block: hidden mask, output mask, then (blocks with a DropPath) the per-sample keep flags."""
    hid = int(cfg.embed_dim * cfg.mlp_ratio)
    per_fwd, cur, seen_h = [], {}, {}
    for name, m in records:
        layer = int(name.split(".")[1])
        if name.endswith("drop_path"):
            key = ("drop_path", layer)
        elif m.shape[1] == hid and ("mlp_hidden", layer) not in cur:
            key = ("mlp_hidden", layer)
        else:
            key = ("mlp_out", layer)
        if key in cur:          # next forward starts
            per_fwd.append(cur)
            cur = {}
            key = ("mlp_hidden", layer) if not name.endswith("drop_path") else key
        cur[key] = m
    if cur:
        per_fwd.append(cur)
    return per_fwd


def mask_fn_from(d):
    def fn(kind, layer, shape):
        m = d.get((kind, layer))
        if m is None:
            return None
        return m.reshape(-1, 1, 1, 1) if kind == "drop_path" else m
    return fn


def seeded_case(z):
    """Fixtures that store SEEDS instead of weights / inputs (the full-width ones: fx_sfno_full, fx_sfno_wide_masks):
    -> (cfg, n_in, n_cond, state_dict, x, cond, time), rebuilt with the generators tools/gen_golden.py used and checked
    against the checksums the generating run stored, so a drift of torch's CPU generator cannot pass as a parity failure."""
    import numpy as np

    from oracle.sfno import make_state_dict

    cfg, n_in, n_cond = cfg_from(z)
    sd = make_state_dict(cfg, seed=int(z["seed_w"]))
    dig = np.array([float(sum(v.double().abs().sum() for v in sd.values())),
                    float(sum((v.double() ** 2).sum() for v in sd.values()))])
    assert np.allclose(dig, z["weights_digest"], rtol=1e-9), "make_state_dict(seed) no longer reproduces the fixture's weights"
    B = z["y"].shape[0]
    g = torch.Generator(device="cpu").manual_seed(int(z["seed_x"]))
    x = torch.randn(B, n_in, cfg.nlat, cfg.nlon, generator=g)
    cond = torch.randn(B, n_cond, cfg.nlat, cfg.nlon, generator=g)
    idig = np.array([float(x.double().abs().sum()), float(cond.double().abs().sum())])
    assert np.allclose(idig, z["inputs_digest"], rtol=1e-9), "the seeded inputs no longer reproduce the fixture's"
    return cfg, n_in, n_cond, sd, x, cond, torch.from_numpy(z["time"])
