"""Oracle: restatement of the reference's autoregressive stepper (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `run_on_batch_multistep` (`src/ace_inference/core/stepper_multistep.py:298-466`) with
`StandardNormalizer` (`src/ace_inference/core/normalizer.py:96-110`), `Packer` (`src/utilities/packer.py:70-77`),
`Prescriber.__call__` (`src/ace_inference/core/prescriber.py:68-92`) and the relative `LpLoss`
(`src/ace_inference/training/utils/darcy_loss.py:214-228`).  `module` is any object with `true_horizon`,
`get_preds_at_t_for_batch(...)` and the `ema_scope` / `inference_dropout_scope` context managers.
Pinned by `tests/golden/fx_stepper_tiny.npz` (produced by the reference's own function).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


def lp_rel(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    n = x.shape[0]
    d = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), 2, 1)
    yn = torch.norm(y.reshape(n, -1), 2, 1)
    return torch.mean(d / yn)


def run_on_batch(data: Dict[str, torch.Tensor], module, in_names: List[str], out_names: List[str],
                 forcing_names: List[str], means: Dict[str, torch.Tensor], stds: Dict[str, torch.Tensor],
                 n_forward_steps: int, prescriber: Optional[dict] = None, hack: bool = False):
    """-> (metrics, gen_data, gen_data_norm); `in_names` excludes the forcings (in_packer names)."""
    norm = {k: (v - means[k]) / stds[k] if k in means else v for k, v in data.items()}
    pack = lambda d, names, t: torch.stack([d[n][:, t] for n in names], dim=-3)  # noqa: E731
    horizon = module.true_horizon
    inp = {n: norm[n][:, 0] for n in in_names}
    forc = {n: norm[n][:, 0] for n in forcing_names}
    gen_steps, metrics, loss = [], {}, 0.0
    for th in range(1, n_forward_steps + 1):
        h = th % horizon or horizon
        batch = {"dynamics": torch.stack([inp[n] for n in in_names], dim=-3)}
        if hack:
            batch["static_condition"] = torch.stack([forc[n] for n in forcing_names], dim=-3)
        with module.ema_scope(), module.inference_dropout_scope():
            res = module.get_preds_at_t_for_batch(batch, horizon=h, split="predict", ensemble=False,
                                                  is_autoregressive=th > horizon, prepare_inputs=False,
                                                  num_predictions=1)
        g = res[f"t{h}_preds_normed"]
        tgt = pack(norm, out_names, th)
        step_loss = lp_rel(g, tgt)
        loss = loss + step_loss
        metrics[f"loss_step_{th - 1}"] = float(step_loss)
        gen = {n: g.select(-3, i) for i, n in enumerate(out_names)}
        if prescriber is not None:
            p = prescriber["prescribed_name"]
            mask = data[prescriber["mask_name"]][:, th]
            tn = norm[p][:, th]
            if prescriber.get("interpolate", False):
                gen[p] = mask * tn + (1 - mask) * gen[p]
            else:
                gen[p] = torch.where(torch.round(mask).to(int) == prescriber["mask_value"], tn, gen[p])
        gen_steps.append(gen)
        ar = dict(gen)
        if "preds_autoregressive_init_normed" in res:     # stepper_multistep.py:412-418: a separate state seeds the next window
            a_t = res["preds_autoregressive_init_normed"]
            ar = {n: a_t.select(-3, i) for i, n in enumerate(out_names)}
            if prescriber is not None:
                p = prescriber["prescribed_name"]
                if prescriber.get("interpolate", False):
                    ar[p] = mask * tn + (1 - mask) * ar[p]
                else:
                    ar[p] = torch.where(torch.round(mask).to(int) == prescriber["mask_value"], tn, ar[p])
        forc = {n: norm[n][:, th] for n in forcing_names}
        if hack:
            ar["HGTsfc"] = inp["HGTsfc"]
        inp = ar
    metrics["loss"] = float(loss)
    initial = {n: norm[n][:, 0] for n in out_names}
    gen_norm = {n: torch.stack([s[n] for s in [initial] + gen_steps], dim=1) for n in out_names}
    gen = {n: v * stds[n] + means[n] if n in means else v for n, v in gen_norm.items()}
    return metrics, gen, gen_norm
