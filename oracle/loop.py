"""Oracle: restatement of the reference's window driver (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `run_inference`, `_inference_internal_loop` and `WindowStitcher`
(`src/ace_inference/inference/loop.py:26-117,120-264`) with `SteppedData.remove_initial_condition`
(`src/ace_inference/core/stepper.py:186-200`): windows of `forward_steps_in_memory` steps, members looped SERIALLY, the
last generated state carried into the next window per member, members stacked on a leading axis, the initial time of
every window but the first dropped before it reaches the writer / aggregator.  Derived variables
(`compute_derived_quantities`) are outside the path and not restated.
Pinned by `tests/golden/fx_loop_tiny.npz` (produced by the reference's own `run_inference`).
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch


def run_inference(windows: List[Dict[str, torch.Tensor]], run_on_batch: Callable, n_forward_steps: int,
                  forward_steps_in_memory: int, n_members: int):
    """`windows[i]`: name -> (n_sample, forward_steps_in_memory + 1, H, W); `run_on_batch(data, member)` ->
    (metrics dict, gen_data dict, gen_data_norm dict) for one member (oracle.stepper.run_on_batch signature tail).
    Returns (writer_calls, aggregator_calls): [(start_timestep, prediction dict)], [(loss, i_time_start)]."""
    is_ens = n_members > 1
    ic = None
    ens_keys: List[str] = []
    st_time = 0                                   # WindowStitcher.i_time
    writer_calls, agg_calls = [], []
    for i, window in enumerate(windows):
        i_time = i * forward_steps_in_memory
        data = {k: v.clone() for k, v in window.items()}
        gens, gens_norm, losses = [], [], []
        for m in range(n_members):
            if ic is not None:                    # apply_initial_condition (loop.py:85-117)
                for k, v in data.items():
                    c = ic[k]
                    if is_ens and k in ens_keys:
                        c = c[m]
                    v[:, 0] = c
            metrics, gen, gen_norm = run_on_batch(data, m)
            gens.append(gen)
            gens_norm.append(gen_norm)
            losses.append(metrics["loss"])
        if is_ens:                                # loop.py:218-237
            gen = {k: torch.stack([g[k] for g in gens], 0) for k in gens[0]}
            loss = sum(losses) / len(losses)
        else:
            gen, loss = gens[0], losses[0]
        target = data
        if i_time > 0:                            # _inference_internal_loop (loop.py:133-139)
            gen = {k: (v[:, :, 1:] if is_ens else v[:, 1:]) for k, v in gen.items()}
            target = {k: v[:, 1:] for k, v in target.items()}
            i_time_agg = i_time + 1
        else:
            i_time_agg = i_time
        writer_calls.append((st_time, gen))       # WindowStitcher.append (loop.py:50-83)
        st_time += next(iter(target.values())).shape[1]
        if st_time < n_forward_steps:
            ic = {k: v[:, -1].clone() for k, v in target.items()}
            ens_keys = list(gen.keys())
            for k, v in gen.items():
                ic[k] = v[..., -1, :, :].clone()
        agg_calls.append((float(loss), i_time_agg))
    return writer_calls, agg_calls
