"""`run_inference(module, datamodule)` of the reference's Python interface (`src/interface.py:302-313`), minus Lightning.

The reference hands the module to `pytorch_lightning.Trainer.predict`, which calls `module.predict_step(batch, batch_idx)`
for every batch of `datamodule.predict_dataloader()` and `on_predict_epoch_end` at the end; `run_inference` then merges the
per-batch results with `module._evaluation_get_preds(results, split="predict")` and, when the datamodule offers it, turns
them into an xarray dataset.  The trainer is plumbing: this driver makes the same calls in the same order, on the current
device (one process per GPU; shard the dataloader per rank as the reference's DistributedSampler would).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch


def _to_device(x: Any, device):
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    if isinstance(x, dict):
        return {k: _to_device(v, device) for k, v in x.items()}
    return x


def run_inference(module, datamodule, trainer=None, trainer_kwargs: Optional[Dict[str, Any]] = None, device=None):
    """`src/interface.py:302-313`.  `trainer` / `trainer_kwargs` are accepted for call compatibility and ignored (there is
    no Lightning on this path); `datamodule.predict_dataloader()` yields dictionaries with `dynamics` (and conditions)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("sdy_amd.interface.run_inference runs on the GPU only (no CPU fallback)")
    if hasattr(datamodule, "setup"):
        datamodule.setup("predict")
    results = []
    with torch.cuda.device(dev):
        for batch_idx, batch in enumerate(datamodule.predict_dataloader()):
            results.append(module.predict_step(_to_device(batch, dev), batch_idx))
        module.on_predict_epoch_end()
    results = module._evaluation_get_preds(results, split="predict")
    if hasattr(datamodule, "numpy_results_to_xr_dataset"):
        results = datamodule.numpy_results_to_xr_dataset(results, split="predict")
    return results
