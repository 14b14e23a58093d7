"""Per-variable centering / scaling scalars for the stepper (`src/ace_inference/core/normalizer.py`).

The reference keeps them in `data_statistics/{centering,scaling}.nc` and reads them with netCDF4
(`normalizer.py:113-126`, found as `stepper_multistep.py:112-127` describes: `data_dir_stats`, else `data_dir`, else the
repository's own `data_statistics/`).  This package ships the same 55 float32 scalars per file as JSON beside this
module (`data_statistics/*.json`, written by `tools/convert_statistics.py`), so the stepper needs no HDF5 reader; a `.nc`
path is still accepted when netCDF4 or h5py happens to be importable.  The normalisation itself is not done here: the
scalars go into the variable tables of `sdy_norm_pack` / `sdy_step_finish` (`stepper.py`).
"""
from __future__ import annotations

import json
import math
from pathlib import Path
from typing import Dict, Iterable, List, Mapping, Optional, Tuple

PACKAGED_STATISTICS = Path(__file__).parent / "data_statistics"


def load_dict(path, names: Optional[Iterable[str]] = None) -> Dict[str, float]:
    """`load_Dict_from_netcdf` (`normalizer.py:121-126`): {name: scalar} for `names` (all variables of the file when None);
    KeyError for a name the file does not hold, as `ds.variables[c]` raises there."""
    path = Path(path)
    if path.suffix == ".json":
        with open(path) as fh:
            table = json.load(fh)["variables"]
    elif path.suffix == ".nc":
        table = _read_netcdf_scalars(path)
    else:
        raise ValueError(f"{path}: expected a .json (tools/convert_statistics.py) or .nc statistics file")
    if names is None:
        return {k: float(v) for k, v in table.items()}
    return {n: float(table[n]) for n in names}


def _read_netcdf_scalars(path: Path) -> Dict[str, float]:
    try:
        import netCDF4
        ds = netCDF4.Dataset(path)
        ds.set_auto_mask(False)
        out = {k: float(v[:]) for k, v in ds.variables.items()}
        ds.close()
        return out
    except ImportError:
        pass
    try:
        import h5py
        with h5py.File(path, "r") as f:
            return {k: float(f[k][()]) for k in f.keys()}
    except ImportError:
        raise ImportError(f"{path}: neither netCDF4 nor h5py is importable; convert the file once with "
                          "tools/convert_statistics.py and pass the .json (or use the packaged statistics)") from None


def find_statistics(data_dir_stats=None, data_dir=None) -> Tuple[Path, Path]:
    """(centering, scaling) paths in the reference's search order (`stepper_multistep.py:112-127`), a `.json` of the same
    stem preferred over the `.nc` in every directory, the packaged copy standing in for the repository's
    `data_statistics/`."""
    dirs: List[Path] = [Path(d) for d in (data_dir_stats or data_dir,) if d] + [PACKAGED_STATISTICS, Path("/data/climate-model/fv3gfs")]
    for d in dirs:
        for ext in (".json", ".nc"):
            mean, std = d / ("centering" + ext), d / ("scaling" + ext)
            if mean.exists() and std.exists():
                return mean, std
    raise FileNotFoundError(f"Could not find centering and scaling files in {[str(d) for d in dirs]}")


class StandardNormalizer:
    """Holder of the scalars under the reference's names (`normalizer.py:57-95`): `means`, `stds`, `get_state`,
    `from_state`.  `normalize` / `denormalize` are for host-side checks on small tensors only; the product path applies the
    same scalars inside the pack / finish kernels."""

    def __init__(self, means: Mapping[str, float], stds: Mapping[str, float]):
        self.means = {k: float(v) for k, v in means.items()}
        self.stds = {k: float(v) for k, v in stds.items()}

    def normalize(self, tensors):
        return {k: (t - self.means[k]) / self.stds[k] if k in self.means else t for k, t in tensors.items()}

    def denormalize(self, tensors):
        return {k: t * self.stds[k] + self.means[k] if k in self.means else t for k, t in tensors.items()}

    def get_state(self):
        return {"means": dict(self.means), "stds": dict(self.stds)}

    @classmethod
    def from_state(cls, state) -> "StandardNormalizer":
        return cls(state["means"], state["stds"])


def get_normalizer(global_means_path, global_stds_path, names: List[str]) -> StandardNormalizer:
    """`normalizer.py:113-118`.  A NaN or zero scaling for a requested name is refused here (the reference would carry it
    into every normalised field: `soil_moisture` is NaN in the shipped files and is in no shipped variable list)."""
    means, stds = load_dict(global_means_path, names), load_dict(global_stds_path, names)
    bad = [n for n in names if not (math.isfinite(means[n]) and math.isfinite(stds[n]) and stds[n] != 0.0)]
    if bad:
        raise ValueError(f"non-finite centering / zero or non-finite scaling for {bad}")
    return StandardNormalizer(means, stds)
