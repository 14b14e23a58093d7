#!/usr/bin/env python3
"""`data_statistics/{centering,scaling}.nc` of the reference -> `spherical-dyffusion_amd/data_statistics/*.json`.

The reference reads the per-variable means and standard deviations with netCDF4 (`src/ace_inference/core/normalizer.py:121-126`,
called from `src/ace_inference/core/stepper_multistep.py:112-131`); neither netCDF4 nor h5py travels with this package, so
the 55 scalars of each file are shipped as JSON (float32 values printed with enough digits to round-trip; NaN stays NaN:
`soil_moisture` has none in the source).  Runs in the build container only: the files are netCDF-4 = HDF5 and are read
here through the C API of a libhdf5 found on the system (ctypes; nothing is installed, nothing is linked into the product).

    python tools/convert_statistics.py [/root/reference/data_statistics]
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import json
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spherical-dyffusion_amd", "data_statistics")


def _libhdf5():
    cands = [ctypes.util.find_library("hdf5")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(
        glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for c in cands:
        if c:
            try:
                return C.CDLL(c)
            except OSError:
                pass
    raise SystemExit("no libhdf5 on this system: convert the statistics where one exists (or with netCDF4 / h5py)")


def read_scalars(path: str) -> dict:
    h = _libhdf5()
    h.H5open()
    hid = C.c_int64
    h.H5Fopen.restype = hid; h.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    h.H5Dopen2.restype = hid; h.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    h.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    h.H5Dget_space.restype = hid; h.H5Dget_space.argtypes = [hid]
    h.H5Sget_simple_extent_npoints.restype = C.c_int64; h.H5Sget_simple_extent_npoints.argtypes = [hid]
    h.H5Gget_num_objs.argtypes = [hid, C.POINTER(C.c_uint64)]
    h.H5Gget_objname_by_idx.restype = C.c_ssize_t
    h.H5Gget_objname_by_idx.argtypes = [hid, C.c_uint64, C.c_char_p, C.c_size_t]
    for fn in ("H5Dclose", "H5Sclose", "H5Fclose"):
        getattr(h, fn).argtypes = [hid]
    f32 = hid.in_dll(h, "H5T_NATIVE_FLOAT_g").value
    f = h.H5Fopen(path.encode(), 0, 0)
    if f < 0:
        raise SystemExit(f"cannot open {path}")
    n = C.c_uint64()
    h.H5Gget_num_objs(f, C.byref(n))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(256)
        h.H5Gget_objname_by_idx(f, i, name, 256)
        d = h.H5Dopen2(f, name.value, 0)
        sp = h.H5Dget_space(d)
        if h.H5Sget_simple_extent_npoints(sp) != 1:
            raise SystemExit(f"{path}:{name.value.decode()} is not a scalar")
        v = C.c_float()
        if h.H5Dread(d, f32, 0, 0, 0, C.byref(v)) < 0:
            raise SystemExit(f"cannot read {name.value.decode()}")
        out[name.value.decode()] = float(np.float32(v.value))
        h.H5Sclose(sp); h.H5Dclose(d)
    h.H5Fclose(f)
    return out


def main(src: str) -> None:
    os.makedirs(OUT, exist_ok=True)
    for stem in ("centering", "scaling"):
        vals = read_scalars(os.path.join(src, stem + ".nc"))
        with open(os.path.join(OUT, stem + ".json"), "w") as fh:
            json.dump({"source": f"data_statistics/{stem}.nc", "dtype": "float32", "variables": vals}, fh, indent=1, sort_keys=True)
            fh.write("\n")
        print(stem, len(vals), "variables")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data_statistics")
