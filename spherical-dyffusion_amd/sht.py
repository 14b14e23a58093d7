"""Host mirror of `torch_harmonics.RealSHT` / `InverseRealSHT` on top of the HIP library.

Same constructor signature and attributes the reference relies on
(`src/models/sfno/sfnonet.py:551-554`; attributes read at `src/models/sfno/s2convolutions.py:73-83,106-115`),
same call convention: real (..., nlat, nlon) -> complex64 (..., lmax, mmax) and back.  All arithmetic runs in
`libsdy_amd.so` (longitude FFT kernel + fp32-MFMA Legendre GEMM); torch only owns the buffers.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, current_stream, lib, ptr


class ShtPlan:
    """Owns one native plan (device Legendre tables + FFT twiddles) for (nlat, nlon, lmax, mmax, grid)."""

    _cache = {}

    def __init__(self, nlat, nlon, lmax, mmax, grid, gemm_mode="h3"):
        if grid not in _lib.SDY_GRID:
            raise ValueError(f"unsupported grid {grid!r} (supported: {sorted(_lib.SDY_GRID)})")
        h = C.c_void_p()
        check(lib.sdy_sht_plan_create_ex(nlat, nlon, lmax, mmax, _lib.SDY_GRID[grid], 1 if gemm_mode == "h3" else 0,
                                         C.byref(h)), "sdy_sht_plan_create_ex")
        self.handle = h
        dims = (C.c_int * 6)()
        check(lib.sdy_sht_plan_dims(h, C.byref(dims)))
        self.nlat, self.nlon, self.lmax, self.mmax, self.mtr = dims[0], dims[1], dims[2], dims[3], dims[4]
        self.grid = grid

    @classmethod
    def get(cls, nlat, nlon, lmax, mmax, grid, device_index, gemm_mode=None):
        gemm_mode = gemm_mode or _lib.default_gemm_mode()
        key = (nlat, nlon, lmax, mmax, grid, device_index, gemm_mode)
        if key not in cls._cache:
            with torch.cuda.device(device_index):
                cls._cache[key] = cls(nlat, nlon, lmax, mmax, grid, gemm_mode)
        return cls._cache[key]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.sdy_sht_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _ShtBase(torch.nn.Module):
    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="equiangular", norm="ortho", csphase=True,
                 gemm_mode=None):
        super().__init__()
        self.gemm_mode = gemm_mode   # None -> $SDY_GEMM_MODE or "h3" (split-fp16 MFMA); "f32" = fp32 MFMA
        if norm != "ortho" or not csphase:
            raise NotImplementedError("only norm='ortho', csphase=True (what the reference uses) is implemented")
        self.nlat, self.nlon, self.grid = nlat, nlon, grid
        self.lmax = lmax or nlat
        self.mmax = mmax or nlon // 2 + 1
        self.norm, self.csphase = norm, csphase
        self._plans = {}

    def float(self):  # the reference calls `.float()` on the transforms (sfnonet.py:551-554); tables are fp32 already
        return self

    def _plan(self, device) -> ShtPlan:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return ShtPlan.get(self.nlat, self.nlon, self.lmax, self.mmax, self.grid, idx, self.gemm_mode)

    @staticmethod
    def _require_gpu(x):
        if not x.is_cuda:
            raise RuntimeError("sdy_amd transforms run on the GPU only (no CPU fallback); got a CPU tensor")


class RealSHT(_ShtBase):
    """x (..., nlat, nlon) float32 -> (..., lmax, mmax) complex64."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._require_gpu(x)
        assert x.shape[-2] == self.nlat and x.shape[-1] == self.nlon, f"bad grid {tuple(x.shape[-2:])}"
        lead = x.shape[:-2]
        xf = x.reshape(-1, self.nlat, self.nlon).to(torch.float32).contiguous()
        n = xf.shape[0]
        pad = (-n) % 4  # channel count seen by the kernels must be a multiple of 4
        if pad:
            xf = torch.cat([xf, xf.new_zeros(pad, self.nlat, self.nlon)], 0)
        C_ = xf.shape[0]
        plan = self._plan(x.device)
        out = torch.empty(C_, self.lmax, self.mmax, 2, dtype=torch.float32, device=x.device)
        nws = lib.sdy_sht_workspace_floats(plan.handle, 1, C_)
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.sdy_sht_forward(plan.handle, ptr(xf), ptr(out), 1, C_, ptr(ws), nws, current_stream()),
                  "sdy_sht_forward")
        out = torch.view_as_complex(out[:n])
        return out.reshape(*lead, self.lmax, self.mmax)


class InverseRealSHT(_ShtBase):
    """c (..., lmax, mmax) complex64 -> (..., nlat, nlon) float32."""

    def forward(self, c: torch.Tensor) -> torch.Tensor:
        self._require_gpu(c)
        assert c.shape[-2] == self.lmax and c.shape[-1] == self.mmax, f"bad spectrum {tuple(c.shape[-2:])}"
        lead = c.shape[:-2]
        cf = torch.view_as_real(c.to(torch.complex64).contiguous()).reshape(-1, self.lmax, self.mmax, 2).contiguous()
        n = cf.shape[0]
        pad = (-n) % 4
        if pad:
            cf = torch.cat([cf, cf.new_zeros(pad, self.lmax, self.mmax, 2)], 0)
        C_ = cf.shape[0]
        plan = self._plan(c.device)
        out = torch.empty(C_, self.nlat, self.nlon, dtype=torch.float32, device=c.device)
        nws = lib.sdy_sht_workspace_floats(plan.handle, 1, C_)
        ws = torch.empty(nws, dtype=torch.float32, device=c.device)
        with torch.cuda.device(c.device):
            check(lib.sdy_sht_inverse(plan.handle, ptr(cf), ptr(out), 1, C_, ptr(ws), nws, current_stream()),
                  "sdy_sht_inverse")
        return out[:n].reshape(*lead, self.nlat, self.nlon)
