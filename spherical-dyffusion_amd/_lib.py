"""ctypes binding of libsdy_amd.so (the C ABI declared in include/sdy_amd.h).

The HIP library is the product: there is NO Python/PyTorch fallback for any op.  If the shared object is
missing the import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C spherical-dyffusion_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDY_AMD_LIB: another build of the same library (A/B measurements of compile-time kernel switches, csrc/Makefile); never a fallback
LIB_PATH = os.environ.get("SDY_AMD_LIB") or os.path.join(_HERE, "libsdy_amd.so")

SDY_GRID = {"equiangular": 0, "legendre-gauss": 1}


class SdyError(RuntimeError):
    pass


class SdyConvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bstride", C.c_long),
        ("wt", C.c_void_p), ("ldw", C.c_int),
        ("out", C.c_void_p), ("out_bstride", C.c_long),
        ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("HW", C.c_int),
        ("pa", C.c_void_p), ("pd", C.c_void_p),
        ("bias", C.c_void_p),
        ("add", C.c_void_p), ("add_bstride", C.c_long),
        ("add_mode", C.c_int),
        ("act", C.c_int),
        ("drop_p", C.c_float),
        ("keep_mask", C.c_void_p),
        ("seed", C.c_uint64), ("call", C.c_uint32), ("stream_id", C.c_uint32), ("batch_offset", C.c_uint32),
        ("rows_per_call", C.c_int),
        ("batch_scale", C.c_void_p),
        ("kernel_tag", C.c_int),
        ("w_h3", C.c_void_p),
        ("w_h3_scale", C.c_float),
        ("w_frag", C.c_void_p),
        ("w_frag_scale", C.c_float),
        ("stats", C.c_void_p),
        ("out_tiled", C.c_int),
        ("x_rows", C.c_void_p),
    ]


class SdyMlpArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bstride", C.c_long),
        ("pa", C.c_void_p), ("pd", C.c_void_p),
        ("w", C.c_void_p), ("w1_scale", C.c_float), ("w2_scale", C.c_float),
        ("b1", C.c_void_p), ("b2", C.c_void_p),
        ("out", C.c_void_p), ("out_bstride", C.c_long),
        ("add", C.c_void_p), ("add_bstride", C.c_long),
        ("add_a", C.c_void_p), ("add_d", C.c_void_p),
        ("B", C.c_int), ("E", C.c_int), ("hidden", C.c_int), ("HW", C.c_int),
        ("drop_p", C.c_float),
        ("seed", C.c_uint64), ("call", C.c_uint32), ("stream_fc1", C.c_uint32), ("stream_fc2", C.c_uint32),
        ("batch_offset", C.c_uint32),
        ("rows_per_call", C.c_int),
        ("batch_scale", C.c_void_p),
        ("stats", C.c_void_p),
        ("x_tiled", C.c_int),
        ("keep_hidden", C.c_void_p), ("keep_out", C.c_void_p),
        ("out_rows", C.c_void_p),
        ("add_by_launch_row", C.c_int),
    ]


class SdyPairArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x_bstride", C.c_long),
        ("w", C.c_void_p), ("w1_scale", C.c_float), ("w2_scale", C.c_float),
        ("b1", C.c_void_p),
        ("out", C.c_void_p), ("out_bstride", C.c_long),
        ("add", C.c_void_p), ("add_bstride", C.c_long),
        ("B", C.c_int), ("Cin", C.c_int), ("hidden", C.c_int), ("Cout", C.c_int), ("HW", C.c_int),
        ("stats", C.c_void_p),
    ]


SDY_MAX_VARS = 96


class SdyVarTable(C.Structure):
    _fields_ = [("nvars", C.c_int), ("data", C.c_void_p * SDY_MAX_VARS), ("mean", C.c_float * SDY_MAX_VARS),
                ("std", C.c_float * SDY_MAX_VARS)]


class SdyStepFinishArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("HW", C.c_int), ("T1", C.c_int), ("t", C.c_int),
        ("gen", C.c_void_p), ("n_out", C.c_int),
        ("prev_in", C.c_void_p), ("next_in", C.c_void_p), ("n_in", C.c_int),
        ("n_entries", C.c_int),
        ("out_idx", C.c_int * SDY_MAX_VARS), ("in_idx", C.c_int * SDY_MAX_VARS),
        ("gen_norm_tl", C.c_void_p * SDY_MAX_VARS), ("gen_tl", C.c_void_p * SDY_MAX_VARS),
        ("mean", C.c_float * SDY_MAX_VARS), ("std", C.c_float * SDY_MAX_VARS),
        ("presc_entry", C.c_int),
        ("presc_target", C.c_void_p), ("presc_mask", C.c_void_p),
        ("mask_value", C.c_int), ("interpolate", C.c_int),
        ("ar_init", C.c_void_p),
    ]


class SdySfnoConfig(C.Structure):
    _fields_ = [
        ("nlat", C.c_int), ("nlon", C.c_int),
        ("in_chans", C.c_int), ("out_chans", C.c_int),
        ("embed_dim", C.c_int), ("num_layers", C.c_int), ("mlp_hidden", C.c_int),
        ("lmax", C.c_int), ("mmax", C.c_int),
        ("data_grid", C.c_int),
        ("with_time_emb", C.c_int), ("time_dim", C.c_int),
        ("dropout_mlp", C.c_float), ("drop_path_rate", C.c_float),
        ("big_skip", C.c_int), ("pos_embed", C.c_int),
        ("gemm_mode", C.c_int),
    ]


class SdySfnoFwdArgs(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p * 3), ("in_chans", C.c_int * 3),
        ("time", C.c_void_p),
        ("out", C.c_void_p),
        ("B", C.c_int),
        ("enable_dropout", C.c_int),
        ("seed", C.c_uint64), ("call", C.c_uint32), ("batch_offset", C.c_uint32),
        ("rows_per_call", C.c_int),
        ("keep_masks", C.POINTER(C.c_void_p)),
        ("drop_path_keep", C.c_void_p),
        ("ws", C.c_void_p), ("ws_floats", C.c_size_t),
        ("reuse_encoder", C.c_int),
        ("shared_inputs", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/sdy_amd.h declares
SIGNATURES = {
    "sdy_version": (C.c_int, []),
    "sdy_error_string": (C.c_char_p, [C.c_int]),
    "sdy_abi_check": (C.c_int, [C.POINTER(C.c_size_t), C.c_int]),
    "sdy_sht_tables_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdy_sht_plan_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdy_sht_plan_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sdy_sht_plan_destroy": (None, [C.c_void_p]),
    "sdy_sht_plan_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int * 6)]),
    "sdy_sht_workspace_floats": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "sdy_sht_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdy_sht_inverse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdy_rfft_lon": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdy_legendre_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdy_legendre_inv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdy_irfft_lon": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sdy_dhconv_pack_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sdy_dhconv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdy_dhconv_h3_pack_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sdy_dhconv_h3_pack_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "sdy_dhconv_h3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "sdy_dhconv_frag_supported": (C.c_int, [C.c_int, C.c_int]),
    "sdy_dhconv_frag_pack_bytes": (C.c_size_t, [C.c_int]),
    "sdy_dhconv_frag_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "sdy_dhconv_frag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "sdy_instnorm_coeffs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdy_conv1x1": (C.c_int, [C.POINTER(SdyConvArgs), C.c_void_p]),
    "sdy_ensemble_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p]),
    "sdy_instnorm_from_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_long, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdy_mlp_h3_supported": (C.c_int, [C.c_int, C.c_int]),
    "sdy_mlp_h3_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sdy_mlp_h3_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                  C.POINTER(C.c_float)]),
    "sdy_mlp_h3": (C.c_int, [C.POINTER(SdyMlpArgs), C.c_void_p]),
    "sdy_pair_h3_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "sdy_pair_h3_pack_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sdy_pair_h3_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                   C.POINTER(C.c_float)]),
    "sdy_pair_h3": (C.c_int, [C.POINTER(SdyPairArgs), C.c_void_p]),
    "sdy_conv256_h3_supported": (C.c_int, [C.c_int, C.c_int]),
    "sdy_conv256_h3_pack_bytes": (C.c_size_t, []),
    "sdy_conv256_h3_pack_bytes_cin": (C.c_size_t, [C.c_int]),
    "sdy_conv256_h3_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "sdy_conv256_h3_pack_cin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "sdy_h3_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sdy_h3_pack_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "sdy_sfno_create": (C.c_int, [C.POINTER(SdySfnoConfig), C.POINTER(C.c_void_p)]),
    "sdy_sfno_destroy": (None, [C.c_void_p]),
    "sdy_sfno_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "sdy_sfno_ready": (C.c_int, [C.c_void_p]),
    "sdy_sfno_missing": (C.c_char_p, [C.c_void_p]),
    "sdy_sfno_workspace_floats": (C.c_size_t, [C.c_void_p, C.c_int]),
    "sdy_sfno_max_batch": (C.c_int, [C.c_void_p]),
    "sdy_sfno_forward": (C.c_int, [C.c_void_p, C.POINTER(SdySfnoFwdArgs), C.c_void_p]),
    "sdy_sfno_time_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdy_norm_pack": (C.c_int, [C.POINTER(SdyVarTable), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sdy_step_finish": (C.c_int, [C.POINTER(SdyStepFinishArgs), C.c_void_p]),
    "sdy_init_timeline": (C.c_int, [C.POINTER(SdyVarTable), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p), C.c_void_p]),
    "sdy_lp_rel_terms": (C.c_int, [C.c_void_p, C.POINTER(SdyVarTable), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]),
    "sdy_cold_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sdy_concat_channels": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_void_p]),
    "sdy_time_mean_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_void_p, C.c_void_p]),
    "sdy_status_flags": (C.c_int, [C.POINTER(C.c_uint), C.c_int, C.c_void_p]),
    "sdy_status_flags_async": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "sdy_range_headroom_enable": (C.c_int, [C.c_int]),
    "sdy_range_headroom": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "sdy_dropout_stream_rounds": (C.c_int, []),
    "sdy_dropout_stream_words": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_uint32)]),
    "sdy_ensemble_series": (C.c_int, [C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "sdy_profile_enable": (C.c_int, [C.c_int]),
    "sdy_profile_stage_count": (C.c_int, []),
    "sdy_profile_stage_name": (C.c_char_p, [C.c_int]),
    "sdy_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_long), C.c_int]),
    "sdy_profile_read_rows": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long), C.c_int]),
}


def _load():
    # torch first: PyTorch-ROCm ships its own libamdhip64, and the process must end up with ONE HIP runtime.  Loaded after
    # torch, libsdy_amd.so binds to the runtime torch already brought in (same soname); loaded BEFORE it, the library pulls
    # /opt/rocm's copy, torch adds its own, and every sdy_* call then fails with "no ROCm-capable device is detected"
    # (seen on the GPU box with `import sdy_amd` ahead of `import torch`).
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is required (no CPU/PyTorch fallback exists). "
            "Build it with `make -C spherical-dyffusion_amd/csrc` or `__graft_entry__.build()`."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    # the argument structures carry no size field: the bindings' layouts must be the library's (include/sdy_amd.h, sdy_abi_check)
    structs = (SdyConvArgs, SdyMlpArgs, SdyPairArgs, SdySfnoConfig, SdySfnoFwdArgs, SdyVarTable, SdyStepFinishArgs)
    sizes = (C.c_size_t * len(structs))(*[C.sizeof(t) for t in structs])
    if lib.sdy_abi_check(sizes, len(structs)) != 0:
        raise ImportError(f"{LIB_PATH}: argument structures of the bindings and of the library differ in size "
                          f"({[C.sizeof(t) for t in structs]}): rebuild the library (make -C spherical-dyffusion_amd/csrc)")
    return lib


lib = _load()


def default_gemm_mode() -> str:
    """"h3" (split-fp16 3-pass MFMA, the default) or "f32" (fp32-input MFMA); both hold the same parity tolerances."""
    m = os.environ.get("SDY_GEMM_MODE", "h3")
    if m not in ("h3", "f32"):
        raise ValueError(f"SDY_GEMM_MODE must be 'h3' or 'f32', got {m!r}")
    return m


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.sdy_error_string(rc).decode()
        raise SdyError(f"{what or 'sdy call'} failed with code {rc}: {msg}")


def ptr(t) -> int:
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
