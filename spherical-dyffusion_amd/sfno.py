"""`SphericalFourierNeuralOperatorNet` on the MI355X-native library.

Drop-in for the reference network (`src/models/sfno/sfnonet.py:340-841`) in the configuration its shipped YAML
selects (`src/configs/model/sfno.yaml`): same constructor keywords (the ones that matter for inference), same
`state_dict` names and shapes, same `forward(inputs, time=, condition=, static_condition=)`
and `predict_forward`, same inference-dropout toggles (`src/models/_base_model.py:273-295`).  The whole forward is
ONE native call (`sdy_sfno_forward`) that enqueues every HIP kernel on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import math
from contextlib import contextmanager
from typing import Dict, Optional

import torch
from torch import nn

from . import _lib
from ._lib import SdySfnoConfig, SdySfnoFwdArgs, check, current_stream, lib, ptr


class SphericalFourierNeuralOperatorNet(nn.Module):
    def __init__(
        self,
        num_input_channels: int,
        num_output_channels: int,
        num_conditional_channels: int = 0,
        spatial_shape_in=(180, 360),
        spectral_transform: str = "sht",
        filter_type: str = "linear",
        operator_type: str = "dhconv",
        scale_factor: int = 1,
        embed_dim: int = 256,
        num_layers: int = 8,
        use_mlp: bool = True,
        mlp_ratio: float = 2.0,
        activation_function: str = "gelu",
        encoder_layers: int = 1,
        pos_embed: bool = True,
        dropout_mlp: float = 0.0,
        drop_path_rate: float = 0.0,
        normalization_layer: str = "instance_norm",
        hard_thresholding_fraction: float = 1.0,
        big_skip: bool = True,
        factorization=None,
        separable: bool = False,
        with_time_emb: bool = False,
        time_dim_mult: int = 2,
        time_rescale: bool = False,
        time_scale_shift_before_filter: bool = True,
        data_grid: str = "equiangular",
        seed: int = 0,
        gemm_mode: Optional[str] = None,   # "f32" (fp32 MFMA) | "h3" (split-fp16 3-pass MFMA); default $SDY_GEMM_MODE or "f32"
        **other,
    ):
        super().__init__()
        unsupported = []
        # Keywords of the reference constructor that cannot change the numerics of this configuration (linear dhconv
        # filter, dense weights): accepted and ignored.  `dropout_filter` is one of them - the reference itself prints
        # "Dropout is not used for linear filters!" and drops it (sfnonet.py:136-137).  Everything else is an error,
        # so a checkpoint whose `model_config` asks for a different stochastic model cannot load silently.
        inert = {"params", "dropout_filter", "num_blocks", "sparsity_threshold", "use_complex_kernels", "rank",
                 "complex_network", "complex_activation", "spectral_layers", "checkpointing", "verbose", "name",
                 "loss_function", "loss_function_weights", "datamodule_config", "num_output_channels_raw",
                 "spatial_shape_out"}
        if float(other.get("pos_emb_dropout", 0.0) or 0.0) > 0.0:       # nn.Dropout on the embedding (sfnonet.py:621,827)
            unsupported.append(f"pos_emb_dropout={other['pos_emb_dropout']}")
        if other.get("debug_mode", False):                               # shrinks the network (sfnonet.py:468-471)
            unsupported.append("debug_mode=True")
        if other.get("params"):
            unsupported.append("params=<non-empty> (modulus-style parameter object)")
        unknown = sorted(set(other) - inert - {"pos_emb_dropout", "debug_mode"})
        if unknown:
            raise TypeError(f"SphericalFourierNeuralOperatorNet: unexpected keyword arguments {unknown}")
        if spectral_transform != "sht": unsupported.append(f"spectral_transform={spectral_transform}")
        if filter_type != "linear": unsupported.append(f"filter_type={filter_type}")
        if operator_type != "dhconv": unsupported.append(f"operator_type={operator_type}")
        if scale_factor != 1: unsupported.append(f"scale_factor={scale_factor}")
        if not use_mlp: unsupported.append("use_mlp=False")
        if activation_function != "gelu": unsupported.append(f"activation_function={activation_function}")
        if encoder_layers != 1: unsupported.append(f"encoder_layers={encoder_layers}")
        if normalization_layer != "instance_norm": unsupported.append(f"normalization_layer={normalization_layer}")
        if factorization is not None or separable: unsupported.append("factorized/separable weights")
        if time_rescale: unsupported.append("time_rescale=True")
        if with_time_emb and not time_scale_shift_before_filter: unsupported.append("time_scale_shift_before_filter=False")
        if unsupported:
            raise NotImplementedError("outside the hot-path scope (SURVEY.md section 8): " + ", ".join(unsupported))

        self.num_input_channels = num_input_channels
        self.num_output_channels = num_output_channels
        self.num_conditional_channels = num_conditional_channels
        self.img_shape = tuple(spatial_shape_in)
        self.in_chans = num_input_channels + num_conditional_channels     # sfnonet.py:486-490
        self.out_chans = num_output_channels
        self.embed_dim = embed_dim
        self.num_layers = num_layers
        self.mlp_hidden = int(embed_dim * mlp_ratio)
        self.dropout_mlp = float(dropout_mlp)
        self.drop_path_rate = float(drop_path_rate)
        self.big_skip = bool(big_skip)
        self.use_pos_embed = bool(pos_embed)
        self.with_time_emb = bool(with_time_emb)
        self.time_dim = embed_dim * time_dim_mult if with_time_emb else 0
        self.data_grid = data_grid
        nlat, nlon = self.img_shape
        self.h, self.w = nlat // scale_factor, nlon // scale_factor
        self.modes_lat = int(self.h * hard_thresholding_fraction)           # sfnonet.py:526-527
        self.modes_lon = int((self.w // 2 + 1) * hard_thresholding_fraction)
        self.min_time: Optional[float] = None
        self.max_time: Optional[float] = None
        self.inference_dropout = False
        self.seed = int(seed)
        self.gemm_mode = gemm_mode or _lib.default_gemm_mode()
        if self.gemm_mode not in ("f32", "h3"):
            raise ValueError(f"gemm_mode must be 'f32' or 'h3', got {self.gemm_mode!r}")
        self.batch_offset = 0           # global index of the first trajectory this rank owns
        self._call = 0                  # advances the dropout stream on every forward
        self.mask_injector = None       # tests: callable(call_index) -> (keep_masks, drop_path_keep) replacing Philox
        self.supports_shared_inputs = True     # forward(shared_inputs=True): stacked calls that share their input rows

        E, T, H, Cin, L = embed_dim, self.time_dim, self.mlp_hidden, self.in_chans, self.modes_lat
        P = lambda *s: nn.Parameter(torch.zeros(*s), requires_grad=False)  # noqa: E731
        self._params: Dict[str, nn.Parameter] = {}

        def reg(name, *shape):
            p = P(*shape)
            self.register_parameter(name.replace(".", "__"), p)
            self._params[name] = p
            return p

        reg("encoder.0.weight", E, Cin, 1, 1); reg("encoder.0.bias", E); reg("encoder.2.weight", E, E, 1, 1)
        if self.use_pos_embed:
            reg("pos_embed", 1, E, nlat, nlon)
        if with_time_emb:
            reg("time_emb_mlp.1.weight", T, E); reg("time_emb_mlp.1.bias", T)
            reg("time_emb_mlp.3.weight", T, T); reg("time_emb_mlp.3.bias", T)
        self._fc2 = "mlp.fwd.3" if self.dropout_mlp > 0.0 else "mlp.fwd.2"    # layers.py:76-80
        for i in range(num_layers):
            p = f"blocks.{i}."
            reg(p + "norm0.weight", E); reg(p + "norm0.bias", E)
            if with_time_emb:
                reg(p + "time_mlp.1.weight", 2 * E, T); reg(p + "time_mlp.1.bias", 2 * E)
            reg(p + "filter.filter.weight", E, E, L, 2); reg(p + "filter.filter.bias", 1, E, 1, 1)
            reg(p + "inner_skip.weight", E, E, 1, 1); reg(p + "inner_skip.bias", E)
            reg(p + "norm1.weight", E); reg(p + "norm1.bias", E)
            reg(p + "mlp.fwd.0.weight", H, E, 1, 1); reg(p + "mlp.fwd.0.bias", H)
            reg(p + self._fc2 + ".weight", E, H, 1, 1); reg(p + self._fc2 + ".bias", E)
        dec_in = E + (Cin if big_skip else 0)
        reg("decoder.0.weight", E, dec_in, 1, 1); reg("decoder.0.bias", E); reg("decoder.2.weight", self.out_chans, E, 1, 1)
        # nn.InstanceNorm2d default init
        for i in range(num_layers):
            for nme in ("norm0", "norm1"):
                self._params[f"blocks.{i}.{nme}.weight"].data.fill_(1.0)

        self._native: Dict[int, C.c_void_p] = {}   # device index -> sdy_sfno*
        self._native_dirty = True
        self._ws: Dict[Optional[int], torch.Tensor] = {}     # device index -> workspace (grow-only)

    # ---- state_dict with the reference's names ------------------------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        out = destination if destination is not None else {}
        for k, v in self._params.items():
            out[prefix + k] = v if keep_vars else v.detach()
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        missing, unexpected = [], []
        for k, p in self._params.items():
            if k in state_dict:
                v = state_dict[k]
                if tuple(v.shape) != tuple(p.shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(p.shape)}")
                p.data.copy_(v.detach().to(torch.float32))
            else:
                missing.append(k)
        for k in state_dict:
            if k not in self._params:
                spec = self._persisted_sht_table(k)
                if spec is not None:     # persistent SHT buffer of an old torch-harmonics: the authoritative table
                    self._check_persisted_sht_table(k, state_dict[k], *spec)
                    continue
                unexpected.append(k)
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing={missing[:5]}... unexpected={unexpected[:5]}...")
        self._native_dirty = True
        return missing, unexpected

    # ---- persisted SHT tables --------------------------------------------------------
    def _persisted_sht_table(self, key: str):
        """Older torch-harmonics releases registered `RealSHT.weights` / `InverseRealSHT.pct` as PERSISTENT buffers, so a
        checkpoint written with them carries the tables the network was trained with (under the four transform modules
        of `sfnonet.py:551-554` and again under every block's `filter.filter.{forward,inverse}_transform`).  Returns
        (kind, grid) for such a key, None for anything else."""
        leaf = key.rsplit(".", 1)[-1]
        if leaf not in ("weights", "pct"):
            return None
        owner = key[: -len(leaf) - 1]
        lg, dg = "legendre-gauss", self.data_grid
        fixed = {"trans_down": ("weights", dg), "itrans_up": ("pct", dg), "trans": ("weights", lg), "itrans": ("pct", lg)}
        if owner in fixed:
            return fixed[owner] if fixed[owner][0] == leaf else None
        parts = owner.split(".")
        if len(parts) == 5 and parts[0] == "blocks" and parts[1].isdigit() and parts[2:4] == ["filter", "filter"]:
            i = int(parts[1])
            if parts[4] == "forward_transform" and leaf == "weights":       # sfnonet.py:676: first block reads the data grid
                return "weights", (dg if i == 0 else lg)
            if parts[4] == "inverse_transform" and leaf == "pct":           # sfnonet.py:677: last block writes the data grid
                return "pct", (dg if i == self.num_layers - 1 else lg)
        return None

    _table_cache: Dict[tuple, torch.Tensor] = {}

    def _computed_sht_table(self, kind: str, grid: str) -> torch.Tensor:
        """fp32 table exactly as the native plan uploads it (fp64 on the host, cast like `.float()`)."""
        nlat, nlon = self.img_shape
        L, M = self.modes_lat, self.modes_lon
        key = (kind, grid, nlat, nlon, L, M)
        if key not in self._table_cache:
            pct = torch.zeros(M, L, nlat, dtype=torch.float64)
            w = torch.zeros(nlat, dtype=torch.float64)
            check(lib.sdy_sht_tables_host(nlat, nlon, L, M, _lib.SDY_GRID[grid], ptr(pct), ptr(w), None),
                  "sdy_sht_tables_host")
            t = pct * w[None, None, :] if kind == "weights" else pct
            self._table_cache[key] = t.to(torch.float32)
        return self._table_cache[key]

    def _check_persisted_sht_table(self, key: str, value: torch.Tensor, kind: str, grid: str) -> None:
        """First-contact check: the product recomputes its tables, so a checkpoint that was trained with different ones
        (another normalisation, phase convention, node order or quadrature) must not load silently."""
        mine = self._computed_sht_table(kind, grid)
        theirs = value.detach().to("cpu", torch.float64)
        if tuple(theirs.shape) != tuple(mine.shape):
            raise _lib.SdyError(f"persisted SHT table {key}: shape {tuple(theirs.shape)} != computed {tuple(mine.shape)} "
                                f"([mmax][lmax][nlat] for {grid}); this checkpoint's torch-harmonics lays its tables out "
                                f"differently - refusing to guess")
        scale = float(mine.abs().max())
        diff = float((theirs - mine.double()).abs().max())
        if not diff <= 4e-6 * scale:          # fp32 rounding of either side is <= 6e-8 * scale; a convention change is O(scale)
            i = int((theirs - mine.double()).abs().argmax())
            m, rem = divmod(i, mine.shape[1] * mine.shape[2])
            l, k = divmod(rem, mine.shape[2])
            raise _lib.SdyError(f"persisted SHT table {key} ({kind}, {grid}) differs from the table this library computes: "
                                f"max |diff| {diff:.3e} (table scale {scale:.3e}) at (m={m}, l={l}, k={k}).  The checkpoint "
                                f"was trained with different Legendre tables (torch-harmonics version / norm / csphase); "
                                f"results would not match the reference.")

    # ---- reference helpers ----------------------------------------------------------------------------------
    def set_min_max_time(self, min_time: float, max_time: float):     # sfnonet.py:761-773
        self.min_time, self.max_time = float(min_time), float(max_time)

    def enable_inference_dropout(self):                               # _base_model.py:288-290
        self.inference_dropout = True

    def disable_inference_dropout(self):                              # _base_model.py:292-294
        self.inference_dropout = False

    @contextmanager
    def inference_dropout_scope(self, condition: bool, context=None):  # _base_model.py:273-286
        assert isinstance(condition, bool), f"Condition must be a boolean, got {condition}"
        if condition:
            self.enable_inference_dropout()
        try:
            yield None
        finally:
            if condition:
                self.disable_inference_dropout()

    # ---- native object --------------------------------------------------------------------------------------
    def _cfg(self) -> SdySfnoConfig:
        c = SdySfnoConfig()
        c.nlat, c.nlon = self.img_shape
        c.in_chans, c.out_chans = self.in_chans, self.out_chans
        c.embed_dim, c.num_layers, c.mlp_hidden = self.embed_dim, self.num_layers, self.mlp_hidden
        c.lmax, c.mmax = self.modes_lat, self.modes_lon
        c.data_grid = _lib.SDY_GRID[self.data_grid]
        c.with_time_emb, c.time_dim = int(self.with_time_emb), self.time_dim
        c.dropout_mlp, c.drop_path_rate = self.dropout_mlp, self.drop_path_rate
        c.big_skip, c.pos_embed = int(self.big_skip), int(self.use_pos_embed)
        c.gemm_mode = 1 if self.gemm_mode == "h3" else 0
        return c

    def _get_native(self, device: torch.device) -> C.c_void_p:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._native_dirty:
            for h in self._native.values():
                lib.sdy_sfno_destroy(h)
            self._native.clear()
            self._native_dirty = False
        if idx not in self._native:
            with torch.cuda.device(idx):
                h = C.c_void_p()
                cfg = self._cfg()
                check(lib.sdy_sfno_create(C.byref(cfg), C.byref(h)), "sdy_sfno_create")
                for name, p in self._params.items():
                    t = p.detach().to("cpu", torch.float32).contiguous()
                    check(lib.sdy_sfno_set_param(h, name.encode(), ptr(t), t.numel()), f"sdy_sfno_set_param({name})")
                if self.with_time_emb:   # exactly the table SinusoidalPosEmb builds (misc.py:26-29)
                    half = self.embed_dim // 2
                    f = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).contiguous()
                    check(lib.sdy_sfno_set_param(h, b"@time_freq", ptr(f), half))
                dpr = torch.linspace(0, self.drop_path_rate, self.num_layers).to(torch.float32).contiguous()  # sfnonet.py:622
                check(lib.sdy_sfno_set_param(h, b"@drop_path_rates", ptr(dpr), self.num_layers))
                rc = lib.sdy_sfno_ready(h)
                if rc != 0:
                    raise _lib.SdyError(f"native SFNO not ready, missing {lib.sdy_sfno_missing(h).decode()}")
                self._native[idx] = h
        return self._native[idx]

    def _workspace(self, h, device, B):
        """ONE workspace per device, sized for the largest batch seen so far (the native call lays the buffer out for its own
        B and only needs `ws_floats >= sdy_sfno_workspace_floats(B)`): ragged shards, `max_batch` chunks and the 2B rows of
        a stacked interpolator pair share it instead of pinning a multi-GB buffer each.  Forwards of one network are issued
        from one host thread on one stream at a time (the same contract a per-batch cache had for equal batches)."""
        n = int(lib.sdy_sfno_workspace_floats(h, B))
        ws = self._ws.get(device.index)
        if ws is None or ws.numel() < n:
            self._ws.pop(device.index, None)     # the smaller buffer goes back to the caching allocator first
            del ws
            self._ws[device.index] = torch.empty(n, dtype=torch.float32, device=device)
        return self._ws[device.index]

    def __del__(self):
        try:
            for h in self._native.values():
                lib.sdy_sfno_destroy(h)
        except Exception:
            pass

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, inputs, time=None, condition=None, static_condition=None, return_time_emb: bool = False,
                keep_masks=None, drop_path_keep=None, rows_per_call: Optional[int] = None, reuse_encoder: bool = False,
                shared_inputs: bool = False, **kwargs):
        """`reuse_encoder=True`: the caller guarantees that `inputs` / `condition` / `static_condition` hold the same values
        as in the previous forward of this network (same batch): the input concat and the encoder are skipped and the
        forward restarts from the stored encoder output -- bit-identical results; `time` and the dropout call number may
        differ (the two interpolations of a cold-sampling step, reference dyffusion.py:497,515).
        `rows_per_call=n` (n divides the batch): the batch stacks B / n CALLS of n trajectories each -- row b draws the
        dropout stream of call number `_call + b // n`, trajectory `batch_offset + b % n`, exactly as if the calls had
        been issued one after the other, and the call counter advances by B / n.  (The two interpolator calls of a DYffusion
        sampling step share their inputs, reference dyffusion.py:497,515.)
        `shared_inputs=True` (with `rows_per_call=n`): the stacked calls share their inputs -- `inputs` / `condition` /
        `static_condition` hold n rows, `time` one value per row of the stacked batch (B = len(time)), and row b reads input
        row b % n: no stacked copy of the inputs is made and the encoder runs once (bit-identical to stacking copies)."""
        if return_time_emb:
            raise NotImplementedError("return_time_emb is a training-path feature")
        if not inputs.is_cuda:
            raise RuntimeError("sdy_amd SFNO runs on the GPU only (no CPU fallback); move inputs to cuda")
        dev = inputs.device
        # concat_condition_if_needed (_base_model.py:166-192): same checks, concat happens inside the native call
        if self.num_conditional_channels > 0:
            if condition is None and static_condition is None:
                raise ValueError(f"condition and static_condition are both None but num_conditional_channels is "
                                 f"{self.num_conditional_channels}")
        else:
            assert condition is None, "condition is not None but num_conditional_channels is 0"
            assert static_condition is None, "static_condition is not None but num_conditional_channels is 0"
        pieces = [t.to(torch.float32).contiguous() for t in (inputs, condition, static_condition) if t is not None]
        B = B_in = inputs.shape[0]
        if shared_inputs:
            assert rows_per_call == B_in and self.with_time_emb and torch.is_tensor(time) and time.numel() % B_in == 0 \
                and time.numel() > B_in, "shared_inputs: rows_per_call = the inputs' rows, one time per stacked row"
            B = time.numel()
        nlat, nlon = self.img_shape
        for t in pieces:
            assert t.shape[0] == B_in and tuple(t.shape[-2:]) == (nlat, nlon), f"bad input shape {tuple(t.shape)}"
        if sum(t.shape[1] for t in pieces) != self.in_chans:
            raise RuntimeError(f"inputs.shape: {tuple(inputs.shape)}, expected {self.in_chans} channels in total, got "
                               f"{[t.shape[1] for t in pieces]}")
        tt = None
        if self.with_time_emb:
            assert self.min_time is not None and self.max_time is not None, \
                "min_time and max_time must be set before using time embedding"          # sfnonet.py:777-779
            assert time is not None, "time is required when with_time_emb=True"
            tt = torch.as_tensor(time, dtype=torch.float32, device=dev).reshape(-1).contiguous()
            if tt.numel() == 1 and B > 1:
                tt = tt.expand(B).contiguous()
            assert tt.numel() == B
            if isinstance(time, (int, float)) or not torch.is_tensor(time) or not time.is_cuda:
                # host-side range check (the reference's device assert at sfnonet.py:780-782 forces a sync;
                # device-resident times are range-checked by the sampler, which builds them on the host)
                tv = torch.as_tensor(time, dtype=torch.float32).reshape(-1)
                assert (self.min_time <= tv).all() and (tv <= self.max_time).all(), \
                    f"time must be in [{self.min_time}, {self.max_time}], but time is {tv}"

        h = self._get_native(dev)
        max_b = int(lib.sdy_sfno_max_batch(h))
        if B > max_b:
            # One native call covers `max_b` rows (32-bit lane offsets in the spectral workspace: 60 at 180 x 360, E = 256).
            # A larger batch runs as consecutive calls on near-equal row ranges; every row keeps its dropout stream (same call
            # number, batch_offset + first row of the range), so the result equals the single call's row for row.
            if rows_per_call not in (None, B) or shared_inputs or keep_masks is not None or drop_path_keep is not None or \
                    (self.mask_injector is not None and self.inference_dropout):
                raise _lib.SdyError(f"batch {B} > {max_b} rows per native call cannot be split with stacked calls / injected masks")
            n_chunks = -(-B // max_b)
            step = -(-B // n_chunks)
            out = torch.empty(B, self.out_chans, nlat, nlon, dtype=torch.float32, device=dev)
            for r0 in range(0, B, step):       # the prepared pieces are sliced: no second pass through the checks above
                r1 = min(B, r0 + step)
                self._native_call(h, dev, [t[r0:r1] for t in pieces], None if tt is None else tt[r0:r1], out[r0:r1],
                                  self._call, self.batch_offset + r0)
            self._call += 1                    # only once every chunk has been enqueued: a failed chunk leaves the counter alone
            return out
        out = torch.empty(B, self.out_chans, nlat, nlon, dtype=torch.float32, device=dev)
        n_calls = 1
        if rows_per_call is not None:
            assert rows_per_call >= 1 and B % rows_per_call == 0, f"rows_per_call={rows_per_call} must divide the batch {B}"
            assert keep_masks is None and drop_path_keep is None and self.mask_injector is None, \
                "injected masks address one call per forward"
            n_calls = B // int(rows_per_call)
        if keep_masks is None and drop_path_keep is None and self.mask_injector is not None and self.inference_dropout:
            keep_masks, drop_path_keep = self.mask_injector(self._call)
        self._native_call(h, dev, pieces, tt, out, self._call, self.batch_offset, rows_per_call, keep_masks, drop_path_keep,
                          reuse_encoder=bool(reuse_encoder), shared_inputs=bool(shared_inputs))
        self._call += n_calls
        return out

    def _native_call(self, h, dev, pieces, tt, out, call: int, batch_offset: int, rows_per_call: Optional[int] = None,
                     keep_masks=None, drop_path_keep=None, reuse_encoder: bool = False, shared_inputs: bool = False) -> None:
        """One sdy_sfno_forward on prepared (fp32, contiguous-per-row) inputs; `out` is a (B, out_chans, nlat, nlon) view."""
        B = out.shape[0]
        ws = self._workspace(h, dev, B)
        a = SdySfnoFwdArgs()
        for i in range(3):
            a.in_[i] = ptr(pieces[i]) if i < len(pieces) else None
            a.in_chans[i] = pieces[i].shape[1] if i < len(pieces) else 0
        a.time, a.out, a.B = ptr(tt), ptr(out), B
        a.enable_dropout = int(self.inference_dropout)
        a.seed, a.call, a.batch_offset = self.seed, call & 0xFFFFFFFF, batch_offset
        if rows_per_call is not None:
            a.rows_per_call = int(rows_per_call)
        keep = []
        if keep_masks is not None:
            arr = (C.c_void_p * (2 * self.num_layers))()
            for j, m in enumerate(keep_masks):
                if m is not None:
                    mm = m.to(dev, torch.float32).contiguous()
                    keep.append(mm)
                    arr[j] = ptr(mm)
            a.keep_masks = C.cast(arr, C.POINTER(C.c_void_p))
            keep.append(arr)
        if drop_path_keep is not None:   # (num_layers, B) keep flags
            dk = drop_path_keep.to(dev, torch.float32).contiguous()
            assert dk.shape == (self.num_layers, B)
            keep.append(dk)
            a.drop_path_keep = ptr(dk)
        a.ws, a.ws_floats = ptr(ws), ws.numel()
        a.reuse_encoder = int(reuse_encoder)
        a.shared_inputs = int(shared_inputs)
        with torch.cuda.device(dev):
            check(lib.sdy_sfno_forward(h, C.byref(a), current_stream()), "sdy_sfno_forward")

    def predict_forward(self, *inputs, metadata=None, **kwargs):      # _base_model.py:265-270
        return self(*inputs, **kwargs)

    def time_embedding(self, time: torch.Tensor):
        """(t_repr (B,T), scale_shift (B, L, 2E)) exactly as the fused forward computes them (parity tap)."""
        dev = time.device
        h = self._get_native(dev)
        B = time.numel()
        tt = time.to(torch.float32).contiguous()
        trep = torch.empty(B, self.time_dim, dtype=torch.float32, device=dev)
        ss = torch.empty(B, self.num_layers, 2 * self.embed_dim, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.sdy_sfno_time_embed(h, ptr(tt), B, ptr(trep), ptr(ss), current_stream()), "sdy_sfno_time_embed")
        return trep, ss
