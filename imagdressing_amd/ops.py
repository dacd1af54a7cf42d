"""Typed wrappers around the C ABI that take torch tensors as *device memory handles* only
(``data_ptr()`` + the current HIP stream).  No torch math happens here; CPU tensors, wrong dtypes
and non-contiguous views raise -- there is no eager fallback.
"""
from __future__ import annotations

import functools
import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib as L

ACT_NONE, ACT_SILU, ACT_GEGLU, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2, 3, 4
bf16 = torch.bfloat16
f16 = torch.float16
DTYPE_CODE = {torch.bfloat16: 0, torch.float16: 1}     # IMD_DTYPE_*


def _code(t: torch.Tensor, name: str = "tensor") -> int:
    """Element-type code of a 16-bit activation/weight tensor (bf16 or fp16; both run the MFMA at the
    same rate, fp16 is what the reference computes in)."""
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise L.ImdError(f"{name}: expected a bfloat16 or float16 tensor, got {t.dtype}") from None


try:        # the raw handle of torch's current stream without building a torch.cuda.Stream object per call (6 x ~4 us per processor call)
    _raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:      # (a torch build without the private entry points)
    _raw_stream = _cur_dev = None


def _stream() -> int:
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, dtype, name: str) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise L.ImdError(f"{name}: tensor is on {t.device}; imagdressing_amd runs on MI355X only (no CPU path)")
    if t.dtype != dtype:
        raise L.ImdError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.ImdError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], dtype, name: str) -> Optional[int]:
    return None if t is None else _dev(t, dtype, name)


_checked_devices = set()


def ensure_device(device: torch.device):
    if device.type != "cuda":
        raise L.ImdError(f"tensor is on {device}; imagdressing_amd runs on MI355X (gfx950) only -- there is no CPU path")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _checked_devices:
        L.check(L.load().imd_device_check(idx))
        _checked_devices.add(idx)


# ---------------------------------------------------------------------------------------------
# persistent zero-initialised workspaces (attention Q/K/V^T buffers rely on their padding staying 0)
# ---------------------------------------------------------------------------------------------
_ws: Dict[Tuple, torch.Tensor] = {}


def workspace(tag: str, shape: Sequence[int], dtype, device, init=None) -> torch.Tensor:
    """Persistent scratch keyed by (tag, shape, dtype, device, CURRENT STREAM): launches on one stream are ordered, so one buffer
    per stream is race-free; two pipelines driven on two streams of one process get separate buffers instead of silently sharing."""
    key = (tag, tuple(shape), dtype, device.type, device.index, _stream()) if isinstance(device, torch.device) else (tag, tuple(shape), dtype, str(device), None, _stream())
    t = _ws.get(key)
    if t is None:
        t = torch.zeros(tuple(shape), dtype=dtype, device=device)
        if init is not None:
            init(t)
        _ws[key] = t
    return t


def k_buffer(shape: Sequence[int], D: int, dtype, device, tag: Optional[str] = None) -> torch.Tensor:
    """Zeroed K operand [Bk, H, L, DPK] of the attention kernels.  When the head dim leaves pad columns (D = 40 -> DPK = 48)
    column D is set to 1.0 ONCE: it is the slot through which the deferred row maximum enters the QK^T MFMA
    (``imd_attn_params.k_pad_one``).  The projection epilogue writes columns [0, D) only, so the 1 persists."""
    def init(t):
        if t.shape[-1] > D:
            t[..., D] = 1.0
    if tag is not None:
        return workspace(tag, shape, dtype, device, init=init)
    t = torch.zeros(tuple(shape), dtype=dtype, device=device)
    init(t)
    return t


def clear_workspaces(stream: Optional[int] = None):
    """Drop every persistent scratch buffer (attention operand buffers, GroupNorm partials, split-K slabs, cached fp8 K / V).
    ``stream`` (a raw stream handle, ``torch.cuda.Stream.cuda_stream``): only the buffers keyed by THAT stream -- what a pipeline calls
    when it releases its side stream / captured step graph, so that a later stream that happens to reuse the handle value cannot pick up
    a buffer the caching allocator still associates with the old stream, and nothing accumulates per stream for the process lifetime."""
    if stream is None:
        _CFG_DECISIONS.clear()
        _ws.clear()
        _splitk_ws.clear()
        _splitk_cnt.clear()
        _proj_cnt.clear()
        for fn in _clear_hooks:
            fn()
        return
    for table in (_ws, _splitk_ws, _splitk_cnt, _proj_cnt):
        for key in [k for k in table if k[-1] == stream]:
            del table[key]


_clear_hooks = []          # modules holding device-side caches of their own register a clearer here (adapter/attention_processor.py)

# ---------------------------------------------------------------------------------------------
# algorithmic FLOP accounting (bench.py: `flops_per_step`): when FLOP_COUNTER is a dict the matrix-shaped wrappers add the
# FLOPs their launch is DEFINED to compute (2 M N K per GEMM / conv, 4 N L d per attention key set; padding excluded).
# ---------------------------------------------------------------------------------------------
FLOP_COUNTER: Optional[dict] = None


def _count(kind: str, flops: float):
    c = FLOP_COUNTER
    if c is not None:
        c[kind] = c.get(kind, 0.0) + float(flops)


def attn_padded_dims(D: int) -> Tuple[int, int]:
    a, b = C.c_int(), C.c_int()
    L.check(L.load().imd_attn_padded_dims(D, C.byref(a), C.byref(b)))
    return a.value, b.value


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


# ---------------------------------------------------------------------------------------------
# Per-shape tile / split-K choices measured on MI355X by tools/gemm_tune.py (like a BLAS tuning file); shapes that
# are not listed fall back to the library heuristic.  Key: "M,N,K,taps,stride,ups".
PATCH_CONV = True          # untabulated 3x3 stride-1 convs on maps >= PATCH_MIN_W wide use the halo-patch kernel (cfg 5)
PATCH_MIN_W = 32
SPLITK_IN_KERNEL = False   # opt-in: K slices summed by each tile's last-arriving workgroup instead of by the finish launch (bit-identical;
                           # measured SLOWER end to end, 660.5 -> 687.5 ms: the slab traffic must bypass the per-XCD L2s)
FUSED_FF = True            # engines run norm3 -> GEGLU feed-forward -> + residual of the 320-channel blocks as one launch (ff_fused.hip)
FUSED_FF_MIN_ROWS = 24576  # below this the 128-row workgroups cannot fill the chip (one per CU at 32768 rows) and the tiled kernels win
import os as _os
FUSED_GN_STATS = _os.environ.get("IMD_FUSED_GN_STATS", "1") != "0"   # 3x3 convs on the halo-patch kernel emit the GroupNorm statistics of their output from the epilogue (A/B switch)
CFG_PAIR_DEDUP = _os.environ.get("IMD_CFG_PAIR_DEDUP", "1") != "0"   # sampling loop: conv_in + first resnet once for the two identical CFG halves (A/B switch)
# A/B only (round 6, measured slower -- DESIGN section 6): GroupNorm + SiLU of a ResNet's 3x3 convolutions applied inside the halo-patch kernel while
# its patch is staged (register-staged form, coefficients from group_norm_coeffs) instead of by a gn_apply launch in front of the LDS-DMA form
FUSED_GN_CONV = _os.environ.get("IMD_FUSED_GN_CONV", "0") == "1"
# (round 6) ResnetBlock2D.norm2 + SiLU inside the finish launch of a K-sliced conv1 (the 16x16 / 8x8 levels): one launch and one HBM round trip less per block.
# OPT-IN: correct and tested, measured neutral at the bench batch (592.0 vs 592.6 ms) and 1.1 % SLOWER at batch 1 (350.8 -> 354.9 ms), same box,
# interleaved twice (profiles/r6h_*): a workgroup per (image, group) reads 160-byte column strips of the fp32 slabs where the plain finish reads whole rows
FUSED_GN_FINISH = _os.environ.get("IMD_FUSED_GN_FINISH", "0") == "1"
# (round 6) Transformer2DModel.norm inside proj_in's row-resident launch (gn_in_*): the normalised tensor never exists in memory (A/B switch)
FUSED_GN_PROJ = _os.environ.get("IMD_FUSED_GN_PROJ", "1") != "0"
# (round 6) the skip concatenation of an up block also writes the GroupNorm statistics of its output: norm1 of the resnet behind it skips its statistics launch (A/B switch)
FUSED_CONCAT_STATS = _os.environ.get("IMD_FUSED_CONCAT_STATS", "1") != "0"
# (round 6) a residual that repeats over the batch (the two halves of a CFG batch) is read in place by the K = 320 row-resident projection instead of being repeated first (A/B switch)
PERIODIC_RES = _os.environ.get("IMD_PERIODIC_RES", "1") != "0"
# (round 6) GroupNorm statistics from the epilogue of the register-staged tile kernel (conv_in, the 64x64-level downsampler, proj_out of the 8x8 level: three
# statistics launches per forward).  OPT-IN: correct and tested, but measured neutral at batch 4 (twelve interleaved pairs: 563.95 vs 563.92 ms) and 0.36 % SLOWER at
# batch 1 (six pairs: 342.7 vs 344.0 ms) -- the reduction at the end of the tile kernel sits on the critical path of launches that do not fill the chip
# (profiles/r6ah_*)
GENERIC_GN_STATS = _os.environ.get("IMD_GENERIC_GN_STATS", "0") == "1"
# (round 6) the pipelines compute the time embeddings of a whole schedule in one pass before the loop (unet._Encoder.precompute_time_embeddings) (A/B switch)
TEMB_TABLE = _os.environ.get("IMD_TEMB_TABLE", "1") != "0"
# ... on which row-resident kernels (A/B): the prologue costs 6-8 us per launch in the running loop (profiles/r6final_kernel_trace_summary.md) -- less than the 10.3 us
# gn_apply launch it replaces at the 64x64 level (tile config 12), about what the 5.4 / 4.3 us launches of the 32x32 / 16x16 levels (13 / 14) cost WITH their launch
# boundary: all levels vs the 64x64 level only measured 593.0 vs 593.0 ms over four pairs (profiles/r6n_*) -> all levels (fewer launches, fewer bytes)
FUSED_GN_PROJ_CFGS = tuple(int(c) for c in _os.environ.get("IMD_FUSED_GN_PROJ_CFGS", "12,13,14").split(",") if c)
CFG_PAIR_ATTN = _os.environ.get("IMD_CFG_PAIR_ATTN", "1") != "0"     # ... and the first hybrid block up to its self-attention phase (unet.Transformer2D.call_pair_half; A/B switch)
FUSED_LN = True            # engines hand `LayerNorm -> attn2.to_q` on 320 channels to the row-resident kernel as ONE launch (A/B switch)
GEMM_TRACE = None          # tools/gemm_tune.py sets this to a list to record the shapes a forward pass launches
GEMM_EVENT_HOOK = None     # tools/insitu_conv.py sets this to a dict: every conv_gemm launch is bracketed by HIP events, keyed by (shape key, cfg, split)
_GEMM_TABLE = None
_CFG_DECISIONS: Dict[Tuple, Tuple[int, int]] = {}      # conv_gemm: problem description -> (tile config, K slices), see there

# ---- per-call tuning (include/imagdressing_hip.h: IMD_TUNING_PER_CALL) ----------------------------------------------------------------
# imd_set_tuning() is process-wide.  Inside ``with tuning_scope(...)`` every params block built by this module carries the scope's choice in
# its `flags` field instead, so two pipelines of one process can run different settings (PipelineBase.set_tuning) and nothing global changes.
TUNING_PER_CALL = 0x5A000000   # 8-bit tag in bits 24..31 (ABI v9): anything else non-zero there is refused by the library
ATTN_VARIANT_MAX = 13      # head-dim-40 variants a product build accepts (14..54 exist in -DIMD_ABLATIONS builds only)


import contextvars

# None | dict(attn_variant=int|None, attn_xcd=bool|None, gemm_flags=int|None).  A ContextVar, not a module global: two pipelines driven from
# two threads (each on its own stream) keep their own scopes -- one thread's __exit__ cannot restore over the other's active scope.
_TUNING = contextvars.ContextVar("imd_tuning_scope", default=None)


class tuning_scope:
    """``with ops.tuning_scope(attn_variant=12, gemm_flags=3): ...`` -- head-dim-40 attention variant (knob 0), XCD-aware attention work
    order (knob 1) and bits 0..4 of the GEMM tuning flags (knob 2) for the launches issued inside, per call.  Scopes nest; None = inherit.
    The scope belongs to the calling thread / context.  It reaches ``ops.conv_gemm`` and ``ops.attention`` only: the fp8 attention, the
    row-resident projections (``row_linear`` / ``row_qkv``) and the fused feed-forward have no per-call choice and ignore it."""

    def __init__(self, attn_variant=None, attn_xcd=None, gemm_flags=None):
        self.new = dict(attn_variant=attn_variant, attn_xcd=attn_xcd, gemm_flags=gemm_flags)

    def __enter__(self):
        merged = dict(_TUNING.get() or {})
        merged.update({k: v for k, v in self.new.items() if v is not None})
        self.token = _TUNING.set(merged if any(v is not None for v in merged.values()) else None)
        return self

    def __exit__(self, *exc):
        _TUNING.reset(self.token)
        return False


def _gemm_call_flags() -> int:
    t = _TUNING.get()
    if t is None or t.get("gemm_flags") is None:
        return 0
    return TUNING_PER_CALL | (int(t["gemm_flags"]) & 31)


def _attn_call_flags() -> int:
    t = _TUNING.get()
    if t is None or (t.get("attn_variant") is None and t.get("attn_xcd") is None):
        return 0
    v = int(t.get("attn_variant") or 0)
    if not 0 <= v <= ATTN_VARIANT_MAX:
        raise L.ImdError(f"tuning_scope: attention variant {v} out of range 0..{ATTN_VARIANT_MAX} (imd_set_tuning(0, .) of the product build)")
    if t.get("attn_xcd") is None:           # inherit the process-wide order
        xcd = bool(L.load().imd_get_tuning(1))
    else:
        xcd = bool(t["attn_xcd"])
    return TUNING_PER_CALL | v | (0 if xcd else 256)


def _gemm_table() -> dict:
    global _GEMM_TABLE
    if _GEMM_TABLE is None:
        import json
        import os
        path = os.environ.get("IMD_GEMM_TUNING") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning.json")      # (override: A/B of two tables)
        try:
            with open(path) as f:
                _GEMM_TABLE = json.load(f).get("shapes", {})
        except FileNotFoundError:
            _GEMM_TABLE = {}
    return _GEMM_TABLE


def conv_gemm(
    x: torch.Tensor, w: torch.Tensor, *, M: int, N: int, Cin: int, taps: int = 1,
    Hin: int = 1, Win: int = 1, Hout: int = 1, Wout: int = 1, stride: int = 1, ups: bool = False,
    x_pix_stride: Optional[int] = None, out: Optional[torch.Tensor] = None, out_ld: Optional[int] = None,
    bias: Optional[torch.Tensor] = None, rowvec: Optional[torch.Tensor] = None, rowvec_stride: int = 0,
    rowvec_off: int = 0, res: Optional[torch.Tensor] = None, res_ld: Optional[int] = None, out_scale: float = 1.0,
    act: int = ACT_NONE, out_f32: bool = False,
    heads: Optional[dict] = None, cfg: int = -1, split_k: int = 0,
    gn: Optional[tuple] = None, pad_br_only: bool = False, ln_eps: Optional[float] = None, gn_stats_groups: int = 0,
    gn_out: Optional[tuple] = None, gn_in: Optional[tuple] = None,
) -> Optional[torch.Tensor]:
    """out[M, N] = epilogue(A(M, K) @ w[N, K]^T); see include/imagdressing_hip.h::imd_conv_gemm.

    ``heads`` = dict(C=, H=, D=, dests=[(tensor|None, kind, DP, L, scale), ...]) selects the head-split
    epilogue (no ``out``).  Returns the output tensor (allocated when ``out`` is None).
    ``gn`` = (coef_a [B, Cin] fp32, coef_b [B, Cin] fp32, silu) from :func:`group_norm_coeffs` fuses GroupNorm(+SiLU)
    of the input into the 3x3 halo-patch kernel (tile config 5).
    ``ln_eps``: LayerNorm WITHOUT affine over the K channels of every row of ``x`` is applied on the fly (row-resident kernel,
    K = 320 and N <= 320 only; fold gamma / beta into ``w`` / ``bias`` with :func:`fold_layernorm_affine`).
    ``gn_stats_groups`` = G: when the launch lands on the halo-patch kernel without K slices, or is K-sliced with a separate finish launch,
    ``gn_in`` = (gamma, beta, eps, silu, groups): GroupNorm (+ SiLU) of the INPUT ``x`` [B, HW, K] of a plain linear layer.  Where the layer runs on a
    row-resident projection kernel (tile configs 12 / 13 / 14) and ``x`` carries its producer's statistics (``_imd_gn_stats``) the normalisation happens inside
    that launch (``imd_conv_gemm_params.gn_in_*``: bit-identical, no normalised tensor in memory); everywhere else :func:`group_norm` runs first.
    ``gn_out`` = (gamma, beta, eps, silu, groups): where the problem is K-sliced with a separate finish launch and that launch can own whole
    (image, group) slabs (``imd_conv_gemm_gn_out_supported``: the 16x16 / 8x8 levels), the finish launch applies GroupNorm (+ SiLU) to its output
    itself; the returned tensor then carries ``_imd_gn_applied = True`` and holds the NORMALISED values.  Ignored (raw output) everywhere else.
    the epilogue / the finish launch also writes the GroupNorm(G) statistics of the OUTPUT (per-tile / per-pixel-part fp32 partials); they ride on the returned tensor (``_imd_gn_stats``) and the next
    :func:`group_norm` of that tensor skips its statistics pass.  Silently not produced on every other path (FUSED_GN_STATS = False: never).
    """
    ensure_device(x.device)
    K = taps * Cin
    p = L.ConvGemmParams()
    p.flags = _gemm_call_flags()
    dt = x.dtype
    p.dtype = _code(x, "x")
    p.x = _dev(x, dt, "x")
    p.w = _dev(w, dt, "w")
    if w.numel() != N * K:
        raise L.ImdError(f"conv_gemm: weight has {w.numel()} elements, expected N*K = {N}*{K}")
    p.M, p.N, p.K = M, N, K
    p.Cin, p.taps, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.ups = Cin, taps, Hin, Win, Hout, Wout, stride, int(ups)
    p.x_pix_stride = Cin if x_pix_stride is None else x_pix_stride
    p.bias = _opt(bias, torch.float32, "bias")
    p.rowvec = _opt(rowvec, torch.float32, "rowvec")
    if rowvec is not None and rowvec_off:
        if rowvec_off % 4:
            raise L.ImdError("conv_gemm: rowvec_off must be a multiple of 4")
        p.rowvec = p.rowvec + 4 * rowvec_off
    p.rowvec_stride = rowvec_stride
    p.res = _opt(res, dt, "res")
    p.res_ld = (N if res_ld is None else res_ld)
    # a residual with FEWER rows than the output is periodic (row m adds res[m % rows]): one copy of a tensor that is the same for both halves of a
    # CFG batch.  The K = 320 row-resident projection reads it in place (res_rows); everywhere else it is repeated into a full-size tensor first.
    res_rows = 0
    if res is not None and res_ld is None and res.numel() != M * N:
        res_rows = res.numel() // N
        if res_rows <= 0 or res_rows * N != res.numel() or M % res_rows:
            raise L.ImdError(f"conv_gemm: the residual has {res.numel()} elements: neither M x N = {M} x {N} nor a whole divisor of it")
    p.out_scale = out_scale
    p.pad_br_only = int(pad_br_only)
    p.act = act
    p.out_f32 = int(out_f32)
    if heads is not None:
        p.mode = 1
        p.hC, p.hH, p.hD = heads["C"], heads["H"], heads["D"]
        for i, (t, kind, DP, Ltok, scale) in enumerate(heads["dests"]):
            p.hd[i].ptr = None if t is None else _dev(t, dt, f"heads[{i}]")
            p.hd[i].kind, p.hd[i].DP, p.hd[i].L, p.hd[i].scale = kind, DP, Ltok, scale
        p.out = None
        p.out_ld = 0
    else:
        p.mode = 0
        n_out = N // 2 if act == ACT_GEGLU else N
        if out is None:
            out = torch.empty((M, n_out), dtype=torch.float32 if out_f32 else dt, device=x.device)
        p.out = _dev(out, torch.float32 if out_f32 else dt, "out")
        p.out_ld = n_out if out_ld is None else out_ld
    _count("gemm_conv", 2.0 * M * N * K)
    lib = L.load()
    if gn is not None:
        p.gn_a, p.gn_b, p.gn_silu = _dev(gn[0], torch.float32, "gn_a"), _dev(gn[1], torch.float32, "gn_b"), int(gn[2])
        if cfg == -1:
            cfg = 5
    if ln_eps is not None:
        if res_rows:
            if K == 320 and res_rows % 128 == 0 and PERIODIC_RES:
                p.res_rows = res_rows
            else:
                res = repeat_batch(res.reshape(res_rows, N), M // res_rows)
                p.res = _dev(res, dt, "res")
        p.split_k = 1
        L.check(lib.imd_row_linear(C.byref(p), 1, float(ln_eps), _stream()))
        return out
    splittable = heads is None and act != ACT_GEGLU
    if GEMM_TRACE is not None:
        GEMM_TRACE.append(dict(M=M, N=N, K=K, Cin=Cin, taps=taps, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride,
                               ups=int(ups), splittable=splittable, dtype=str(dt)))
    # (round 6) the (tile config, K slices) decision of a call site is a pure function of the problem description: remembered per description, so
    # that a repeated layer pays neither the table key formatting nor the library's *_supported queries again (~5 of the ~75 us a processor call
    # costs on the host at the small levels).  Dropped with the tuning table (IMD_GEMM_TUNING / _GEMM_TABLE reset) and by clear_workspaces().
    dkey = None
    if cfg == -1 and split_k == 0 and GEMM_TRACE is None:
        hk = None if heads is None else (heads["C"], heads["H"], heads["D"], tuple((t is None, kind, DP, Ltok) for t, kind, DP, Ltok, _ in heads["dests"]))
        dkey = (M, N, K, Cin, taps, stride, int(ups), Hin, Win, Hout, Wout, p.x_pix_stride, p.res_ld, p.out_ld, act, int(out_f32), int(pad_br_only), p.dtype,
                rowvec is None, res is None, bias is None, hk, PATCH_CONV, id(_GEMM_TABLE))
        hit = _CFG_DECISIONS.get(dkey)
        if hit is not None:
            cfg, split_k = hit
    if cfg == -1 and split_k == 0:
        # 3x3 convs are keyed WITH their output map as well (round 3): the same (M, N, K) occurs for different maps -- 2 images of
        # 20x16 and 8 images of 10x8 are both 640 rows -- and the halo-patch kernel only takes maps at least 16 wide
        key = f"{M},{N},{K},{taps},{stride},{int(ups)}"
        ent = (_gemm_table().get(f"{key}|{Hout}x{Wout}") if taps == 9 else None) or _gemm_table().get(key)
        if ent is not None:
            if splittable:
                cfg, split_k = ent["cfg"], ent["split"]
            else:
                cfg, split_k = ent["cfg_nosplit"], 1
            # the table is keyed by (M, N, K, taps, stride, ups) only: another geometry with the same key (W % 16 != 0,
            # pad_br_only, strided pixels ...) may not qualify for the halo-patch kernel -> back to the library heuristic
            if cfg == 5 and not lib.imd_conv_patch_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if cfg in (12, 13, 14, 15) and not lib.imd_row_linear_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if cfg in (16, 17, 19, 25, 27, 30, 31, 32) and not lib.imd_gemm_dma_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if (cfg in (18, 20) and (taps != 9 or Cin % 32 or stride not in (1, 2) or gn is not None)) or \
                    (cfg in (26, 28) and (taps != 9 or Cin % 64 or stride not in (1, 2) or gn is not None)):
                cfg, split_k = -1, 0
            if cfg == 21 and not lib.imd_conv_patch2_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if cfg == 22 and not lib.imd_conv_patch3_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if cfg == 23 and not lib.imd_conv_patch4_supported(C.byref(p)):
                cfg, split_k = -1, 0
            if cfg == 24:                       # whole-map kernel of the 8-wide levels: K-sliced only
                p.split_k = split_k
                if not lib.imd_conv_img_supported(C.byref(p)):
                    cfg, split_k = -1, 0
    # shapes outside the measured table: 3x3 stride-1 convs on wide maps go to the halo-patch kernel (always ahead of the
    # gather kernel there: profiles/r1k_patch_conv_ab.jsonl)
    if PATCH_CONV and cfg == -1 and taps == 9 and stride == 1 and Wout >= PATCH_MIN_W and N >= 64 \
            and lib.imd_conv_patch_supported(C.byref(p)):
        cfg = 5
    if split_k == 0:        # auto: K slices only where the tile grid cannot fill the chip
        split_k = 1 if not splittable else lib.imd_conv_gemm_auto_split(M, N, K, cfg)
    if dkey is not None and len(_CFG_DECISIONS) < 4096:
        _CFG_DECISIONS[dkey] = (cfg, split_k)
    if res_rows:
        if cfg == 12 and res_rows % 128 == 0 and PERIODIC_RES:
            p.res_rows = res_rows
        else:
            res = repeat_batch(res.reshape(res_rows, N), M // res_rows)
            p.res = _dev(res, dt, "res")
    p.split_k = split_k
    if gn_in is not None:
        # Transformer2DModel.norm -> proj_in: inside the projection launch where that launch is a row-resident kernel and x came with its statistics
        gi_gamma, gi_beta, gi_eps, gi_silu, gi_groups = gn_in
        st = getattr(x, "_imd_gn_stats", None)
        fused = False
        if FUSED_GN_PROJ and cfg in FUSED_GN_PROJ_CFGS and st is not None and st[2] == gi_groups and FUSED_GN_STATS and st[0].shape[0] * Hout * Wout == M:
            p.gn_in_partial, p.gn_in_nparts, p.gn_in_groups = st[0].data_ptr(), st[1], int(gi_groups)
            p.gn_in_gamma, p.gn_in_beta = _dev(gi_gamma, torch.float32, "gn_in gamma"), _dev(gi_beta, torch.float32, "gn_in beta")
            p.gn_in_eps, p.gn_in_silu = float(gi_eps), int(bool(gi_silu))
            fused = bool(lib.imd_row_linear_gn_in_supported(C.byref(p), cfg))
            if not fused:
                p.gn_in_partial = None
                p.gn_in_nparts = p.gn_in_groups = 0
        if not fused:
            xv = x.view(M // (Hout * Wout), Hout * Wout, Cin)
            if st is not None:
                xv._imd_gn_stats = st
            xn = group_norm(xv, gi_gamma, gi_beta, groups=gi_groups, eps=gi_eps, silu=gi_silu)
            p.x = _dev(xn, dt, "x")
    if split_k > 1:
        p.splitk_ws = splitk_workspace(split_k * M * N, x.device).data_ptr()
        if SPLITK_IN_KERNEL:
            p.splitk_counters = splitk_counters(x.device).data_ptr()
    if gn_out is not None and FUSED_GN_FINISH and split_k > 1 and heads is None and not out_f32 and act == ACT_NONE and res is None and out_scale == 1.0:
        # GroupNorm (+ SiLU) of the OUTPUT inside the finish launch of the K slices (ResnetBlock2D: conv1 -> norm2 -> SiLU): the caller finds
        # `_imd_gn_applied` on the returned tensor and skips its own group_norm
        g_gamma, g_beta, g_eps, g_silu, g_groups = gn_out
        p.gn_out_gamma, p.gn_out_beta = _dev(g_gamma, torch.float32, "gn_out gamma"), _dev(g_beta, torch.float32, "gn_out beta")
        p.gn_out_eps, p.gn_out_silu, p.gn_out_groups = float(g_eps), int(bool(g_silu)), int(g_groups)
        if lib.imd_conv_gemm_gn_out_supported(C.byref(p)):
            p.splitk_counters = None
            gn_stats_groups = 0
            L.check(lib.imd_conv_gemm(C.byref(p), cfg, _stream()))
            out._imd_gn_applied = True
            return out
        p.gn_out_gamma = p.gn_out_beta = None
        p.gn_out_groups = 0
    stats = None
    # (round 6: the register-staged tiles 0..4 / 7 write them too -- conv_in, the stride-2 downsampler of the 64x64 level -- wherever a tile's rows lie in one image;
    #  GENERIC_GN_STATS is the A/B switch of that addition)
    if gn_stats_groups and FUSED_GN_STATS and heads is None and not out_f32 and act != ACT_GEGLU and \
            (cfg in (5, 22, 23, 29) or split_k > 1 or (GENERIC_GN_STATS and cfg in (-1, 0, 1, 2, 3, 4, 7))):
        p.gn_stats_groups = gn_stats_groups
        nparts = lib.imd_conv_gemm_stats_parts(C.byref(p), cfg)       # halo-patch epilogue (un-split) or the finish launch of the K slices
        if nparts > 0:
            Bimg = M // (Hout * Wout)
            stats = (torch.empty((Bimg, nparts, gn_stats_groups, 2), dtype=torch.float32, device=x.device), nparts, gn_stats_groups)
            p.gn_stats_out = stats[0].data_ptr()
        else:
            p.gn_stats_groups = 0
    if GEMM_EVENT_HOOK is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.imd_conv_gemm(C.byref(p), cfg, _stream()))
        e1.record()
        GEMM_EVENT_HOOK.setdefault((f"{M},{N},{K},{taps},{stride},{int(ups)}|{Hout}x{Wout}", cfg, split_k), []).append((e0, e1))
    else:
        L.check(lib.imd_conv_gemm(C.byref(p), cfg, _stream()))
    if stats is not None:
        out._imd_gn_stats = stats
    return out


_splitk_ws: Dict[Tuple, torch.Tensor] = {}


_splitk_cnt: Dict[Tuple, torch.Tensor] = {}


def splitk_counters(device) -> torch.Tensor:
    """Zeroed per-tile arrival counters of the in-kernel split-K reduction (include/imagdressing_hip.h::splitk_counters); every
    launch leaves them zero, launches are stream-ordered, so one array per device serves all of them."""
    key = (str(device), _stream())
    t = _splitk_cnt.get(key)
    if t is None:
        t = torch.zeros(16384, dtype=torch.int32, device=device)
        _splitk_cnt[key] = t
    return t


def splitk_workspace(nfloats: int, device) -> torch.Tensor:
    """Grow-only fp32 scratch for split-K partial tiles (stream-ordered reuse: one per device AND stream)."""
    key = (str(device), _stream())
    t = _splitk_ws.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(nfloats, 1 << 22), dtype=torch.float32, device=device)
        _splitk_ws[key] = t
    return t


def linear(x2d: torch.Tensor, w: torch.Tensor, bias=None, *, res=None, act=ACT_NONE, out_f32=False, out=None,
           out_ld=None, res_ld=None, cfg=-1, split_k=0, ln_eps=None) -> torch.Tensor:
    M, K = x2d.shape
    N = w.shape[0]
    return conv_gemm(x2d, w, M=M, N=N, Cin=K, bias=bias, res=res, act=act, out_f32=out_f32, out=out,
                     out_ld=out_ld, res_ld=res_ld, cfg=cfg, split_k=split_k, ln_eps=ln_eps)


def fold_layernorm_affine(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(W', b') with  LN_affine(x) @ W^T + b == LN_plain(x) @ W'^T + b':  W' = W diag(gamma) (rounded once to the weight dtype),
    b' = b + W beta (fp32).  Host-side, once per layer (``ln_eps`` of :func:`conv_gemm`)."""
    wf = w.float()
    w2 = (wf * gamma.float()[None, :]).to(w.dtype).contiguous()
    b2 = wf @ beta.float()
    if bias is not None:
        b2 = b2 + bias.float()
    return w2, b2.contiguous()


def pack_ff_fused(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, gamma: Optional[torch.Tensor] = None,
                  beta: Optional[torch.Tensor] = None, dtype=None):
    """Operands of :func:`ff_geglu_fused` from the diffusers-layout FeedForward parameters: ``w1`` [2*inner, C] / ``b1`` =
    ff.net.0.proj (rows [0, inner) = value, [inner, 2*inner) = gate: ``hidden, gate = proj(x).chunk(2, -1)``), ``w2`` [C, inner] /
    ``b2`` = ff.net.2; ``gamma`` / ``beta`` = norm3, folded into w1 / b1 (the kernel normalises without affine).  Layout: see
    include/imagdressing_hip.h::imd_ff_params.  Host-side, once per layer."""
    dtype = dtype or w1.dtype
    inner, C = w2.shape[1], w2.shape[0]
    w1f, b1f = w1.float(), b1.float()
    if gamma is not None:
        b1f = b1f + w1f @ beta.float()
        w1f = w1f * gamma.float()[None, :]
    dev = w1.device
    i = torch.arange(32, device=dev)
    jj = (i & 7) + 8 * (i >> 4)                                  # inner channel of packed row i inside its 16-block
    gate = ((i >> 3) & 1).bool()
    blk = torch.arange(inner // 16, device=dev)
    src = 16 * blk[:, None] + jj[None, :] + torch.where(gate, inner, 0)[None, :]          # [blocks, 32] rows of w1
    w1p = w1f[src.reshape(-1)].to(dtype).contiguous()            # [blocks * 32, C]
    b1p = b1f[src.reshape(-1)].contiguous()
    ks = torch.arange(16, device=dev)
    jk = torch.where(ks < 4, ks, torch.where(ks < 8, ks + 4, torch.where(ks < 12, ks - 4, ks)))     # slot -> inner channel in the 16-group
    cols = (16 * torch.arange(inner // 16, device=dev)[:, None] + jk[None, :]).reshape(inner // 32, 32)     # [chunks, 32]
    w2p = w2.float()[:, cols].permute(1, 0, 2).to(dtype).contiguous()                     # [chunks, C, 32]
    return dict(w1=w1p, b1=b1p, w2=w2p, b2=b2.float().contiguous(), ln=gamma is not None, C=C, inner=inner)


def ff_geglu_fused(x2d: torch.Tensor, packed: dict, ln_eps: float = 1e-5, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x + b2 + W2 geglu(W1 LN(x) + b1) in one launch (include/imagdressing_hip.h::imd_ff_geglu; C = 320, inner = 1280)."""
    ensure_device(x2d.device)
    M, C_ = x2d.shape
    dt = x2d.dtype
    if out is None:
        out = torch.empty((M, C_), dtype=dt, device=x2d.device)
    p = L.FfParams()
    p.x, p.w1, p.b1 = _dev(x2d, dt, "x"), _dev(packed["w1"], dt, "w1"), _dev(packed["b1"], torch.float32, "b1")
    p.w2, p.b2, p.out = _dev(packed["w2"], dt, "w2"), _dev(packed["b2"], torch.float32, "b2"), _dev(out, dt, "out")
    p.M, p.C, p.inner, p.x_ld, p.out_ld = M, packed["C"], packed["inner"], x2d.stride(0), out.stride(0)
    p.ln, p.ln_eps, p.dtype = int(packed["ln"]), float(ln_eps), _code(x2d, "x")
    _count("gemm_conv", 2.0 * M * packed["C"] * 2 * packed["inner"] + 2.0 * M * packed["inner"] * packed["C"])
    L.check(L.load().imd_ff_geglu(C.byref(p), _stream()))
    return out


def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, bias=None, *, taps=9, stride=1, ups=False, rowvec=None,
                rowvec_stride=0, rowvec_off=0, res=None, out_scale=1.0, act=ACT_NONE, out_f32=False, cfg=-1, split_k=0, gn=None,
                pad_br_only=False, gn_stats_groups=0, gn_out=None, gn_in=None) -> torch.Tensor:
    """x [B, H, W, Cin] bf16 -> [B, Ho, Wo, Cout].  ``pad_br_only``: F.pad(x, (0, 1, 0, 1)) + conv(padding=0) (VAE encoder)."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = ((Hl + stride - 1) // stride, (Wl + stride - 1) // stride)
    M = B * Ho * Wo
    out = conv_gemm(x, w, M=M, N=Cout, Cin=Cin, taps=taps, Hin=H, Win=W, Hout=Ho, Wout=Wo, stride=stride, ups=ups,
                    bias=bias, rowvec=rowvec, rowvec_stride=rowvec_stride, rowvec_off=rowvec_off, res=res, out_scale=out_scale, act=act,
                    out_f32=out_f32, cfg=cfg, split_k=split_k, gn=gn, pad_br_only=pad_br_only, gn_stats_groups=gn_stats_groups, gn_out=gn_out, gn_in=gn_in)
    r = out.view(B, Ho, Wo, -1)
    if getattr(out, "_imd_gn_applied", False):
        r._imd_gn_applied = True
    st = getattr(out, "_imd_gn_stats", None)
    if st is not None:
        r._imd_gn_stats = st          # (a view is a new tensor object: carry the producer's GroupNorm statistics over)
    return r


# bench.py installs {"match": fn(**shape) -> bool, "events": []} to bracket matching launches with HIP
# events on the launch stream (roofline measurement inside the timed region); None = no overhead.
ATTN_EVENT_HOOK = None


# the out-projection of the 64x64-level attention layers inside the attention launch (ABI v7, attention_d40.hip PROJ): the level-0
# hybrid block = two launches (norm1 + q/k/v, attention + out-projection + residual).  OPT-IN (IMD_FUSED_OUT_PROJ=1): correct and
# tested, but measured 7.5 % SLOWER end to end than the three-launch form (profiles/r3ao_*, r3ap_*)
FUSED_OUT_PROJ = _os.environ.get("IMD_FUSED_OUT_PROJ", "0") == "1"


def attention_proj_supported(H: int, N: int, D: int) -> bool:
    """Can imd_attention carry the block's out-projection (ABI v7)?  Head dim 40, 8 heads, N >= 512: the 64x64-level blocks."""
    return D == 40 and H * D == 320 and N >= 512


_proj_cnt: Dict[Tuple, torch.Tensor] = {}


def proj_counters(n: int, device) -> torch.Tensor:
    """Zeroed arrival counters of the fused out-projection (one per batch entry and 256-row block); every launch leaves them zero."""
    key = (str(device), _stream())
    t = _proj_cnt.get(key)
    if t is None or t.numel() < n:
        t = torch.zeros(max(n, 4096), dtype=torch.int32, device=device)
        _proj_cnt[key] = t
    return t


def attention_dup_supported(H: int, N: int, D: int) -> bool:
    """Can imd_attention also store the first-phase result to ``out_dup`` (ABI v9)?  Head dim 40, N >= 512: the 64x64-level blocks."""
    return bool(L.load().imd_attention_dup_supported(H, N, D))


# (round 6) phase-split launch of the hybrid attention at the 32x32 / 16x16 / 8x8 levels (imd_attn_params.phase2_rows; A/B switch)
ATTN_PHASE_SPLIT = _os.environ.get("IMD_ATTN_PHASE_SPLIT", "1") != "0"
ATTN_PHASE_SPLIT_MAX_N = int(_os.environ.get("IMD_ATTN_PHASE_SPLIT_MAX_N", "512"))      # (kernel alone, tools/attn_bench.py --phase-split: N = 256: 29.6 -> 19.8 us, N = 64: 15.5 -> 11.3 us, N = 1024: 60.3 -> 64.0 us)


@functools.lru_cache(maxsize=None)
def attention_phase_split_supported(D: int) -> bool:
    return bool(L.load().imd_attention_phase_split_supported(D))


def attention(q, k1, v1t, out, *, B, H, N, D, L1, L1P, kv1_bdiv=1, k2=None, v2t=None, scale2=None,
              L2=0, L2P=0, kv2_bdiv=1, out_ld=None, causal=False, k_pad_one=False, proj=None, out_dup=None, phase2_rows=0):
    """``k_pad_one``: k1 (and k2) came from :func:`k_buffer`, i.e. their pad column D holds 1.0 (see the header).
    ``phase2_rows`` = R: the caller guarantees that exactly the rows [0, R) have a non-zero ``scale2`` (the cond half of a CFG batch); where the library
    takes it (:func:`attention_phase_split_supported`, switch ``ATTN_PHASE_SPLIT``) the two softmaxes of those rows run as separate workgroups of ONE launch
    and a follow-up elementwise launch adds them -- bit-identical to the one-workgroup form, half as long per workgroup.  Ignored elsewhere.
    ``out_dup`` [B, N, C]: also receives softmax(Q K1^T) V1 of every batch entry (the paired uncond rows of a CFG batch's first hybrid
    block, :func:`attention_dup_supported`).
    ``proj`` = (w [C, C], bias [C] fp32 | None, residual [B, N, C] | None, proj_out [B, N, C]): the out-projection fused into the
    launch (:func:`attention_proj_supported`); returns proj_out then."""
    ensure_device(q.device)
    p = L.AttnParams()
    dt = q.dtype
    p.dtype = _code(q, "q")
    p.q, p.k1, p.v1t = _dev(q, dt, "q"), _dev(k1, dt, "k1"), _dev(v1t, dt, "v1t")
    p.k2, p.v2t = _opt(k2, dt, "k2"), _opt(v2t, dt, "v2t")
    p.scale2 = _opt(scale2, torch.float32, "scale2")
    p.out = _dev(out, dt, "out")
    p.B, p.H, p.N, p.D = B, H, N, D
    p.L1, p.L1P, p.kv1_bdiv = L1, L1P, kv1_bdiv
    p.L2, p.L2P, p.kv2_bdiv = L2, L2P, kv2_bdiv
    p.out_ld = H * D if out_ld is None else out_ld
    p.causal = int(causal)
    p.k_pad_one = int(bool(k_pad_one))
    p.flags = _attn_call_flags()
    if out_dup is not None:
        if out_dup.numel() != out.numel():
            raise L.ImdError("attention: out_dup must have the shape of out")
        p.out_dup = _dev(out_dup, dt, "out_dup")
    if phase2_rows and ATTN_PHASE_SPLIT and k2 is not None and scale2 is not None and proj is None and out_dup is None and not causal \
            and 0 < phase2_rows <= B and attention_phase_split_supported(D) and N <= ATTN_PHASE_SPLIT_MAX_N:
        p.phase2_rows = int(phase2_rows)
        p.phase2_out = workspace("attn_phase2", (phase2_rows * N * H * D,), torch.float32, q.device).data_ptr()
    ret = out
    if proj is not None:
        pw, pb, pres, pout = proj
        Cc = H * D
        if pw.numel() != Cc * Cc or pout.numel() != B * N * Cc or (pres is not None and pres.numel() != B * N * Cc):
            raise L.ImdError(f"attention: fused out-projection operands do not match B={B} N={N} C={Cc}")
        p.proj_w, p.proj_b = _dev(pw, dt, "proj_w"), _opt(pb, torch.float32, "proj_b")
        p.proj_res, p.proj_out = _opt(pres, dt, "proj_res"), _dev(pout, dt, "proj_out")
        p.proj_res_ld = p.proj_out_ld = Cc
        p.proj_counters = proj_counters(B * ((N + 255) // 256), q.device).data_ptr()
        ret = pout
        _count("gemm", 2.0 * B * N * Cc * Cc)
    if FLOP_COUNTER is not None:          # (reads scale2 back: counting mode only)
        rows2 = 0 if (k2 is None or scale2 is None) else int((scale2 != 0).sum().item())
        _count("attention", 4.0 * H * N * D * (B * L1 * (0.5 if causal else 1.0) + rows2 * L2))
    hook = ATTN_EVENT_HOOK
    if hook is not None and hook["match"](B=B, H=H, N=N, D=D, L1=L1, L2=L2 if k2 is not None else 0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(L.load().imd_attention(C.byref(p), _stream()))
        e1.record()
        hook["events"].append((e0, e1))
        return ret
    L.check(L.load().imd_attention(C.byref(p), _stream()))
    return ret


# ---------------------------------------------------------------------------------------------------------------------
# MX-FP8 attention (BASELINE.json configs[4]: "... with fp8 MFMA attention"): opt-in, head dim 40 (UNet level 0) only
# ---------------------------------------------------------------------------------------------------------------------
ATTN_FP8 = False                       # processors route d = 40 attention through imd_attention_fp8 when set
FP8_EXPS = dict(q=4, k=2, v=3)         # operands are stored as value * 2^e in e4m3 (range 2^-9 .. 448); eq + ek <= 8


def quantize_fp8_rows(x: torch.Tensor, exp: int, pad_val: float = 0.0) -> torch.Tensor:
    """Q or K [.., L, 48] 16-bit -> e4m3 bytes [.., L, 64] (columns 0..39 * 2^exp, columns 40 / 41 = pad_val)."""
    ensure_device(x.device)
    if x.shape[-1] != 48:
        raise L.ImdError(f"quantize_fp8_rows: expected head-dim-40 rows padded to 48, got {tuple(x.shape)}")
    rows = x.numel() // 48
    out = torch.empty(x.shape[:-1] + (64,), dtype=torch.uint8, device=x.device)
    L.check(L.load().imd_attn_quantize_fp8(_dev(x, x.dtype, "x"), out.data_ptr(), 0, rows, 0, exp, float(pad_val), _code(x, "x"), _stream()))
    return out


def quantize_fp8_vt(vt: torch.Tensor, exp: int) -> torch.Tensor:
    """V^T [.., 64, LP] 16-bit -> e4m3 bytes [.., 64, LP] (rows 0..39), keys permuted inside 64-groups as the kernel expects."""
    ensure_device(vt.device)
    if vt.shape[-2] != 64 or vt.shape[-1] % 64:
        raise L.ImdError(f"quantize_fp8_vt: expected [.., 64, LP] with LP % 64 == 0, got {tuple(vt.shape)}")
    LP = vt.shape[-1]
    groups = vt.numel() // (64 * LP)
    out = torch.zeros(vt.shape, dtype=torch.uint8, device=vt.device)
    L.check(L.load().imd_attn_quantize_fp8(_dev(vt, vt.dtype, "vt"), out.data_ptr(), 1, groups, LP, exp, 0.0, _code(vt, "vt"), _stream()))
    return out


def attention_fp8(q8, k1, v1t, out, *, B, H, N, L1, L1P, kv1_bdiv=1, k2=None, v2t=None, scale2=None, L2=0, L2P=0, kv2_bdiv=1,
                  out_ld=None, exps=None):
    """imd_attention_fp8 on e4m3 operands from quantize_fp8_rows / quantize_fp8_vt (head dim 40); ``out`` is 16-bit."""
    ensure_device(out.device)
    e = dict(FP8_EXPS, **(exps or {}))
    u8 = torch.uint8
    p = L.AttnParams()
    p.dtype = _code(out, "out")
    p.q, p.k1, p.v1t = _dev(q8, u8, "q8"), _dev(k1, u8, "k1"), _dev(v1t, u8, "v1t")
    p.k2, p.v2t = _opt(k2, u8, "k2"), _opt(v2t, u8, "v2t")
    p.scale2 = _opt(scale2, torch.float32, "scale2")
    p.out = _dev(out, out.dtype, "out")
    p.B, p.H, p.N, p.D = B, H, N, 40
    p.L1, p.L1P, p.kv1_bdiv = L1, L1P, kv1_bdiv
    p.L2, p.L2P, p.kv2_bdiv = L2, L2P, kv2_bdiv
    p.out_ld = H * 40 if out_ld is None else out_ld
    L.check(L.load().imd_attention_fp8(C.byref(p), e["q"], e["k"], e["v"], _stream()))
    return out


def group_norm(x: torch.Tensor, gamma, beta, *, groups=32, eps=1e-5, silu=False, out=None) -> torch.Tensor:
    """x [B, HW, C] (or [B, H, W, C]) bf16 NHWC."""
    ensure_device(x.device)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    if out is None:
        out = torch.empty_like(x)
    lib = L.load()
    nws = lib.imd_groupnorm_workspace_floats(B, HW, Cc, groups)
    part = workspace("gn_partial", (max(nws, 1),), torch.float32, x.device)
    p = L.GroupNormParams()
    p.dtype = _code(x, "x")
    p.x, p.y = _dev(x, x.dtype, "x"), _dev(out, x.dtype, "out")
    p.gamma, p.beta = _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta")
    p.partial = part.data_ptr()
    p.B, p.HW, p.C, p.G, p.x_ld, p.y_ld = B, HW, Cc, groups, Cc, Cc
    p.eps, p.silu = eps, int(silu)
    st = getattr(x, "_imd_gn_stats", None)        # statistics written by the epilogue of the convolution that produced x
    if st is not None and st[2] == groups and st[0].shape[0] == B and FUSED_GN_STATS:
        p.partial, p.nparts = st[0].data_ptr(), st[1]
    L.check(lib.imd_groupnorm(C.byref(p), _stream()))
    return out


def group_norm_coeffs(x: torch.Tensor, gamma, beta, *, groups=32, eps=1e-5):
    """GroupNorm statistics of x [B, HW, C] as per-(batch, channel) fp32 (a, b) with y = x*a + b: the operand of
    conv_gemm's fused prologue (``gn=(a, b, silu)``)."""
    ensure_device(x.device)
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    lib = L.load()
    nws = lib.imd_groupnorm_workspace_floats(B, HW, Cc, groups)
    part = workspace("gn_partial", (max(nws, 1),), torch.float32, x.device)
    ab = torch.empty((2, B, Cc), dtype=torch.float32, device=x.device)
    p = L.GroupNormParams()
    p.dtype = _code(x, "x")
    p.x, p.y = _dev(x, x.dtype, "x"), None
    p.gamma, p.beta = _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta")
    p.partial = part.data_ptr()
    p.B, p.HW, p.C, p.G, p.x_ld, p.y_ld = B, HW, Cc, groups, Cc, Cc
    p.eps, p.silu = eps, 0
    st = getattr(x, "_imd_gn_stats", None)        # statistics written by the producer of x (see group_norm): no statistics pass
    if st is not None and st[2] == groups and st[0].shape[0] == B and FUSED_GN_STATS:
        p.partial, p.nparts = st[0].data_ptr(), st[1]
    L.check(lib.imd_groupnorm_coeffs(C.byref(p), ab[0].data_ptr(), ab[1].data_ptr(), _stream()))
    return ab[0], ab[1]


def lincomb(terms, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_j c_j * x_j over 1..8 (coefficient, fp32 tensor) pairs of equal numel; ``out`` may be one of the inputs."""
    xs = [t for _, t in terms]
    ensure_device(xs[0].device)
    n, numel = len(terms), xs[0].numel()
    for t in xs:
        if t.numel() != numel:
            raise L.ImdError("lincomb: all tensors must have the same number of elements")
    if out is None:
        out = torch.empty_like(xs[0])
    ptrs = (C.c_void_p * n)(*[_dev(t, torch.float32, "lincomb input") for t in xs])
    coefs = (C.c_float * n)(*[float(c) for c, _ in terms])
    L.check(L.load().imd_lincomb(ptrs, coefs, n, _dev(out, torch.float32, "out"), numel, _stream()))
    return out


def embed_tokens(table: torch.Tensor, pos: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """table [V, C], pos [T, C] 16-bit, ids [B, T] int64 -> [B, T, C] = table[ids] + pos."""
    ensure_device(table.device)
    B, T = ids.shape
    V, Cc = table.shape
    out = torch.empty((B, T, Cc), dtype=table.dtype, device=table.device)
    L.check(L.load().imd_embed_tokens(_dev(table, table.dtype, "table"), V, _dev(pos, table.dtype, "pos"), pos.shape[0],
                                      _dev(ids, torch.int64, "ids"), out.data_ptr(), B * T, Cc, _code(table, "table"), _stream()))
    return out


def vit_assemble(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """patches [B, P, C], cls [C], pos [P + 1, C] -> [B, P + 1, C] (class token first, position embeddings added)."""
    ensure_device(patches.device)
    B, P, Cc = patches.shape
    out = torch.empty((B, P + 1, Cc), dtype=patches.dtype, device=patches.device)
    L.check(L.load().imd_vit_assemble(_dev(patches, patches.dtype, "patches"), _dev(cls, patches.dtype, "cls"), _dev(pos, patches.dtype, "pos"),
                                      out.data_ptr(), B, P, Cc, _code(patches, "patches"), _stream()))
    return out


def softmax_rows(s: torch.Tensor, dtype=bf16, out=None) -> torch.Tensor:
    """Row softmax of an fp32 matrix [rows, cols] -> 16-bit probabilities (cols <= 16384).  ``out`` may have wider rows
    (only the first ``cols`` columns of each row are written)."""
    ensure_device(s.device)
    rows, cols = s.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=dtype, device=s.device)
    elif out.shape[0] != rows or out.shape[1] < cols:
        raise L.ImdError(f"softmax_rows: out {tuple(out.shape)} does not hold a [{rows}, {cols}] matrix")
    L.check(L.load().imd_softmax_rows(_dev(s, torch.float32, "s"), s.stride(0), _dev(out, out.dtype, "out"), out.stride(0),
                                      rows, cols, _code(out, "out"), _stream()))
    return out


def layer_norm(x: torch.Tensor, gamma, beta, eps=1e-5, out=None) -> torch.Tensor:
    ensure_device(x.device)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    p = L.LayerNormParams()
    p.dtype = _code(x, "x")
    p.x, p.y = _dev(x, x.dtype, "x"), _dev(out, x.dtype, "out")
    p.gamma, p.beta = _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta")
    p.rows, p.C, p.x_ld, p.y_ld, p.eps = rows, Cc, Cc, Cc, eps
    L.check(L.load().imd_layernorm(C.byref(p), _stream()))
    return out


def ddim_coefs(a_t, a_prev, a_next=None):
    """The six schedule coefficients of one DDIM step in the order imd_ddim_params.coefs reads them."""
    an = (1.0, 0.0) if a_next is None else (a_next ** 0.5, (1 - a_next) ** 0.5)
    return [a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5, an[0], an[1]]


def ddim_cfg_step(z, eps, x_next, *, guidance, a_t=1.0, a_prev=1.0, mask=None, z_img=None, noise=None, a_next=None, coefs=None,
                  var_noise=None, sigma=0.0):
    """z [B,HW,4] fp32 (in place); eps [2B,HW,4] fp32; x_next [2B,HW,8] bf16 or None.  ``coefs``: device fp32 [6]
    (:func:`ddim_coefs`) read by the kernel instead of a_t / a_prev / a_next (HIP-graph replay of a step).
    ``var_noise`` [B,HW,4] fp32 + ``sigma``: the stochastic DDIM step (eta > 0) -- diffusers' ``DDIMScheduler.step``:
    direction coefficient sqrt(1 - a_prev - sigma^2), ``+ sigma * var_noise``."""
    ensure_device(z.device)
    B, HW = z.shape[0], z.shape[1]
    p = L.DdimParams()
    p.z, p.eps = _dev(z, torch.float32, "z"), _dev(eps, torch.float32, "eps")
    p.dtype = 0 if x_next is None else _code(x_next, "x_next")
    p.x_next = None if x_next is None else _dev(x_next, x_next.dtype, "x_next")
    p.B, p.HW = B, HW
    p.guidance = guidance
    p.sqrt_a_t, p.sqrt_1m_a_t = a_t ** 0.5, (1 - a_t) ** 0.5
    p.sqrt_a_prev, p.sqrt_1m_a_prev = a_prev ** 0.5, (1 - a_prev) ** 0.5
    p.var_noise, p.sigma = None, 0.0
    if var_noise is not None:
        if var_noise.numel() != B * HW * 4:
            raise L.ImdError(f"ddim_cfg_step: var_noise has {var_noise.numel()} elements, expected {B * HW * 4}")
        p.var_noise, p.sigma = _dev(var_noise, torch.float32, "var_noise"), float(sigma)
        p.sqrt_1m_a_prev = max(1 - a_prev - float(sigma) ** 2, 0.0) ** 0.5
    p.mask = _opt(mask, torch.float32, "mask")
    p.z_img = _opt(z_img, torch.float32, "z_img")
    p.noise = _opt(noise, torch.float32, "noise")
    if a_next is None:
        p.sqrt_a_next, p.sqrt_1m_a_next = 1.0, 0.0
    else:
        p.sqrt_a_next, p.sqrt_1m_a_next = a_next ** 0.5, (1 - a_next) ** 0.5
    p.coefs = _opt(coefs, torch.float32, "coefs")
    if coefs is not None and coefs.numel() < 6:
        raise L.ImdError("ddim_cfg_step: coefs needs 6 fp32 values")
    L.check(L.load().imd_ddim_cfg_step(C.byref(p), _stream()))
    return z


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    ensure_device(t.device)
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    L.check(L.load().imd_timestep_embedding(_dev(t, torch.float32, "t"), out.data_ptr(), t.shape[0], dim, _stream()))
    return out


def add(a: torch.Tensor, b: torch.Tensor, b_scale: float = 1.0, out=None) -> torch.Tensor:
    ensure_device(a.device)
    Cc = a.shape[-1]
    rows = a.numel() // Cc
    if out is None:
        out = torch.empty_like(a)
    dt = a.dtype
    L.check(L.load().imd_add(_dev(a, dt, "a"), Cc, _dev(b, dt, "b"), Cc, _dev(out, dt, "out"), Cc, rows, Cc,
                             b_scale, _code(a, "a"), _stream()))
    return out


def concat_channels(a: torch.Tensor, b: torch.Tensor, b_add: Optional[torch.Tensor] = None, gn_stats_groups: int = 0) -> torch.Tensor:
    """cat([a, b (+ b_add)], channel) for NHWC tensors [..., Ca] and [..., Cb].

    ``gn_stats_groups`` = G (4-D operands [B, H, W, C]): the launch also writes the GroupNorm(G) statistics of its output, which ride on the
    returned tensor (``_imd_gn_stats``) exactly as a convolution's do -- the :func:`group_norm` behind an up block's concatenation then
    normalises only.  Same partials as the statistics launch would write (same chunking and order): bit-identical either way."""
    ensure_device(a.device)
    Ca, Cb = a.shape[-1], b.shape[-1]
    rows = a.numel() // Ca
    dt = a.dtype
    out = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=dt, device=a.device)
    lib = L.load()
    b_rows = b.numel() // Cb            # b may hold HALF the rows: one skip tensor for both (identical) halves of a CFG batch
    if b_rows * Cb != b.numel() or b_rows == 0 or rows % b_rows or (b_add is not None and b_add.numel() != rows * Cb):
        raise L.ImdError(f"concat_channels: operands disagree on the row count ({rows} rows of {Ca} + {Cb} channels, b has {b_rows})")
    G = gn_stats_groups
    if G and FUSED_CONCAT_STATS and FUSED_GN_STATS and a.dim() == 4 and b.dim() == 4 and (Ca + Cb) % G == 0:
        B = a.shape[0]
        HW = rows // B
        cpg = (Ca + Cb) // G
        nparts = lib.imd_groupnorm_parts(B, HW, Ca + Cb)
        if nparts > 0 and b_rows % HW == 0 and G <= 64 and (cpg >= 8 or cpg == 4):
            part = torch.empty((B, nparts, G, 2), dtype=torch.float32, device=a.device)
            L.check(lib.imd_concat2_gn_stats(_dev(a, dt, "a"), Ca, _dev(b, dt, "b"), Cb, _opt(b_add, dt, "b_add"), out.data_ptr(), B, HW, b_rows // HW, G,
                                             part.data_ptr(), _code(a, "a"), _stream()))
            out._imd_gn_stats = (part, nparts, G)
            return out
    L.check(lib.imd_concat2(_dev(a, dt, "a"), Ca, _dev(b, dt, "b"), Cb, _opt(b_add, dt, "b_add"), out.data_ptr(), rows, b_rows, _code(a, "a"), _stream()))
    return out


def repeat_batch(x: torch.Tensor, times: int = 2) -> torch.Tensor:
    """cat([x] * times, dim=0) for a contiguous NHWC tensor (strided 2-D copies)."""
    ensure_device(x.device)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    out = torch.empty((x.shape[0] * times,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    src = _dev(x, x.dtype, "x")
    for i in range(times):
        L.check(L.load().imd_copy2d(src, Cc, out.data_ptr() + 2 * i * rows * Cc, Cc, rows, Cc, _stream()))
    return out


def f32_to_16(a: torch.Tensor, dtype=bf16) -> torch.Tensor:
    ensure_device(a.device)
    out = torch.empty(a.shape, dtype=dtype, device=a.device)
    L.check(L.load().imd_f32_to_16(_dev(a, torch.float32, "a"), out.data_ptr(), a.numel(), DTYPE_CODE[dtype], _stream()))
    return out


def concat_tokens(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat([a, b], dim=1) for [B, La, C] and [B, Lb, C] (one strided 2-D copy per operand)."""
    ensure_device(a.device)
    B, La, Cc = a.shape
    Lb = b.shape[1]
    dt = a.dtype
    _code(a, "a")
    out = torch.empty(B, La + Lb, Cc, dtype=dt, device=a.device)
    lib = L.load()
    ld = (La + Lb) * Cc
    L.check(lib.imd_copy2d(_dev(a, dt, "a"), La * Cc, out.data_ptr(), ld, B, La * Cc, _stream()))
    L.check(lib.imd_copy2d(_dev(b, dt, "b"), Lb * Cc, out.data_ptr() + 2 * La * Cc, ld, B, Lb * Cc, _stream()))
    return out
