"""Attention processors of IMAGDressing-v1 on the fused HIP path (the plugin surface).

Same class names, constructor signatures, ``state_dict`` keys and mutable attributes
(``.scale``, ``.lora_scale``, ``.name``, ``.cache["hidden_states"]``) as
``/root/reference/adapter/attention_processor.py`` so that
``unet.set_attn_processor({...})``, ``ModuleList(unet.attn_processors.values()).load_state_dict``
(inference_IMAGdressing.py:85-87,117) and ``pipe.set_scale`` keep working, and the same call
protocol ``proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
scale=1.0, **cross_attention_kwargs)``.

What differs is *how* a call executes.  Per attention layer the reference launches 6 GEMMs, 6
reshapes, 2 SDPAs and an add (attention_processor.py:568-617); here it is three launches:

  1. one GEMM whose epilogue writes Q (pre-scaled by d^-1/2 log2 e), K and V^T straight into the
     per-head layouts the attention kernel wants,
  2. one fused attention kernel running BOTH softmaxes (self + garment, or text + IP tokens) and
     summing them before anything leaves registers,
  3. the out-projection GEMM with bias and the transformer block's residual add in its epilogue.

Step-invariant operands are projected once and cached on the processor: the garment K_ref/V_ref
(the reference re-projects them at every step, :600-601), the text / IP-token K/V, and LoRA deltas,
which are folded into effective weights W + lambda * up @ down whenever ``lora_scale`` changes.

Extensions over the reference (all optional, all default to reference behaviour):
  * ``sa_batch_mask`` (cross_attention_kwargs): float tensor [B]; the garment branch is applied to
    batch row b with weight ``scale * sa_batch_mask[b]``.  Lets cond and uncond rows of a CFG batch
    share one UNet call (the reference issues two B=1 calls, IMAGDressing_v1_pipeline.py:499-518).
  * garment tokens [1, M, C] broadcast over the batch (the reference's ``view(batch_size, ...)`` at
    :602-603 is only defined for B == 1).
  * ``encoder_hidden_states`` with fewer rows than ``hidden_states`` is shared by contiguous groups
    of batch rows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops

bf16 = torch.bfloat16
LOG2E = math.log2(math.e)


class LoRALinearLayer(nn.Module):
    """Parameter container with the diffusers-0.24 ``LoRALinearLayer`` layout (down / up)."""

    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha = network_alpha
        self.rank = rank
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def delta(self) -> torch.Tensor:
        """up @ down (* alpha / rank) in fp32 -- the matrix added to the frozen weight."""
        d = self.up.weight.float() @ self.down.weight.float()
        if self.network_alpha is not None:
            d = d * (self.network_alpha / self.rank)
        return d


def _version_key(*tensors) -> Tuple:
    return tuple((t.data_ptr(), t._version, t.device, t.dtype) for t in tensors)


class _TensorCache:
    """Small LRU of values cached against the *identity and version* of their source tensors (which the
    cache keeps alive, so an address can never be recycled under it).  A few entries, because the
    reference-style loop alternates two conditionings per step (cond / uncond UNet calls)."""

    def __init__(self, capacity: int = 4):
        self._entries = []          # [(srcs, key, value)], most recent last
        self._cap = capacity

    def get(self, srcs, extra=()):
        key = (_version_key(*srcs), extra)
        for i, (s, k, v) in enumerate(self._entries):
            if k == key and len(s) == len(srcs) and all(a is b for a, b in zip(s, srcs)):
                if i != len(self._entries) - 1:
                    self._entries.append(self._entries.pop(i))
                return v
        return None

    def put(self, srcs, value, extra=()):
        self._entries.append((tuple(srcs), (_version_key(*srcs), extra), value))
        if len(self._entries) > self._cap:
            self._entries.pop(0)
        return value


_W_PARTS = {"qkv": ("to_q", "to_k", "to_v"), "kv": ("to_k", "to_v"), "q": ("to_q",), "o": ("to_out",), "bo": ("to_out",)}
_16BIT = (torch.bfloat16, torch.float16)


def _compute_dtype(attn, hidden_states: torch.Tensor, attention_mask=None):
    """Also the one place that refuses the module options the reference processors honour
    (attention_processor.py:545-566, :619-625) but SD1.5 never sets -- refusing beats silently diverging.

    16-bit element type a layer runs in: that of its projection weights (the engines' own ``Attention`` and a
    diffusers ``Attention`` after ``.to(dtype=torch.float16)``, inference_IMAGdressing.py:50-52), else that of the
    activations, else fp16 (the reference's GPU dtype)."""
    if attention_mask is not None:
        raise NotImplementedError("attention_mask is always None on the IMAGDressing path (SURVEY appendix 3); masks are not implemented")
    if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None \
            or getattr(attn, "norm_cross", None):
        raise NotImplementedError("attn.spatial_norm / group_norm / norm_cross are None for every SD1.5 attention layer; "
                                  "modules that set them are not supported by the fused processors")
    wd = attn.to_q.weight.dtype
    if wd in _16BIT:
        return wd
    return hidden_states.dtype if hidden_states.dtype in _16BIT else torch.float16


def _layer_weights(attn, which: str, dtype, device) -> torch.Tensor:
    """Projection weights of ``attn`` read ONLY through the attribute surface a diffusers ``Attention`` has
    (attention_processor.py:545-625: ``to_q`` / ``to_k`` / ``to_v`` / ``to_out[0]``): 'qkv' [3C, C] and 'kv' [2C, Kd] are
    concatenated once, 'q' / 'o' are passed through (cast once if the module is not 16-bit / not on the device), 'bo' is
    the out-projection bias as fp32 (or None).  Cached on the module, keyed by the identity and version of the source
    parameters, so ``load_state_dict`` / ``.to()`` / in-place edits rebuild it."""
    mods = [attn.to_out[0] if n == "to_out" else getattr(attn, n) for n in _W_PARTS[which]]
    srcs = [m.bias if which == "bo" else m.weight for m in mods]
    if srcs[0] is None:
        return None
    want = torch.float32 if which == "bo" else dtype
    if len(srcs) == 1 and srcs[0].dtype == want and srcs[0].device == device and srcs[0].is_contiguous():
        return srcs[0].detach()
    cache = attn.__dict__.setdefault("_imd_wcache", {})
    key = (_version_key(*srcs), want, str(device))
    ent = cache.get(which)
    if ent is None or ent[0] != key or not all(a is b for a, b in zip(ent[1], srcs)):
        t = torch.cat([t.detach().to(device=device, dtype=want) for t in srcs], 0).contiguous()
        ent = cache[which] = (key, tuple(srcs), t)
    return ent[2]


def _ln_folded_q(attn, gamma: torch.Tensor, beta: torch.Tensor, dtype, device, which: str = "q"):
    """(W diag(gamma), W beta) for the fused ``LayerNorm -> to_q`` (which = "q") / ``LayerNorm -> to_q, to_k, to_v`` ("qkv")
    launch (ops.conv_gemm ``ln_eps``), cached on the module against the identity / version of the projection weights and the
    two norm parameters."""
    mods = [getattr(attn, n) for n in _W_PARTS[which]]
    srcs = tuple(m.weight for m in mods) + (gamma, beta)
    cache = attn.__dict__.setdefault("_imd_wcache", {})
    key = (_version_key(*srcs), dtype, str(device))
    ent = cache.get(which + "_ln")
    if ent is None or ent[0] != key or not all(a is b for a, b in zip(ent[1], srcs)):
        w = torch.cat([m.weight.detach().to(device=device, dtype=dtype) for m in mods], 0)
        ent = cache[which + "_ln"] = (key, srcs, ops.fold_layernorm_affine(w, None, gamma.to(device), beta.to(device)))
    return ent[2]


# fused ``norm1 -> to_q / to_k / to_v`` (csrc/row_qkv.hip): 320 channels, token count per image a multiple of 128
def _qkv_ln_ok(x: torch.Tensor) -> bool:
    return x.shape[-1] == 320 and x.shape[1] % 128 == 0 and ops.FUSED_LN


# channel counts the row-resident ``LayerNorm -> linear`` kernels exist for (csrc/row_linear.hip, row_linear_k640.hip: the 64x64 and
# 32x32 levels of SD1.5)
FUSED_LN_CHANNELS = (320, 640, 1280)


def _as_tokens(hidden_states: torch.Tensor, dtype):
    """Accept [B, N, C] (transformer blocks) or [B, C, H, W] (attention_processor.py:548-552); cast to the
    16-bit element type of the layer's packed weights (bf16 or fp16)."""
    shape4 = None
    if hidden_states.dim() == 4:
        b, c, h, w = hidden_states.shape
        shape4 = (b, c, h, w)
        hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
    if not hidden_states.is_cuda:
        from .._lib import ImdError
        raise ImdError("imagdressing_amd processors run on MI355X only: hidden_states is on " + str(hidden_states.device))
    return hidden_states.to(dtype).contiguous(), shape4


def _restore(out, shape4, like):
    if shape4 is not None:
        b, c, h, w = shape4
        out = out.transpose(-1, -2).reshape(b, c, h, w)
    return out


def _project_kv(tokens: torch.Tensor, wkv: torch.Tensor, heads: int):
    """tokens [Bk, L, Kd] bf16, wkv [2C, Kd] -> (K [Bk,H,L,dpk], V^T [Bk,H,dpv,LP], L, LP) in freshly
    zeroed persistent buffers (their padding must stay zero)."""
    Bk, Lk, Kd = tokens.shape
    Cc = wkv.shape[0] // 2
    D = Cc // heads
    dpk, dpv = ops.attn_padded_dims(D)
    LP = ops.pad64(Lk)
    k = ops.k_buffer((Bk, heads, Lk, dpk), D, wkv.dtype, tokens.device)
    vt = torch.zeros(Bk, heads, dpv, LP, dtype=wkv.dtype, device=tokens.device)
    ops.conv_gemm(tokens.view(Bk * Lk, Kd), wkv, M=Bk * Lk, N=2 * Cc, Cin=Kd, Hin=Lk, Win=1, Hout=Lk, Wout=1,
                  heads=dict(C=Cc, H=heads, D=D, dests=[(k, 0, dpk, Lk, 1.0), (vt, 1, dpv, LP, 1.0)]))
    return k, vt, Lk, LP


_FP8_KV: Dict[int, tuple] = {}          # (dropped by ops.clear_workspaces(): it pins the 16-bit K and the e4m3 K / V^T of every layer)
ops._clear_hooks.append(_FP8_KV.clear)


def _fp8_kv(kv, cache: bool):
    """e4m3 versions of a (K, V^T, L, LP) pair; step-invariant pairs (garment / text K/V) are quantised once (the cache holds
    the 16-bit source too, so an address cannot be recycled under it)."""
    k, vt = kv[0], kv[1]
    e = ops.FP8_EXPS
    if cache:
        hit = _FP8_KV.get(id(k))
        if hit is not None and hit[0] is k and hit[1] == (e["q"], e["k"], e["v"]):
            return hit[2], hit[3]
    k8 = ops.quantize_fp8_rows(k, e["k"], pad_val=float(2 ** (e["q"] + e["k"])))
    v8 = ops.quantize_fp8_vt(vt, e["v"])
    if cache:
        if len(_FP8_KV) > 64:
            _FP8_KV.clear()
        _FP8_KV[id(k)] = (k, (e["q"], e["k"], e["v"]), k8, v8)
    return k8, v8


def _fused_attention(x: torch.Tensor, heads: int, *, wq_or_qkv: torch.Tensor, self_attn: bool,
                     kv1=None, kv1_bdiv: int = 1, kv2=None, kv2_bdiv: int = 1, scale2: Optional[torch.Tensor] = None,
                     wo: torch.Tensor, bo: Optional[torch.Tensor], residual: Optional[torch.Tensor],
                     q_ln: Optional[tuple] = None, pair_half: bool = False, phase2_rows: int = 0) -> torch.Tensor:
    """x [B, N, C] bf16 -> out-projected attention output [B, N, C] (+ residual).  ``q_ln`` = (W_q', b_q', eps): ``x`` is the
    block's UN-normalised hidden state and LayerNorm runs inside the Q projection (cross-attention, C = 320 only).
    ``pair_half`` (self-attention with a garment key set only): ``x`` holds the B cond rows of a CFG batch whose uncond rows have
    bit-identical hidden states; the result has 2B rows -- [0, B) the hybrid output, [B, 2B) the plain self-attention output of
    the same rows (the attention launch stores its first phase twice, ``imd_attn_params.out_dup``) -- and ``residual`` has 2B rows, or B (the same block input for both halves, one copy).
    ``phase2_rows`` = R: exactly the rows [0, R) carry the second key set (``scale2`` non-zero): ops.attention may run their two softmaxes as separate
    workgroups (``imd_attn_params.phase2_rows``; bit-identical)."""
    B, N, Cc = x.shape
    D = Cc // heads
    dpk, dpv = ops.attn_padded_dims(D)
    dev, dt = x.device, x.dtype
    q = ops.workspace("attn_q", (B, heads, N, dpk), dt, dev)
    qscale = D ** -0.5 * LOG2E
    x2 = x.view(B * N, Cc)
    if self_attn:
        LP = ops.pad64(N)
        k = ops.k_buffer((B, heads, N, dpk), D, dt, dev, tag="attn_k")
        vt = ops.workspace("attn_vt", (B, heads, dpv, LP), dt, dev)
        dests = dict(C=Cc, H=heads, D=D, dests=[(q, 0, dpk, N, qscale), (k, 0, dpk, N, 1.0), (vt, 1, dpv, LP, 1.0)])
        if q_ln is not None:          # x is the un-normalised block state: norm1 runs inside the projection (row_qkv.hip)
            ops.conv_gemm(x2, q_ln[0], M=B * N, N=3 * Cc, Cin=Cc, Hin=N, Win=1, Hout=N, Wout=1, bias=q_ln[1], ln_eps=q_ln[2], heads=dests)
        else:
            ops.conv_gemm(x2, wq_or_qkv, M=B * N, N=3 * Cc, Cin=Cc, Hin=N, Win=1, Hout=N, Wout=1, heads=dests)
        kv1, kv1_bdiv = (k, vt, N, LP), 1
    elif q_ln is not None:
        ops.conv_gemm(x2, q_ln[0], M=B * N, N=Cc, Cin=Cc, Hin=N, Win=1, Hout=N, Wout=1, bias=q_ln[1], ln_eps=q_ln[2],
                      heads=dict(C=Cc, H=heads, D=D, dests=[(q, 0, dpk, N, qscale)]))
    else:
        ops.conv_gemm(x2, wq_or_qkv, M=B * N, N=Cc, Cin=Cc, Hin=N, Win=1, Hout=N, Wout=1,
                      heads=dict(C=Cc, H=heads, D=D, dests=[(q, 0, dpk, N, qscale)]))
    o = torch.empty(B, N, Cc, dtype=dt, device=dev)
    if ops.ATTN_FP8 and D == 40:            # BASELINE configs[4]: level-0 attention on the MX-FP8 MFMA (opt-in)
        e = ops.FP8_EXPS
        q8 = ops.quantize_fp8_rows(q, e["q"])
        k18, v18 = _fp8_kv(kv1, cache=not self_attn)
        kw = {}
        if kv2 is not None and scale2 is not None:
            k28, v28 = _fp8_kv(kv2, cache=True)
            kw = dict(k2=k28, v2t=v28, L2=kv2[2], L2P=kv2[3], kv2_bdiv=kv2_bdiv, scale2=scale2)
        ops.attention_fp8(q8, k18, v18, o, B=B, H=heads, N=N, L1=kv1[2], L1P=kv1[3], kv1_bdiv=kv1_bdiv, **kw)
        res2 = None if residual is None else residual.view(B * N, Cc)
        return ops.linear(o.view(B * N, Cc), wo, bo, res=res2).view(B, N, Cc)
    kw = {}
    if kv2 is not None and scale2 is not None:
        kw = dict(k2=kv2[0], v2t=kv2[1], L2=kv2[2], L2P=kv2[3], kv2_bdiv=kv2_bdiv, scale2=scale2)
    # every K operand on this path comes from ops.k_buffer (pad column = 1): the d = 40 kernel may stage by LDS-DMA
    if ops.FUSED_OUT_PROJ and ops.attention_proj_supported(heads, N, D):
        # the 64x64-level blocks: to_out[0] + bias + residual ride in the attention launch (ABI v7) -- the hybrid block is TWO launches
        out = torch.empty(B, N, Cc, dtype=dt, device=dev)
        return ops.attention(q, kv1[0], kv1[1], o, B=B, H=heads, N=N, D=D, L1=kv1[2], L1P=kv1[3], kv1_bdiv=kv1_bdiv, k_pad_one=True,
                             proj=(wo, bo, residual, out), **kw)
    if pair_half:
        o = torch.empty(2 * B, N, Cc, dtype=dt, device=dev)
        ops.attention(q, kv1[0], kv1[1], o[:B], B=B, H=heads, N=N, D=D, L1=kv1[2], L1P=kv1[3], kv1_bdiv=kv1_bdiv, k_pad_one=True,
                      out_dup=o[B:], **kw)
        # (the residual may hold B or 2B rows: B = one copy for both halves, added periodically by the projection -- ops.conv_gemm)
        res2 = None if residual is None else residual.reshape(-1, Cc)
        return ops.linear(o.view(2 * B * N, Cc), wo, bo, res=res2).view(2 * B, N, Cc)
    ops.attention(q, kv1[0], kv1[1], o, B=B, H=heads, N=N, D=D, L1=kv1[2], L1P=kv1[3], kv1_bdiv=kv1_bdiv, k_pad_one=True, phase2_rows=phase2_rows if kw else 0, **kw)
    res2 = None if residual is None else residual.view(B * N, Cc)
    return ops.linear(o.view(B * N, Cc), wo, bo, res=res2).view(B, N, Cc)


class _FusedBase:
    fused_residual = True     # the engine hands the block residual in and skips its own add

    @staticmethod
    def _ehs_bdiv(B: int, ehs: torch.Tensor) -> int:
        Be = ehs.shape[0]
        if B % Be:
            raise ValueError(f"encoder_hidden_states batch {Be} does not divide hidden_states batch {B}")
        return B // Be

    @staticmethod
    def _finish(attn, out, residual_given, like, shape4, ln_fused=False):
        # attn.residual_connection / rescale_output_factor are False / 1.0 for SD1.5 (:622-625)
        if getattr(attn, "residual_connection", False) and not residual_given:
            if ln_fused:       # `like` is then the block's UN-normalised state, not this layer's input: the add would be wrong
                raise NotImplementedError("attn.residual_connection together with the engine's fused LayerNorm (imd_layernorm)")
            out = ops.add(out, _as_tokens(like, out.dtype)[0])
        if getattr(attn, "rescale_output_factor", 1.0) != 1.0:
            raise NotImplementedError("rescale_output_factor != 1 is not used by SD1.5")
        out = _restore(out, shape4, like)
        # same dtype as ``hidden_states`` (the processor protocol); a no-op inside the engines
        return out if out.dtype == like.dtype else out.to(like.dtype)


class AttnProcessor2_0(_FusedBase):
    """Plain attention (diffusers' default processor; what a ControlNet / un-patched UNet runs)."""

    # engine-side extension (like fused_residual): for cross-attention on FUSED_LN_CHANNELS channels the engine may hand in the
    # block's un-normalised hidden state plus ``imd_layernorm=(gamma, beta, eps)``; LayerNorm then runs inside the Q projection
    fused_layernorm = True

    def __init__(self, cache_entries: int = 256):
        # ONE instance may serve every attention layer of a model (``set_attn_processor(proc)``, as diffusers
        # does): cached K/V are keyed by the conditioning tensor AND the layer's own projection weights, hence the
        # default of 256 entries.  Processors that embed a private instance per layer pass a handful (cond / uncond
        # alternation) so that a long-running process does not pin stale text K/V of every past call.
        self._text = _TensorCache(capacity=cache_entries)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 imd_residual=None, imd_layernorm=None, **kwargs):
        dt = _compute_dtype(attn, hidden_states, attention_mask)
        x, shape4 = _as_tokens(hidden_states, dt)
        dev = x.device
        wo, bo = _layer_weights(attn, "o", dt, dev), _layer_weights(attn, "bo", dt, dev)
        q_ln = None
        if imd_layernorm is not None:
            g, be, eps = imd_layernorm
            if encoder_hidden_states is not None and x.shape[-1] in FUSED_LN_CHANNELS:
                q_ln = _ln_folded_q(attn, g, be, dt, dev) + (eps,)
            elif encoder_hidden_states is None and _qkv_ln_ok(x):
                q_ln = _ln_folded_q(attn, g, be, dt, dev, "qkv") + (eps,)
            else:
                x = ops.layer_norm(x, g, be, eps)
        if encoder_hidden_states is None:
            out = _fused_attention(x, attn.heads, wq_or_qkv=_layer_weights(attn, "qkv", dt, dev), self_attn=True, wo=wo, bo=bo,
                                   residual=imd_residual, q_ln=q_ln)
        else:
            srcs = (encoder_hidden_states, attn.to_k.weight, attn.to_v.weight)
            kv = self._text.get(srcs, extra=(dt,))
            if kv is None:
                e = encoder_hidden_states.to(device=dev, dtype=dt).contiguous()
                kv = self._text.put(srcs, _project_kv(e, _layer_weights(attn, "kv", dt, dev), attn.heads), extra=(dt,))
            out = _fused_attention(x, attn.heads, wq_or_qkv=_layer_weights(attn, "q", dt, dev), self_attn=False, kv1=kv,
                                   kv1_bdiv=self._ehs_bdiv(x.shape[0], encoder_hidden_states), wo=wo, bo=bo,
                                   residual=imd_residual, q_ln=q_ln)
        return self._finish(attn, out, imd_residual is not None, hidden_states, shape4, ln_fused=imd_layernorm is not None)


class CacheAttnProcessor2_0(AttnProcessor2_0):
    """Garment-UNet processor: remembers its *input* (attention_processor.py:34), then plain attention."""
    fused_layernorm = False      # the cached tensor must be the normalised hidden state the reference caches

    def __init__(self):
        super().__init__()
        self.cache = {}

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 **kwargs):
        self.cache["hidden_states"] = hidden_states
        return super().__call__(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, **kwargs)


class _RefMixin:
    """Garment branch state shared by the three hybrid processors."""

    def _init_ref(self, name, hidden_size, cross_attention_dim, scale):
        self.name = name
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.to_k_ref = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ref = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.scale = scale
        self._wref = _TensorCache()
        self._garment = _TensorCache()
        self._scale2 = _TensorCache()
        self._scale2_plain: Dict = {}

    def _garment_kv(self, ref: torch.Tensor, heads: int, device, dtype):
        """K_ref / V_ref of the garment tokens, projected ONCE per garment (step-invariant)."""
        wsrc = (self.to_k_ref.weight, self.to_v_ref.weight)
        kv = self._garment.get((ref,) + wsrc, extra=(dtype,))
        if kv is None:
            w = self._wref.get(wsrc, extra=(dtype,))
            if w is None:
                w = self._wref.put(wsrc, torch.cat([t.detach().to(device=device, dtype=dtype) for t in wsrc], 0).contiguous(),
                                   extra=(dtype,))
            r = ref.detach().to(device=device, dtype=dtype).contiguous()
            if r.dim() != 3:
                raise ValueError(f"sa_hidden_states[{self.name!r}] must be [Bg, M, C], got {tuple(ref.shape)}")
            kv = self._garment.put((ref,) + wsrc, _project_kv(r, w, heads), extra=(dtype,))
        return kv

    def _garment_bdiv(self, B: int, ref: torch.Tensor) -> int:
        """Batch rows per garment: [1, M, C] is shared by the whole batch; [Bg, M, C] (the reference's own
        ``view(batch_size, ...)`` layout at :602-603, or several garments in one call) serves contiguous groups of
        B / Bg rows.  Anything else would silently pair rows with the wrong garment, so it raises."""
        Bg = ref.shape[0]
        if Bg < 1 or B % Bg:
            raise ValueError(f"sa_hidden_states[{self.name!r}] has batch {Bg}, which does not divide hidden_states batch {B}")
        return B // Bg

    def _branch_weights(self, B: int, mask: Optional[torch.Tensor], device) -> torch.Tensor:
        """[B] fp32 = scale (* sa_batch_mask)."""
        s = float(self.scale)
        if mask is None:
            t = self._scale2_plain.get((B, s, str(device)))
            if t is None:
                t = torch.full((B,), s, dtype=torch.float32, device=device)
                self._scale2_plain = {(B, s, str(device)): t}
            return t
        t = self._scale2.get((mask,), extra=(s,))
        if t is None:
            t = self._scale2.put((mask,), (mask.to(device=device, dtype=torch.float32) * s).contiguous(), extra=(s,))
        return t


class RefSAttnProcessor2_0(nn.Module, _FusedBase, _RefMixin):
    """Hybrid attention: frozen self-attention + trainable garment cross-attention
    (attention_processor.py:513-627)."""
    fused_layernorm = True      # engine-side opt-in: norm1 may run inside the q / k / v projection (see AttnProcessor2_0)
    # engine-side opt-in (round 6): ``imd_pair_half=True`` -- hidden_states holds only the cond half of a CFG batch whose uncond half is
    # bit-identical (first hybrid block), sa_batch_mask / imd_residual keep all 2B rows, the result has 2B rows (_fused_attention)
    fused_pair_half = True

    def __init__(self, name, hidden_size, cross_attention_dim=None, scale=1.0):
        super().__init__()
        self._init_ref(name, hidden_size, cross_attention_dim, scale)

    def _weights(self, attn, dt, dev):
        return _layer_weights(attn, "qkv", dt, dev), _layer_weights(attn, "o", dt, dev), _layer_weights(attn, "bo", dt, dev)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 num_images_per_prompt=1, cond_hidden_states=None, sa_hidden_states=None, sa_batch_mask=None,
                 imd_residual=None, imd_layernorm=None, imd_pair_half=False, **kwargs):
        if encoder_hidden_states is not None:
            raise NotImplementedError(f"{type(self).__name__} is a self-attention (attn1) processor")
        dt = _compute_dtype(attn, hidden_states, attention_mask)
        x, shape4 = _as_tokens(hidden_states, dt)
        wqkv, wo, bo = self._weights(attn, dt, x.device)
        q_ln = None
        if imd_layernorm is not None:
            g, be, eps = imd_layernorm
            if _qkv_ln_ok(x):
                q_ln = _ln_folded_q(attn, g, be, dt, x.device, "qkv") + (eps,)
            else:
                x = ops.layer_norm(x, g, be, eps)
        kv2 = s2 = None
        bdiv2 = 1
        if sa_hidden_states is not None:                                   # :597
            ref = sa_hidden_states[self.name]
            kv2 = self._garment_kv(ref, attn.heads, x.device, x.dtype)
            bdiv2 = self._garment_bdiv(x.shape[0], ref)
            s2 = self._branch_weights(x.shape[0] * (2 if imd_pair_half else 1), sa_batch_mask, x.device)      # (pair_half: the kernel reads the cond rows' entries [0, B))
        if imd_pair_half and (kv2 is None or imd_residual is None or shape4 is not None):
            raise ValueError("imd_pair_half needs sa_hidden_states, the block residual and token-major hidden states (engine-internal)")
        # (sa_pair_layout: the pipeline's mask is "garment on for rows [0, B/2), off for the rest" -- lets the 32x32 / 16x16 / 8x8 levels split the two softmaxes
        #  of the cond rows over workgroups)
        pr = x.shape[0] // 2 if (kwargs.get("sa_pair_layout") and sa_batch_mask is not None and kv2 is not None and not imd_pair_half and x.shape[0] % 2 == 0) else 0
        out = _fused_attention(x, attn.heads, wq_or_qkv=wqkv, self_attn=True, kv2=kv2, kv2_bdiv=bdiv2,
                               scale2=s2, wo=wo, bo=bo, residual=imd_residual, q_ln=q_ln, pair_half=bool(imd_pair_half), phase2_rows=pr)
        return self._finish(attn, out, imd_residual is not None, hidden_states, shape4, ln_fused=imd_layernorm is not None)


class _LoraFold:
    """W_eff = W + lora_scale * up @ down, recomputed only when lora_scale or the LoRA tensors change."""

    def _init_lora(self, hidden_size, kdim, rank, network_alpha, lora_scale):
        self.rank = rank
        self.lora_scale = lora_scale
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = LoRALinearLayer(kdim, hidden_size, rank, network_alpha)
        self.to_v_lora = LoRALinearLayer(kdim, hidden_size, rank, network_alpha)
        self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank, network_alpha)
        self._folded = _TensorCache()

    def _fold(self, attn, device, dtype):
        ls = float(self.lora_scale)
        srcs = tuple(l.weight for lo in (self.to_q_lora, self.to_k_lora, self.to_v_lora, self.to_out_lora)
                     for l in (lo.down, lo.up)) + (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight)
        w = self._folded.get(srcs, extra=(ls, dtype, str(device)))
        if w is None:
            def eff(base, lora):
                base = base.detach().to(device)
                if ls == 0.0:
                    return base.to(dtype).contiguous()
                return (base.float() + ls * lora.delta().to(device)).to(dtype).contiguous()
            wq, wk, wv = eff(attn.to_q.weight, self.to_q_lora), eff(attn.to_k.weight, self.to_k_lora), eff(attn.to_v.weight, self.to_v_lora)
            wo = eff(attn.to_out[0].weight, self.to_out_lora)
            w = self._folded.put(srcs, dict(q=wq, kv=torch.cat([wk, wv], 0).contiguous(),
                                            qkv=(torch.cat([wq, wk, wv], 0).contiguous() if wk.shape[1] == wq.shape[1] else None),
                                            o=wo), extra=(ls, dtype, str(device)))
        return w


class _LoraRefSBase(nn.Module, _FusedBase, _RefMixin, _LoraFold):
    """Hybrid attention with rank-``rank`` LoRA on q/k/v/out.  Deliberately NOT a subclass of
    RefSAttnProcessor2_0: the reference pipelines select processors with isinstance checks
    (IMAGDressing_v1_pipeline.py:342-345, ..._ipa_controlnet.py:379-383) and the classes are siblings."""

    def __init__(self, name, hidden_size, cross_attention_dim=None, scale=1.0, rank=128, network_alpha=None, lora_scale=1.0):
        super().__init__()
        self._init_ref(name, hidden_size, cross_attention_dim, scale)
        self._init_lora(hidden_size, cross_attention_dim or hidden_size, rank, network_alpha, lora_scale)

    def _weights(self, attn, dt, dev):
        w = self._fold(attn, dev, dt)
        return w["qkv"], w["o"], _layer_weights(attn, "bo", dt, dev)

    __call__ = RefSAttnProcessor2_0.__call__
    fused_pair_half = True      # (fused_layernorm stays off: the folded LayerNorm weights are built from attn.to_q/k/v, not from the LoRA-folded ones)


class LoraRefSAttnProcessor2_0(_LoraRefSBase):
    """attention_processor.py:391-511 (used by inference_IMAGdressing_ipa_controlnetpose.py:88-94)."""


class RefLoraSAttnProcessor2_0(_LoraRefSBase):
    """Same arithmetic under the name ``app.py:90`` uses (attention_processor.py:1006-1128).  As in the
    reference, ``pipe.set_scale`` does not match this class (it checks LoraRefSAttnProcessor2_0)."""


class CAttnProcessor2_0(nn.Module, _FusedBase):
    """Text cross-attention (attention_processor.py:202-295); ignores ``sa_hidden_states``."""

    def __init__(self, name, hidden_size, cross_attention_dim=None):
        super().__init__()
        self.name = name
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self._plain = AttnProcessor2_0(cache_entries=4)

    fused_layernorm = True

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 cond_hidden_states=None, sa_hidden_states=None, imd_residual=None, imd_layernorm=None, **kwargs):
        return self._plain(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, imd_residual=imd_residual,
                           imd_layernorm=imd_layernorm)


class _IPBase(nn.Module, _FusedBase, _LoraFold):
    """Text cross-attention + IP-Adapter tokens (+ LoRA): the last ``num_tokens`` rows of
    ``encoder_hidden_states`` are the face tokens (split at attention_processor.py:811-815)."""

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0, scale=1.0,
                 num_tokens=4):
        super().__init__()
        self.num_tokens = num_tokens
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self._init_lora(hidden_size, cross_attention_dim or hidden_size, rank, network_alpha, lora_scale)
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self._kv = _TensorCache()
        self._wip = _TensorCache()
        self._s2: Dict = {}

    def _weights(self, attn, dt, dev):
        w = self._fold(attn, dev, dt)
        return w["q"], w["kv"], w["o"]

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0, temb=None,
                 *args, imd_residual=None, **kwargs):
        if encoder_hidden_states is None:
            raise NotImplementedError(f"{type(self).__name__} is a cross-attention (attn2) processor")
        dt = _compute_dtype(attn, hidden_states, attention_mask)
        x, shape4 = _as_tokens(hidden_states, dt)
        wq, wkv, wo = self._weights(attn, dt, x.device)
        wsrc = (self.to_k_ip.weight, self.to_v_ip.weight)
        ls = float(getattr(self, "lora_scale", 0.0))
        ksrc = (encoder_hidden_states, attn.to_k.weight, attn.to_v.weight) + wsrc
        kvs = self._kv.get(ksrc, extra=(ls, x.dtype))
        if kvs is None:
            wip = self._wip.get(wsrc, extra=(x.dtype,))
            if wip is None:
                wip = self._wip.put(wsrc, torch.cat([t.detach().to(device=x.device, dtype=x.dtype) for t in wsrc], 0).contiguous(),
                                    extra=(x.dtype,))
            e = encoder_hidden_states.to(device=x.device, dtype=x.dtype)
            end = e.shape[1] - self.num_tokens                                       # :811
            kvs = self._kv.put(ksrc,
                               (_project_kv(e[:, :end].contiguous(), wkv, attn.heads),
                                _project_kv(e[:, end:].contiguous(), wip, attn.heads)), extra=(ls, x.dtype))
        B = x.shape[0]
        key = (B, float(self.scale), str(x.device))
        s2 = self._s2.get(key)
        if s2 is None:
            s2 = torch.full((B,), float(self.scale), dtype=torch.float32, device=x.device)
            self._s2 = {key: s2}
        bdiv = self._ehs_bdiv(B, encoder_hidden_states)
        out = _fused_attention(x, attn.heads, wq_or_qkv=wq, self_attn=False, kv1=kvs[0], kv1_bdiv=bdiv, kv2=kvs[1],
                               kv2_bdiv=bdiv, scale2=s2, wo=wo, bo=_layer_weights(attn, "bo", dt, x.device), residual=imd_residual)
        return self._finish(attn, out, imd_residual is not None, hidden_states, shape4)


class LoRAIPAttnProcessor2_0(_IPBase):
    """attention_processor.py:746-871."""


class IPAttnProcessor2_0(_IPBase):
    """IP-Adapter cross-attention without LoRA (attention_processor.py:873-1003).  The reference also
    computes an unused ``attn_map`` every call (:981-982); that dead work is not replicated."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        nn.Module.__init__(self)
        self.num_tokens = num_tokens
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self._kv = _TensorCache()
        self._wip = _TensorCache()
        self._s2 = {}

    def _weights(self, attn, dt, dev):
        return _layer_weights(attn, "q", dt, dev), _layer_weights(attn, "kv", dt, dev), _layer_weights(attn, "o", dt, dev)


# ---- legacy names: defined by the reference but used by none of its entry points ----------------
class BaseSAttnProcessor2_0(nn.Module, _FusedBase):
    """Plain attention with a name (attention_processor.py:298-389)."""

    def __init__(self, name, cross_attention_dim=None):
        super().__init__()
        self.name, self.cross_attention_dim = name, cross_attention_dim
        self._plain = AttnProcessor2_0(cache_entries=4)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 imd_residual=None, **kwargs):
        return self._plain(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, imd_residual=imd_residual)


class SAttnProcessor2_0(nn.Module, _FusedBase):
    """Concat-KV variant (attention_processor.py:103-199): ONE softmax over the keys of ``cat([hidden_states, garment tokens])``
    (:154-159), both through the layer's own to_k / to_v.  Unused by every reference entry point; here it is one phase of the fused
    attention kernel over concatenated K / V^T buffers: the garment half is projected once per garment and cached, the image half is
    projected per call, and the two are laid side by side (copies, no arithmetic) in front of the launch."""

    def __init__(self, name, hidden_size, cross_attention_dim=None):
        super().__init__()
        self.name, self.hidden_size, self.cross_attention_dim = name, hidden_size, cross_attention_dim
        self._plain = AttnProcessor2_0(cache_entries=4)
        self._garment = _TensorCache()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 cond_hidden_states=None, sa_hidden_states=None, imd_residual=None, **kwargs):
        if sa_hidden_states is None or encoder_hidden_states is not None:        # (:152-160: the garment form exists for self-attention only)
            return self._plain(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, imd_residual=imd_residual)
        dt = _compute_dtype(attn, hidden_states, attention_mask)
        x, shape4 = _as_tokens(hidden_states, dt)
        dev, heads = x.device, attn.heads
        B, N, Cc = x.shape
        D = Cc // heads
        ref = sa_hidden_states[self.name]
        if ref.dim() != 3 or B % ref.shape[0]:
            raise ValueError(f"sa_hidden_states[{self.name!r}] must be [Bg, M, C] with Bg dividing the batch {B}, got {tuple(ref.shape)}")
        wkv = _layer_weights(attn, "kv", dt, dev)
        srcs = (ref, attn.to_k.weight, attn.to_v.weight)
        kvr = self._garment.get(srcs, extra=(dt,))
        if kvr is None:
            kvr = self._garment.put(srcs, _project_kv(ref.detach().to(device=dev, dtype=dt).contiguous(), wkv, heads), extra=(dt,))
        kx, vtx, _, _ = _project_kv(x, wkv, heads)
        M = kvr[2]
        L, LP = N + M, ops.pad64(N + M)
        dpk, dpv = ops.attn_padded_dims(D)
        k = ops.k_buffer((B, heads, L, dpk), D, dt, dev)
        vt = torch.zeros(B, heads, dpv, LP, dtype=dt, device=dev)
        rep = B // ref.shape[0]
        k[:, :, :N] = kx
        k[:, :, N:] = kvr[0].repeat_interleave(rep, 0) if rep > 1 else kvr[0]
        vt[..., :N] = vtx[..., :N]
        vt[..., N:L] = (kvr[1].repeat_interleave(rep, 0) if rep > 1 else kvr[1])[..., :M]
        out = _fused_attention(x, heads, wq_or_qkv=_layer_weights(attn, "q", dt, dev), self_attn=False, kv1=(k, vt, L, LP), kv1_bdiv=1,
                               wo=_layer_weights(attn, "o", dt, dev), bo=_layer_weights(attn, "bo", dt, dev), residual=imd_residual)
        return self._finish(attn, out, imd_residual is not None, hidden_states, shape4)


class RefCAttnProcessor2_0(nn.Module, _FusedBase, _RefMixin):
    """Cross-attention + garment-token variant (attention_processor.py:630-743); unused by the reference's entry points.  Text (or,
    without ``encoder_hidden_states``, self) attention plus a second softmax over the garment tokens through to_k_ref / to_v_ref,
    added with ``self.scale`` -- the hybrid kernel's two phases with the text K / V as the first key set."""

    def __init__(self, name, hidden_size, cross_attention_dim=None, scale=1.0):
        super().__init__()
        self._init_ref(name, hidden_size, None, scale)          # to_k_ref / to_v_ref are hidden_size x hidden_size here (:644-645)
        self.cross_attention_dim = cross_attention_dim
        self._plain = AttnProcessor2_0(cache_entries=4)
        self._text = _TensorCache()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 cond_hidden_states=None, sa_hidden_states=None, sa_batch_mask=None, imd_residual=None, **kwargs):
        if sa_hidden_states is None:                                               # :706 not taken
            return self._plain(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, imd_residual=imd_residual)
        dt = _compute_dtype(attn, hidden_states, attention_mask)
        x, shape4 = _as_tokens(hidden_states, dt)
        dev, heads = x.device, attn.heads
        ref = sa_hidden_states[self.name]
        kv2 = self._garment_kv(ref, heads, dev, dt)
        bdiv2 = self._garment_bdiv(x.shape[0], ref)
        s2 = self._branch_weights(x.shape[0], sa_batch_mask, dev)
        wo, bo = _layer_weights(attn, "o", dt, dev), _layer_weights(attn, "bo", dt, dev)
        if encoder_hidden_states is None:                                          # :681-682
            out = _fused_attention(x, heads, wq_or_qkv=_layer_weights(attn, "qkv", dt, dev), self_attn=True, kv2=kv2, kv2_bdiv=bdiv2,
                                   scale2=s2, wo=wo, bo=bo, residual=imd_residual)
        else:
            srcs = (encoder_hidden_states, attn.to_k.weight, attn.to_v.weight)
            kv = self._text.get(srcs, extra=(dt,))
            if kv is None:
                e = encoder_hidden_states.to(device=dev, dtype=dt).contiguous()
                kv = self._text.put(srcs, _project_kv(e, _layer_weights(attn, "kv", dt, dev), heads), extra=(dt,))
            out = _fused_attention(x, heads, wq_or_qkv=_layer_weights(attn, "q", dt, dev), self_attn=False, kv1=kv,
                                   kv1_bdiv=self._ehs_bdiv(x.shape[0], encoder_hidden_states), kv2=kv2, kv2_bdiv=bdiv2, scale2=s2,
                                   wo=wo, bo=bo, residual=imd_residual)
        return self._finish(attn, out, imd_residual is not None, hidden_states, shape4)
