"""fp32 CPU restatement of the reference attention processors (TEST INFRASTRUCTURE).

Follows ``/root/reference/adapter/attention_processor.py``; each function cites the
lines it restates.  The math is written as explicit softmax(QK^T/sqrt(d))V so that it
does not depend on ``F.scaled_dot_product_attention`` dispatch.  Pinned against the
reference source by ``tests/golden/processors.pt`` (``oracle/make_golden.py``).
"""
from __future__ import annotations

import contextlib
import math
from typing import Optional

import torch
import torch.nn.functional as F


def _heads(x: torch.Tensor, heads: int) -> torch.Tensor:
    # [B, L, C] -> [B, heads, L, d]   (attention_processor.py:582-585)
    b, l, c = x.shape
    return x.view(b, l, heads, c // heads).transpose(1, 2)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v on [B, L, C] tensors (attention_processor.py:589-594)."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    d = qh.shape[-1]
    w = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    o = w @ vh
    b, h, l, _ = o.shape
    return o.transpose(1, 2).reshape(b, l, h * d)


def sdpa_fused(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """The same product through ``F.scaled_dot_product_attention`` on [B, heads, L, d] views with no mask, dropout 0 -- the call
    the reference makes (attention_processor.py:589-591 and :607-609).  On the CPU this dispatches to torch's flash kernel, which
    never materialises the [B, heads, L, L] scores: it is what the reference's CPU path COSTS.  Used only where the oracle is
    timed as the CPU baseline (``reference_sdpa_dispatch``); the oracle proper keeps the explicit form above."""
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    o = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=None, dropout_p=0.0, is_causal=False)
    b, h, l, d = o.shape
    return o.transpose(1, 2).reshape(b, l, h * d)


@contextlib.contextmanager
def reference_sdpa_dispatch():
    """Inside this context every processor of this module computes its attention products with ``sdpa_fused`` -- the timing form.
    bench.py's ``cpu_baseline`` legs run under it so that the reported CPU figure is the reference's own CPU path
    (SDPA, attention_processor.py:589,607) and not the slower explicit-softmax restatement."""
    global sdpa
    explicit = sdpa
    sdpa = sdpa_fused
    try:
        yield
    finally:
        sdpa = explicit


def lora_delta(x, down: Optional[torch.Tensor], up: Optional[torch.Tensor], network_alpha=None):
    """diffusers-0.24 ``LoRALinearLayer.forward``: up(down(x)) [* alpha/rank]."""
    if down is None:
        return 0.0
    y = (x @ down.t()) @ up.t()
    if network_alpha is not None:
        y = y * (network_alpha / down.shape[0])
    return y


def hybrid_self_attention(
    x, wq, wk, wv, wo, bo, heads,
    ref=None, wk_ref=None, wv_ref=None, scale=1.0,
    lora=None, lora_scale=0.0,
):
    """``RefSAttnProcessor2_0.__call__`` (attention_processor.py:531-627) and, with
    ``lora`` given, ``LoraRefSAttnProcessor2_0.__call__`` (:416-511).

    x [B,N,C]; ref [1,M,C] garment hidden states (``sa_hidden_states[name]``, :598) or
    None for the uncond pass (:597 not taken).  ``lora`` is a dict with keys q,k,v,out ->
    (down, up) (:453, :461-462, :500).
    """
    lq = lk = lv = lo = (None, None)
    if lora is not None:
        lq, lk, lv, lo = lora["q"], lora["k"], lora["v"], lora["out"]
    q = x @ wq.t() + lora_scale * lora_delta(x, *lq)          # :568 / :453
    k = x @ wk.t() + lora_scale * lora_delta(x, *lk)          # :576 / :461
    v = x @ wv.t() + lora_scale * lora_delta(x, *lv)          # :577 / :462
    h = sdpa(q, k, v, heads)                                   # :589-594
    if ref is not None:                                        # :597
        rk = ref @ wk_ref.t()                                  # :600
        rv = ref @ wv_ref.t()                                  # :601
        # the reference views rk/rv with x's batch size (:602-603), only defined for B==1;
        # batched generation is defined as B independent B=1 runs -> broadcast over batch.
        rk = rk.expand(x.shape[0], -1, -1)
        rv = rv.expand(x.shape[0], -1, -1)
        h = h + sdpa(q, rk, rv, heads) * scale                 # :607-612 (separate softmax)
    return h @ wo.t() + bo + lora_scale * lora_delta(h, *lo)   # :615 / :500


def text_cross_attention(x, ehs, wq, wk, wv, wo, bo, heads):
    """``CAttnProcessor2_0.__call__`` (attention_processor.py:219-295)."""
    q = x @ wq.t()
    k = ehs @ wk.t()
    v = ehs @ wv.t()
    return sdpa(q, k, v, heads) @ wo.t() + bo


def ip_cross_attention(
    x, ehs, wq, wk, wv, wo, bo, heads, wk_ip, wv_ip, scale=1.0, num_tokens=4,
    lora=None, lora_scale=0.0,
):
    """``LoRAIPAttnProcessor2_0.__call__`` (attention_processor.py:781-871) and, without
    lora, ``IPAttnProcessor2_0.__call__`` (:903-1003, minus the unused attn_map :981-982).

    ehs [B, T+num_tokens, 768]: split at :811-815 into text and IP tokens.
    """
    lq = lk = lv = lo = (None, None)
    if lora is not None:
        lq, lk, lv, lo = lora["q"], lora["k"], lora["v"], lora["out"]
    end = ehs.shape[1] - num_tokens
    text, ip = ehs[:, :end], ehs[:, end:]
    q = x @ wq.t() + lora_scale * lora_delta(x, *lq)           # :807
    k = text @ wk.t() + lora_scale * lora_delta(text, *lk)     # :820
    v = text @ wv.t() + lora_scale * lora_delta(text, *lv)     # :821
    h = sdpa(q, k, v, heads)                                    # :833-838
    hi = sdpa(q, ip @ wk_ip.t(), ip @ wv_ip.t(), heads)         # :841-854
    h = h + scale * hi                                          # :856
    return h @ wo.t() + bo + lora_scale * lora_delta(h, *lo)    # :859


def concat_self_attention(x, wq, wk, wv, wo, bo, heads, ref=None):
    """``SAttnProcessor2_0.__call__`` (attention_processor.py:119-199): ONE softmax over the keys of
    ``cat([hidden_states, sa_hidden_states[name]], dim=1)`` (:154-159), K / V of both through the layer's own to_k / to_v
    (:164-165).  ref [1, M, C] is broadcast over the batch (batched generation = independent B = 1 runs)."""
    ehs = x if ref is None else torch.cat([x, ref.expand(x.shape[0], -1, -1)], dim=1)
    return sdpa(x @ wq.t(), ehs @ wk.t(), ehs @ wv.t(), heads) @ wo.t() + bo


def cross_plus_garment_attention(x, ehs, wq, wk, wv, wo, bo, heads, ref=None, wk_ref=None, wv_ref=None, scale=1.0):
    """``RefCAttnProcessor2_0.__call__`` (attention_processor.py:649-743): text cross-attention (:681-704; self-attention when
    ``encoder_hidden_states`` is None, :681-682) plus, when ``sa_hidden_states`` is given (:706), a second softmax over the garment
    tokens through to_k_ref / to_v_ref added with ``self.scale`` (:709-722)."""
    q = x @ wq.t()
    e = x if ehs is None else ehs
    h = sdpa(q, e @ wk.t(), e @ wv.t(), heads)
    if ref is not None:
        r = ref.expand(x.shape[0], -1, -1)
        h = h + sdpa(q, r @ wk_ref.t(), r @ wv_ref.t(), heads) * scale
    return h @ wo.t() + bo


# --------------------------------------------------------------------------------------
# AttnProcessor-protocol wrappers over the functions above, so the oracle UNet
# (oracle/sd15.py) can run the reference's loop semantics on hosts where /root/reference is
# absent (the GPU box).  State-dict keys equal the reference classes' keys.
# --------------------------------------------------------------------------------------
import torch.nn as nn


class _LoRA(nn.Module):
    """diffusers-0.24 ``LoRALinearLayer`` parameter layout (down/up, no bias)."""

    def __init__(self, cin, cout, rank, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(cin, rank, bias=False)
        self.up = nn.Linear(rank, cout, bias=False)
        self.network_alpha = network_alpha
        nn.init.normal_(self.down.weight, std=1.0 / rank)
        nn.init.zeros_(self.up.weight)

    def pair(self):
        up = self.up.weight
        if self.network_alpha is not None:
            up = up * (self.network_alpha / self.down.weight.shape[0])
        return self.down.weight, up


def _attn_w(attn):
    return (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias)


class CacheAttn:
    """``CacheAttnProcessor2_0`` (attention_processor.py:13-100): stores its *input* (:34)."""

    def __init__(self):
        self.cache = {}

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        self.cache["hidden_states"] = hidden_states
        wq, wk, wv, wo, bo = _attn_w(attn)
        ehs = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        return sdpa(hidden_states @ wq.t(), ehs @ wk.t(), ehs @ wv.t(), attn.heads) @ wo.t() + bo


class RefSAttn(nn.Module):
    """``RefSAttnProcessor2_0`` (:513-627)."""

    def __init__(self, name, hidden_size, cross_attention_dim=None, scale=1.0):
        super().__init__()
        self.name, self.scale = name, scale
        self.to_k_ref = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ref = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None,
                 sa_hidden_states=None, **kw):
        ref = None if sa_hidden_states is None else sa_hidden_states[self.name]
        return hybrid_self_attention(hidden_states, *_attn_w(attn), attn.heads, ref=ref,
                                     wk_ref=self.to_k_ref.weight, wv_ref=self.to_v_ref.weight, scale=self.scale)


class LoraRefSAttn(RefSAttn):
    """``LoraRefSAttnProcessor2_0`` (:391-511) / ``RefLoraSAttnProcessor2_0`` (:1006-1128)."""

    def __init__(self, name, hidden_size, cross_attention_dim=None, scale=1.0, rank=128, network_alpha=None,
                 lora_scale=1.0):
        super().__init__(name, hidden_size, cross_attention_dim, scale)
        self.lora_scale = lora_scale
        kd = cross_attention_dim or hidden_size
        self.to_q_lora = _LoRA(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = _LoRA(kd, hidden_size, rank, network_alpha)
        self.to_v_lora = _LoRA(kd, hidden_size, rank, network_alpha)
        self.to_out_lora = _LoRA(hidden_size, hidden_size, rank, network_alpha)

    def _lora(self):
        return dict(q=self.to_q_lora.pair(), k=self.to_k_lora.pair(), v=self.to_v_lora.pair(),
                    out=self.to_out_lora.pair())

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None,
                 sa_hidden_states=None, **kw):
        ref = None if sa_hidden_states is None else sa_hidden_states[self.name]
        return hybrid_self_attention(hidden_states, *_attn_w(attn), attn.heads, ref=ref,
                                     wk_ref=self.to_k_ref.weight, wv_ref=self.to_v_ref.weight, scale=self.scale,
                                     lora=self._lora(), lora_scale=self.lora_scale)


class CAttn(nn.Module):
    """``CAttnProcessor2_0`` (:202-295)."""

    def __init__(self, name, hidden_size, cross_attention_dim=None):
        super().__init__()
        self.name = name

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return text_cross_attention(hidden_states, encoder_hidden_states, *_attn_w(attn), attn.heads)


class LoRAIPAttn(nn.Module):
    """``LoRAIPAttnProcessor2_0`` (:746-871)."""

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, network_alpha=None, lora_scale=1.0,
                 scale=1.0, num_tokens=4):
        super().__init__()
        self.lora_scale, self.scale, self.num_tokens = lora_scale, scale, num_tokens
        kd = cross_attention_dim or hidden_size
        self.to_q_lora = _LoRA(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_lora = _LoRA(kd, hidden_size, rank, network_alpha)
        self.to_v_lora = _LoRA(kd, hidden_size, rank, network_alpha)
        self.to_out_lora = _LoRA(hidden_size, hidden_size, rank, network_alpha)
        self.to_k_ip = nn.Linear(kd, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(kd, hidden_size, bias=False)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        lora = dict(q=self.to_q_lora.pair(), k=self.to_k_lora.pair(), v=self.to_v_lora.pair(),
                    out=self.to_out_lora.pair())
        return ip_cross_attention(hidden_states, encoder_hidden_states, *_attn_w(attn), attn.heads,
                                  self.to_k_ip.weight, self.to_v_ip.weight, scale=self.scale,
                                  num_tokens=self.num_tokens, lora=lora, lora_scale=self.lora_scale)
