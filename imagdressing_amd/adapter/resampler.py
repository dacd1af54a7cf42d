"""Perceiver resamplers / projection models of IMAGDressing-v1 on the HIP kernels.

Same classes, constructor signatures, ``forward`` signatures and ``state_dict`` keys as
``/root/reference/adapter/resampler.py`` (``latents``, ``proj_in.*``,
``layers.{i}.0.{norm1,norm2,to_q,to_kv,to_out}.*``, ``layers.{i}.1.{0,1,3}.*``, ``proj_out.*``,
``norm_out.*``), so ``image_proj.load_state_dict(image_proj_dict)`` (inference_IMAGdressing.py:116)
works unchanged.  ``forward`` runs LayerNorm, the projections (GEMM + fused GELU / residual
epilogues) and the latent attention (fused attention kernel, fp32 softmax like resampler.py:73) in
HIP; the module tree only holds parameters.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from .attention_processor import LOG2E, _TensorCache, _project_kv

bf16 = torch.bfloat16


def _packed(cache: _TensorCache, params, dtype, device):
    """bf16 (weights) / fp32 (norm + bias) device copies of parameters, rebuilt when they change."""
    v = cache.get(params, extra=(str(dtype), str(device)))
    if v is None:
        v = cache.put(params, [p.detach().to(device=device, dtype=dtype).contiguous() for p in params],
                      extra=(str(dtype), str(device)))
    return v


class _HipModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._wc = _TensorCache(2)
        self._fc = _TensorCache(2)

    def w(self, *params, device, dtype=bf16):
        return _packed(self._wc, params, dtype, device)

    def f(self, *params, device):
        return _packed(self._fc, params, torch.float32, device)


class _FeedForward(nn.Sequential):
    """LN -> Linear(no bias) -> GELU -> Linear(no bias); indices 0,1,(2),3 as in resampler.py:13-20."""

    def __init__(self, dim, mult=4):
        inner = int(dim * mult)
        super().__init__(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))
        self._wc, self._fc = _TensorCache(2), _TensorCache(2)

    def forward(self, x):          # returns ff(x) + x  (the residual is fused into the 2nd GEMM)
        dev = x.device
        g, b = _packed(self._fc, (self[0].weight, self[0].bias), torch.float32, dev)
        w1, w2 = _packed(self._wc, (self[1].weight, self[3].weight), x.dtype, dev)
        B, L, Cc = x.shape
        n = ops.layer_norm(x, g, b, self[0].eps)
        h = ops.linear(n.view(B * L, Cc), w1, None, act=ops.ACT_GELU)
        return ops.linear(h, w2, None, res=x.view(B * L, Cc)).view(B, L, Cc)


def FeedForward(dim, mult=4):
    return _FeedForward(dim, mult)


def reshape_tensor(x, heads):
    bs, length, width = x.shape
    return x.view(bs, length, heads, -1).transpose(1, 2).reshape(bs, heads, length, -1)


class PerceiverAttention(_HipModule):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.dim_head = dim_head
        self.heads = heads
        inner_dim = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, x, latents):
        """x [B, n1, D] image features, latents [B, n2, D] (bf16, cuda) -> attn(x, latents) + latents."""
        dev = x.device
        g1, b1, g2, b2 = self.f(self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, device=dev)
        wq, wkv, wo = self.w(self.to_q.weight, self.to_kv.weight, self.to_out.weight, device=dev, dtype=x.dtype)
        B, Lq, Dm = latents.shape
        inner = self.heads * self.dim_head
        xn = ops.layer_norm(x, g1, b1, self.norm1.eps)                       # resampler.py:57
        ln = ops.layer_norm(latents, g2, b2, self.norm2.eps)                 # :58
        kv_in = ops.concat_tokens(xn, ln)                                    # :63
        k, vt, Lk, LP = _project_kv(kv_in, wkv, self.heads)                  # :64
        dpk, _ = ops.attn_padded_dims(self.dim_head)
        q = torch.zeros(B, self.heads, Lq, dpk, dtype=x.dtype, device=dev)
        # q*d^-1/4 . k*d^-1/4 (:70-71) == qk * d^-1/2, folded into Q together with log2(e)
        ops.conv_gemm(ln.view(B * Lq, Dm), wq, M=B * Lq, N=inner, Cin=Dm, Hin=Lq, Win=1, Hout=Lq, Wout=1,
                      heads=dict(C=inner, H=self.heads, D=self.dim_head,
                                 dests=[(q, 0, dpk, Lq, self.dim_head ** -0.5 * LOG2E)]))
        o = torch.empty(B, Lq, inner, dtype=x.dtype, device=dev)
        ops.attention(q, k, vt, o, B=B, H=self.heads, N=Lq, D=self.dim_head, L1=Lk, L1P=LP)   # :73-74 (fp32 softmax)
        return ops.linear(o.view(B * Lq, inner), wo, None, res=latents.view(B * Lq, Dm)).view(B, Lq, Dm)


def _run_layers(layers, x, latents):
    for attn, ff in layers:
        latents = attn(x, latents)        # includes "+ latents"
        latents = ff(latents)             # includes "+ latents"
    return latents


def _make_layers(dim, depth, dim_head, heads, ff_mult):
    return nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                         FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])


def _to_dev16(x: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    """16-bit compute copy: fp16 / bf16 inputs keep their type, anything else computes in bf16."""
    if not x.is_cuda:
        from .._lib import ImdError
        raise ImdError(f"{name}: tensor is on {x.device}; imagdressing_amd runs on MI355X only (no CPU path)")
    if dtype is None:
        dtype = x.dtype if x.dtype in ops.DTYPE_CODE else bf16
    return x.to(dtype).contiguous()


class _ProjInOut(_HipModule):
    def _proj(self, x, lin):
        w, = self.w(lin.weight, device=x.device, dtype=x.dtype)
        b, = self.f(lin.bias, device=x.device)
        return ops.linear(x.reshape(-1, x.shape[-1]), w, b).view(*x.shape[:-1], lin.out_features)

    def _out(self, latents):
        y = self._proj(latents, self.proj_out)
        g, b = self.f(self.norm_out.weight, self.norm_out.bias, device=latents.device)
        return ops.layer_norm(y, g, b, self.norm_out.eps)


class PerceiverResampler(_ProjInOut):
    def __init__(self, *, dim=1024, depth=8, dim_head=64, heads=16, num_latents=8, embedding_dim=768, output_dim=1024, ff_mult=4):
        super().__init__()
        self.latents = nn.Parameter(torch.randn(1, num_latents, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = _make_layers(dim, depth, dim_head, heads, ff_mult)

    def forward(self, x):
        dt = x.dtype
        x = _to_dev16(x, "x")
        latents = self.latents.detach().to(device=x.device, dtype=x.dtype).repeat(x.size(0), 1, 1).contiguous()
        x = self._proj(x, self.proj_in)
        return self._out(_run_layers(self.layers, x, latents)).to(dt)


class FacePerceiverResampler(_ProjInOut):
    def __init__(self, *, dim=768, depth=4, dim_head=64, heads=16, embedding_dim=1280, output_dim=768, ff_mult=4):
        super().__init__()
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = _make_layers(dim, depth, dim_head, heads, ff_mult)

    def forward(self, latents, x):
        latents = _to_dev16(latents, "latents")
        x = self._proj(_to_dev16(x, "x", latents.dtype), self.proj_in)
        return self._out(_run_layers(self.layers, x, latents))


class Resampler(_ProjInOut):
    """resampler.py:170-236.  ``apply_pos_emb`` / ``num_latents_mean_pooled`` are accepted for signature
    compatibility; no reference entry point enables them (inference_IMAGdressing.py:55-64) and the HIP
    path refuses them instead of silently diverging."""

    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4, max_seq_len: int = 257, apply_pos_emb: bool = False, num_latents_mean_pooled: int = 0):
        super().__init__()
        if apply_pos_emb or num_latents_mean_pooled:
            raise NotImplementedError("Resampler: apply_pos_emb / num_latents_mean_pooled are unused by IMAGDressing")
        self.pos_emb = None
        self.to_latents_from_mean_pooled_seq = None
        self.latents = nn.Parameter(torch.randn(1, num_queries, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = _make_layers(dim, depth, dim_head, heads, ff_mult)

    def forward(self, x):
        dt = x.dtype
        x = _to_dev16(x, "x")
        latents = self.latents.detach().to(device=x.device, dtype=x.dtype).repeat(x.size(0), 1, 1).contiguous()   # :222
        x = self._proj(x, self.proj_in)                                                                         # :224
        return self._out(_run_layers(self.layers, x, latents)).to(dt)                                           # :231-236


def masked_mean(t, *, dim, mask=None):
    if mask is None:
        return t.mean(dim=dim)
    denom = mask.sum(dim=dim, keepdim=True)
    masked_t = t.masked_fill(~mask.unsqueeze(-1), 0.0)
    return masked_t.sum(dim=dim) / denom.clamp(min=1e-5)


class ProjPlusModel(_HipModule):
    """IP-Adapter-FaceID-Plus projection (resampler.py:250-281): id MLP -> 4 tokens -> LN -> face Perceiver."""

    def __init__(self, cross_attention_dim=768, id_embeddings_dim=512, clip_embeddings_dim=1280, num_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.num_tokens = num_tokens
        self.proj = nn.Sequential(nn.Linear(id_embeddings_dim, id_embeddings_dim * 2), nn.GELU(),
                                  nn.Linear(id_embeddings_dim * 2, cross_attention_dim * num_tokens))
        self.norm = nn.LayerNorm(cross_attention_dim)
        self.perceiver_resampler = FacePerceiverResampler(
            dim=cross_attention_dim, depth=4, dim_head=64, heads=cross_attention_dim // 64,
            embedding_dim=clip_embeddings_dim, output_dim=cross_attention_dim, ff_mult=4)

    def forward(self, id_embeds, clip_embeds, shortcut=False, scale=1.0):
        dt = id_embeds.dtype
        idv = _to_dev16(id_embeds, "id_embeds")
        dev = idv.device
        w0, w2 = self.w(self.proj[0].weight, self.proj[2].weight, device=dev, dtype=idv.dtype)
        b0, b2, g, b = self.f(self.proj[0].bias, self.proj[2].bias, self.norm.weight, self.norm.bias, device=dev)
        h = ops.linear(idv.view(-1, idv.shape[-1]), w0, b0, act=ops.ACT_GELU)
        h = ops.linear(h, w2, b2)
        x = h.view(-1, self.num_tokens, self.cross_attention_dim)                 # :276
        x = ops.layer_norm(x, g, b, self.norm.eps)                                # :277
        out = self.perceiver_resampler(x, clip_embeds)                            # :278
        if shortcut:                                                              # :279-280
            out = ops.add(x, out, float(scale))
        return out.to(dt)
