"""fp32 CPU restatement of ``/root/reference/adapter/resampler.py`` (TEST INFRASTRUCTURE).

Functional form over a state-dict with the reference's key layout
(``latents``, ``proj_in.*``, ``layers.{i}.0.{norm1,norm2,to_q,to_kv,to_out}.*``,
``layers.{i}.1.{0,1,3}.*``, ``proj_out.*``, ``norm_out.*``).  Pinned against the reference
module by ``tests/golden/resampler.pt``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def perceiver_attention(sd, p, x, latents, heads):
    """``PerceiverAttention.forward`` (resampler.py:49-78)."""
    x = _ln(x, sd, p + ".norm1")                                # :57
    latents = _ln(latents, sd, p + ".norm2")                    # :58
    b, l, _ = latents.shape
    q = latents @ sd[p + ".to_q.weight"].t()                    # :62
    kv_in = torch.cat((x, latents), dim=-2)                     # :63
    kv = kv_in @ sd[p + ".to_kv.weight"].t()
    k, v = kv.chunk(2, dim=-1)                                  # :64
    inner = q.shape[-1]
    dh = inner // heads

    def split(t):                                               # reshape_tensor :23-31
        return t.view(t.shape[0], t.shape[1], heads, dh).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    s = 1.0 / math.sqrt(math.sqrt(dh))                          # :70 (d^-1/4 on q and on k)
    w = (q * s) @ (k * s).transpose(-2, -1)                     # :71
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)          # :73 fp32 softmax
    o = w @ v                                                   # :74
    o = o.permute(0, 2, 1, 3).reshape(b, l, -1)                 # :76
    return o @ sd[p + ".to_out.weight"].t()                     # :78


def feed_forward(sd, p, x):
    """``FeedForward`` (resampler.py:13-20): LN -> Linear -> GELU(erf) -> Linear, no biases."""
    h = _ln(x, sd, p + ".0")
    h = F.gelu(h @ sd[p + ".1.weight"].t())
    return h @ sd[p + ".3.weight"].t()


def _layers(sd, prefix, x, latents, heads):
    depth = 0
    while f"{prefix}layers.{depth}.0.to_q.weight" in sd:
        depth += 1
    for i in range(depth):
        latents = perceiver_attention(sd, f"{prefix}layers.{i}.0", x, latents, heads) + latents
        latents = feed_forward(sd, f"{prefix}layers.{i}.1", latents) + latents
    return latents


def resampler_forward(sd, x, heads):
    """``Resampler.forward`` (resampler.py:216-236) with apply_pos_emb=False and
    num_latents_mean_pooled=0 (the only configuration the entry scripts build,
    inference_IMAGdressing.py:55-64)."""
    latents = sd["latents"].repeat(x.size(0), 1, 1)             # :222
    x = x @ sd["proj_in.weight"].t() + sd["proj_in.bias"]       # :224
    latents = _layers(sd, "", x, latents, heads)                # :231-233
    latents = latents @ sd["proj_out.weight"].t() + sd["proj_out.bias"]
    return _ln(latents, sd, "norm_out")                         # :235-236


def proj_plus_forward(sd, id_embeds, clip_embeds, heads=12, num_tokens=4, shortcut=False, scale=1.0):
    """``ProjPlusModel.forward`` (resampler.py:274-281) + ``FacePerceiverResampler.forward``
    (:158-167)."""
    h = id_embeds @ sd["proj.0.weight"].t() + sd["proj.0.bias"]
    h = F.gelu(h)
    h = h @ sd["proj.2.weight"].t() + sd["proj.2.bias"]
    cad = sd["norm.weight"].shape[0]
    x = h.reshape(-1, num_tokens, cad)                          # :276
    x = _ln(x, sd, "norm")                                      # :277
    pr = "perceiver_resampler."
    c = clip_embeds @ sd[pr + "proj_in.weight"].t() + sd[pr + "proj_in.bias"]
    lat = _layers(sd, pr, c, x, heads)
    lat = lat @ sd[pr + "proj_out.weight"].t() + sd[pr + "proj_out.bias"]
    out = _ln(lat, sd, pr + "norm_out")
    if shortcut:                                                # :279-280
        out = x + scale * out
    return out
