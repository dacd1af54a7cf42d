"""Generate ``tests/golden/*.pt`` by executing the REFERENCE's own source
(``/root/reference/adapter/attention_processor.py`` and ``adapter/resampler.py``, imported
verbatim through ``oracle/ref_loader.py``) on seeded fp32 inputs (TEST INFRASTRUCTURE).

Run in the build container only:   python -m oracle.make_golden [base|full|geometry|legacy|unet|timesteps|all|trajectory [case ...]]
The fixtures pin ``oracle/processors.py`` and ``oracle/resampler.py`` (tests/test_oracle_golden.py)
and are what the ``-m gpu`` parity tests compare the HIP path with when /root/reference is absent.

Two kinds of cases:
  * "full":   small dims; inputs, weights and reference outputs are all stored.
  * "seeded": real SD1.5 dims (d = 40/80/160, T = 77+4, Resampler 257x1280 -> 16x768); only seeds,
              input digests and reference outputs are stored; tests regenerate inputs with
              ``oracle.seeds.seeded``.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .ref_loader import load_reference_adapter
from .seeds import digest, seeded

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class FakeAttention(nn.Module):
    """The attribute surface the reference processors read (attention_processor.py:545-625)."""

    def __init__(self, c, kdim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(c, c, bias=False)
        self.to_k = nn.Linear(kdim, c, bias=False)
        self.to_v = nn.Linear(kdim, c, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0

    def prepare_attention_mask(self, m, *a, **k):
        return m


def attn_weights(seed, c, kdim):
    s = c ** -0.5
    return dict(
        wq=seeded(seed + 1, c, c, scale=s), wk=seeded(seed + 2, c, kdim, scale=kdim ** -0.5),
        wv=seeded(seed + 3, c, kdim, scale=kdim ** -0.5), wo=seeded(seed + 4, c, c, scale=s),
        bo=seeded(seed + 5, c, scale=0.1),
    )


def make_attn(w, c, kdim, heads):
    a = FakeAttention(c, kdim, heads)
    with torch.no_grad():
        a.to_q.weight.copy_(w["wq"]); a.to_k.weight.copy_(w["wk"]); a.to_v.weight.copy_(w["wv"])
        a.to_out[0].weight.copy_(w["wo"]); a.to_out[0].bias.copy_(w["bo"])
    return a


def lora_weights(seed, c, kdim, rank):
    # up is zero-initialised in diffusers; use non-zero values so the LoRA path is exercised
    out = {}
    for i, (nm, cin) in enumerate((("q", c), ("k", kdim), ("v", kdim), ("out", c))):
        out[nm] = (seeded(seed + 10 + 2 * i, rank, cin, scale=cin ** -0.5),
                   seeded(seed + 11 + 2 * i, c, rank, scale=rank ** -0.5))
    return out


def set_lora(proc, lw):
    with torch.no_grad():
        for nm in ("q", "k", "v", "out"):
            layer = getattr(proc, f"to_{nm}_lora")
            layer.down.weight.copy_(lw[nm][0]); layer.up.weight.copy_(lw[nm][1])


def spike_tokens(x, ref, spike):
    """``spike`` = (image token, garment token, factor): scale those rows of x / ref so that, late in both key sequences, the
    scores of MANY query rows jump far above everything seen before (logits of that key have std ``factor`` instead of 1:
    the online-softmax rescale / redo paths of the kernel).  The factor stays moderate on purpose: a 16-bit Q / K carries
    a logit of magnitude s only to about s * 2^-11 (fp16), so wilder spikes would measure the element type, not the kernel."""
    if spike:
        kt, gt, f = spike
        x[:, kt] *= f
        ref[:, gt] *= f
    return x, ref


@torch.no_grad()
def hybrid_case(ap, seed, B, N, M, C, heads, scale, rank=0, lora_scale=0.0, store_full=True, spike=None, keep_rows=0):
    w = attn_weights(seed, C, C)
    attn = make_attn(w, C, C, heads)
    x = seeded(seed + 20, B, N, C)
    ref = seeded(seed + 21, 1, M, C)
    x, ref = spike_tokens(x, ref, spike)
    wkr, wvr = seeded(seed + 22, C, C, scale=C ** -0.5), seeded(seed + 23, C, C, scale=C ** -0.5)
    name = "blk.attn1.processor"
    if rank:
        proc = ap.LoraRefSAttnProcessor2_0(name, C, rank=rank, lora_scale=lora_scale, scale=scale)
        lw = lora_weights(seed, C, C, rank)
        set_lora(proc, lw)
    else:
        proc = ap.RefSAttnProcessor2_0(name, C, scale=scale)
        lw = None
    proc.to_k_ref.weight.copy_(wkr); proc.to_v_ref.weight.copy_(wvr)
    # the reference is only defined for B == 1 with a 1-batch garment (:602-603): run per sample
    cond = torch.cat([proc(attn, x[b:b + 1], sa_hidden_states={name: ref}) for b in range(B)])
    uncond = torch.cat([proc(attn, x[b:b + 1]) for b in range(B)])
    rows = None
    if keep_rows:            # full-size cases keep a seeded subset of the N output rows (plus the spiked ones)
        import numpy as np
        rows = np.sort(np.random.default_rng(seed + 99).choice(N, keep_rows, replace=False))
        if spike:
            rows = np.unique(np.concatenate([rows, [spike[0]]]))
        rows = torch.from_numpy(rows)
        cond, uncond = cond[:, rows].clone(), uncond[:, rows].clone()
    case = dict(kind="hybrid", seed=seed, B=B, N=N, M=M, C=C, heads=heads, scale=scale, rank=rank,
                lora_scale=lora_scale, out_cond=cond, out_uncond=uncond, spike=spike, rows=rows,
                digests=dict(x=digest(x), ref=digest(ref), wq=digest(w["wq"]), wkr=digest(wkr)))
    if store_full:
        case.update(x=x, ref=ref, wk_ref=wkr, wv_ref=wvr, lora=lw, **w)
    return case


@torch.no_grad()
def cross_case(ap, seed, B, N, T, C, KD, heads, ip_tokens=0, ip_scale=1.0, rank=0, lora_scale=0.0, store_full=True):
    w = attn_weights(seed, C, KD)
    attn = make_attn(w, C, KD, heads)
    x = seeded(seed + 20, B, N, C)
    ehs = seeded(seed + 21, B, T + ip_tokens, KD, scale=0.5)
    case = dict(kind="cross", seed=seed, B=B, N=N, T=T, C=C, KD=KD, heads=heads, ip_tokens=ip_tokens,
                ip_scale=ip_scale, rank=rank, lora_scale=lora_scale,
                digests=dict(x=digest(x), ehs=digest(ehs), wq=digest(w["wq"])))
    lw = wkip = wvip = None
    if ip_tokens:
        proc = ap.LoRAIPAttnProcessor2_0(C, KD, rank=rank, lora_scale=lora_scale, scale=ip_scale, num_tokens=ip_tokens)
        lw = lora_weights(seed, C, KD, rank)
        set_lora(proc, lw)
        wkip, wvip = seeded(seed + 30, C, KD, scale=KD ** -0.5), seeded(seed + 31, C, KD, scale=KD ** -0.5)
        proc.to_k_ip.weight.copy_(wkip); proc.to_v_ip.weight.copy_(wvip)
        out = proc(attn, x, encoder_hidden_states=ehs)
    else:
        proc = ap.CAttnProcessor2_0("blk.attn2.processor", C, KD)
        out = proc(attn, x, encoder_hidden_states=ehs, sa_hidden_states={"unused": None})
    case["out"] = out
    if store_full:
        case.update(x=x, ehs=ehs, lora=lw, wk_ip=wkip, wv_ip=wvip, **w)
    return case


@torch.no_grad()
def cache_case(ap, seed, B, N, C, heads, store_full=True, T=0, KD=0):
    """``CacheAttnProcessor2_0`` as self-attention (T == 0) or as cross-attention over T context tokens of width KD
    (the garment UNet's attn2 layers read the 16 resampler tokens, IMAGDressing_v1_pipeline.py:466-470)."""
    w = attn_weights(seed, C, KD or C)
    attn = make_attn(w, C, KD or C, heads)
    x = seeded(seed + 20, B, N, C)
    ehs = seeded(seed + 21, B, T, KD, scale=0.5) if T else None
    proc = ap.CacheAttnProcessor2_0()
    out = proc(attn, x, encoder_hidden_states=ehs)
    assert proc.cache["hidden_states"] is x            # stores its input (:34)
    case = dict(kind="cache", seed=seed, B=B, N=N, C=C, heads=heads, T=T, KD=KD, out=out,
                digests=dict(x=digest(x), wq=digest(w["wq"])))
    if store_full:
        case.update(x=x, ehs=ehs, **w)
    return case


def resampler_sd(seed, dim, depth, dim_head, heads, nq, emb, out, ff_mult=4, prefix=""):
    inner = dim_head * heads
    sd = {}
    k = [seed]

    def nxt():
        k[0] += 1
        return k[0]

    def lin(name, o, i, bias):
        sd[prefix + name + ".weight"] = seeded(nxt(), o, i, scale=i ** -0.5)
        if bias:
            sd[prefix + name + ".bias"] = seeded(nxt(), o, scale=0.05)

    def ln(name, d):
        sd[prefix + name + ".weight"] = 1.0 + seeded(nxt(), d, scale=0.1)
        sd[prefix + name + ".bias"] = seeded(nxt(), d, scale=0.05)

    if nq:
        sd[prefix + "latents"] = seeded(nxt(), 1, nq, dim, scale=dim ** -0.5)
    lin("proj_in", dim, emb, True)
    lin("proj_out", out, dim, True)
    ln("norm_out", out)
    for i in range(depth):
        p = f"layers.{i}.0"
        ln(p + ".norm1", dim); ln(p + ".norm2", dim)
        lin(p + ".to_q", inner, dim, False); lin(p + ".to_kv", 2 * inner, dim, False); lin(p + ".to_out", dim, inner, False)
        f = f"layers.{i}.1"
        ln(f + ".0", dim); lin(f + ".1", dim * ff_mult, dim, False); lin(f + ".3", dim, dim * ff_mult, False)
    return sd


@torch.no_grad()
def resampler_case(rs, seed, B, L, cfg, store_full):
    sd = resampler_sd(seed, cfg["dim"], cfg["depth"], cfg["dim_head"], cfg["heads"], cfg["num_queries"],
                      cfg["embedding_dim"], cfg["output_dim"])
    m = rs.Resampler(**cfg)
    m.load_state_dict(sd, strict=True)
    x = seeded(seed + 1000, B, L, cfg["embedding_dim"], scale=0.5)
    out = m(x)
    case = dict(kind="resampler", seed=seed, B=B, L=L, cfg=cfg, out=out, digests=dict(x=digest(x), latents=digest(sd["latents"])))
    if store_full:
        case.update(x=x, sd=sd)
    return case


def proj_plus_sd(seed, cad=768, idd=512, clipd=1280, ntok=4):
    sd = resampler_sd(seed, cad, 4, 64, cad // 64, 0, clipd, cad, prefix="perceiver_resampler.")
    sd["proj.0.weight"] = seeded(seed + 501, idd * 2, idd, scale=idd ** -0.5)
    sd["proj.0.bias"] = seeded(seed + 502, idd * 2, scale=0.05)
    sd["proj.2.weight"] = seeded(seed + 503, cad * ntok, idd * 2, scale=(idd * 2) ** -0.5)
    sd["proj.2.bias"] = seeded(seed + 504, cad * ntok, scale=0.05)
    sd["norm.weight"] = 1.0 + seeded(seed + 505, cad, scale=0.1)
    sd["norm.bias"] = seeded(seed + 506, cad, scale=0.05)
    return sd


@torch.no_grad()
def proj_plus_case(rs, seed):
    sd = proj_plus_sd(seed)
    m = rs.ProjPlusModel()
    m.load_state_dict(sd, strict=True)
    idv = seeded(seed + 2000, 1, 512)
    clip = seeded(seed + 2001, 1, 257, 1280, scale=0.5)
    out = m(idv, clip)                                   # default shortcut=False (..._ipa_controlnet.py:375)
    out_sc = m(idv, clip, shortcut=True, scale=0.7)
    return dict(kind="proj_plus", seed=seed, out=out, out_shortcut=out_sc, digests=dict(id=digest(idv), clip=digest(clip)))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ap, rs = load_reference_adapter()
    os.makedirs(OUT, exist_ok=True)
    proc_cases = {
        # full (small) cases
        "hybrid_small": hybrid_case(ap, 100, B=2, N=48, M=80, C=64, heads=8, scale=0.9),
        "hybrid_small_lora": hybrid_case(ap, 200, B=1, N=40, M=24, C=64, heads=8, scale=1.0, rank=16, lora_scale=0.2),
        "cross_small": cross_case(ap, 300, B=2, N=48, T=77, C=64, KD=96, heads=8),
        "cross_small_ip": cross_case(ap, 400, B=2, N=48, T=77, C=64, KD=96, heads=8, ip_tokens=4, ip_scale=0.9, rank=16, lora_scale=0.2),
        "cache_small": cache_case(ap, 500, B=2, N=32, C=64, heads=8),
        # seeded (real SD1.5 head dims; ragged N/M that are not multiples of any tile)
        "hybrid_d40": hybrid_case(ap, 1000, B=2, N=200, M=330, C=320, heads=8, scale=1.0, store_full=False),
        "hybrid_d80": hybrid_case(ap, 1100, B=1, N=144, M=100, C=640, heads=8, scale=0.8, store_full=False),
        "hybrid_d160": hybrid_case(ap, 1200, B=1, N=64, M=80, C=1280, heads=8, scale=1.0, store_full=False),
        "hybrid_d40_lora": hybrid_case(ap, 1300, B=1, N=96, M=128, C=320, heads=8, scale=0.9, rank=128, lora_scale=0.2, store_full=False),
        "cross_d40": cross_case(ap, 1400, B=2, N=130, T=77, C=320, KD=768, heads=8, store_full=False),
        "cross_d160_ip": cross_case(ap, 1500, B=1, N=64, T=77, C=1280, KD=768, heads=8, ip_tokens=4, ip_scale=0.9, rank=128, lora_scale=0.2, store_full=False),
    }
    torch.save(proc_cases, os.path.join(OUT, "processors.pt"))
    small_cfg = dict(dim=64, depth=2, dim_head=16, heads=4, num_queries=4, embedding_dim=48, output_dim=32, ff_mult=4)
    real_cfg = dict(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4)
    res_cases = {
        "resampler_small": resampler_case(rs, 3000, B=2, L=19, cfg=small_cfg, store_full=True),
        "resampler_real": resampler_case(rs, 4000, B=2, L=257, cfg=real_cfg, store_full=False),   # inference_IMAGdressing.py:55-64
        "proj_plus_real": proj_plus_case(rs, 5000),
    }
    torch.save(res_cases, os.path.join(OUT, "resampler.pt"))
    for f in ("processors.pt", "resampler.pt"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


def main_full():
    """Second fixture file: the reference source at the BENCHMARKED shape of the fused kernel (BASELINE.json configs[1]
    level 0: N = M = 4096, C = 320, d = 40 -> the two-query-block instantiation), a ragged N >= 512 case with late score
    spikes (forces the kernel's deferred-max redo path), and ``CacheAttnProcessor2_0`` at a real head dim."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ap, _ = load_reference_adapter()
    cases = {
        "hybrid_d40_n4096": hybrid_case(ap, 2000, B=1, N=4096, M=4096, C=320, heads=8, scale=1.0, store_full=False, keep_rows=384),
        "hybrid_d40_spike": hybrid_case(ap, 2100, B=1, N=840, M=700, C=320, heads=8, scale=0.9, store_full=False,
                                        spike=(801, 650, 4.0), keep_rows=256),
        "cache_d40": cache_case(ap, 2200, B=1, N=200, C=320, heads=8, store_full=False),
        "cache_d80_cross": cache_case(ap, 2300, B=1, N=144, C=640, heads=8, store_full=False, T=16, KD=768),
    }
    torch.save(cases, os.path.join(OUT, "processors_full.pt"))
    print("processors_full.pt", os.path.getsize(os.path.join(OUT, "processors_full.pt")) // 1024, "KiB")


def main_geometry():
    """Third fixture file: the reference source at the token counts of the reference scripts' OWN default geometry --
    width 512 x height 640, garment 640 x 512 (inference_IMAGdressing.py:182-183; latent 80 x 64) -- i.e. the four UNet
    levels (C, N = M) = (320, 5120), (640, 1280), (1280, 320), (1280, 80) a user of the unchanged scripts hits first.
    Level 0 takes the software-pipelined d = 40 kernel (N >= 512), the others the generic kernel at d = 80 / 160."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ap, _ = load_reference_adapter()
    cases = {
        "hybrid_d40_n5120": hybrid_case(ap, 2400, B=1, N=5120, M=5120, C=320, heads=8, scale=1.0, store_full=False, keep_rows=384),
        "hybrid_d80_n1280": hybrid_case(ap, 2500, B=1, N=1280, M=1280, C=640, heads=8, scale=1.0, store_full=False, keep_rows=256),
        "hybrid_d160_n320": hybrid_case(ap, 2600, B=1, N=320, M=320, C=1280, heads=8, scale=1.0, store_full=False, keep_rows=128),
        "hybrid_d160_n80": hybrid_case(ap, 2700, B=1, N=80, M=80, C=1280, heads=8, scale=1.0, store_full=False, keep_rows=80),
    }
    torch.save(cases, os.path.join(OUT, "processors_default_geometry.pt"))
    print("processors_default_geometry.pt", os.path.getsize(os.path.join(OUT, "processors_default_geometry.pt")) // 1024, "KiB")


@torch.no_grad()
def legacy_case(ap, kind, seed, B, N, M, C, heads, T=0, KD=0, scale=0.9, keep_rows=0):
    """``SAttnProcessor2_0`` (kind "sattn": one softmax over [self; garment] keys) / ``RefCAttnProcessor2_0`` (kind "refc": text
    cross-attention + garment softmax) of the reference, run per sample (their ``view(batch_size, ...)`` / ``cat`` need B == 1 with
    a [1, M, C] garment).  Seeded: inputs are regenerated by tests (tests/cases.py::legacy_inputs)."""
    kd = KD or C
    w = attn_weights(seed, C, kd)
    attn = make_attn(w, C, kd, heads)
    x = seeded(seed + 20, B, N, C)
    ref = seeded(seed + 21, 1, M, C)
    ehs = seeded(seed + 24, B, T, KD, scale=0.5) if T else None
    name = "blk.attn.processor"
    wkr = wvr = None
    if kind == "sattn":
        proc = ap.SAttnProcessor2_0(name, C)
        run = lambda xb, eb, sa: proc(attn, xb, sa_hidden_states=sa)          # noqa: E731
    else:
        proc = ap.RefCAttnProcessor2_0(name, C, KD or None, scale=scale)
        wkr, wvr = seeded(seed + 22, C, C, scale=C ** -0.5), seeded(seed + 23, C, C, scale=C ** -0.5)
        proc.to_k_ref.weight.copy_(wkr); proc.to_v_ref.weight.copy_(wvr)
        run = lambda xb, eb, sa: proc(attn, xb, encoder_hidden_states=eb, sa_hidden_states=sa)     # noqa: E731
    cond = torch.cat([run(x[b:b + 1], None if ehs is None else ehs[b:b + 1], {name: ref}) for b in range(B)])
    plain = torch.cat([run(x[b:b + 1], None if ehs is None else ehs[b:b + 1], None) for b in range(B)])
    rows = None
    if keep_rows:
        import numpy as np
        rows = torch.from_numpy(np.sort(np.random.default_rng(seed + 99).choice(N, keep_rows, replace=False)))
        cond, plain = cond[:, rows].clone(), plain[:, rows].clone()
    return dict(kind=kind, seed=seed, B=B, N=N, M=M, C=C, heads=heads, T=T, KD=KD, scale=scale, out_garment=cond, out_plain=plain, rows=rows,
                digests=dict(x=digest(x), ref=digest(ref), wq=digest(w["wq"]), ehs=None if ehs is None else digest(ehs)))


def main_legacy():
    """Seventh fixture file: the two processor classes no reference entry point installs but the module exports
    (attention_processor.py:103-199, :630-743), in their garment forms."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ap, _ = load_reference_adapter()
    cases = {
        "sattn_d40": legacy_case(ap, "sattn", 6000, B=2, N=200, M=330, C=320, heads=8),
        "sattn_d40_n640": legacy_case(ap, "sattn", 6100, B=1, N=640, M=700, C=320, heads=8, keep_rows=256),     # N >= 512: the pipelined d = 40 kernel
        "sattn_d160": legacy_case(ap, "sattn", 6200, B=1, N=64, M=80, C=1280, heads=8),
        "refc_d40": legacy_case(ap, "refc", 6300, B=2, N=130, M=96, C=320, heads=8, T=77, KD=768),
        "refc_d80_self": legacy_case(ap, "refc", 6400, B=1, N=144, M=100, C=640, heads=8, scale=0.8),           # encoder_hidden_states None (:681-682)
    }
    torch.save(cases, os.path.join(OUT, "processors_legacy.pt"))
    print("processors_legacy.pt", os.path.getsize(os.path.join(OUT, "processors_legacy.pt")) // 1024, "KiB")


@torch.no_grad()
def main_unet():
    """Fourth fixture file: ONE full-width (859.5 M parameters) SD1.5 UNet forward of the fp32 oracle (oracle/sd15.py -- the
    restated, UNPINNED diffusers-0.24 UNet -- carrying oracle/processors.py, which IS pinned on the reference source) at
    t = 481 on seeded weights and inputs, uncond (no garment) and cond (garment tokens on all 16 attn1 layers), for the 64x64
    latent of BASELINE configs[1] and the 80x64 latent of the reference scripts' default 512x640 geometry.  bench.py
    measures its `parity` field against these outputs inside the bench process; tests/test_fullsize_gpu.py uses them too.
    Inputs are regenerated from seeds by tests/cases.py::unet_forward_inputs and digest-checked."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from imagdressing_amd import unet as E
    from tests.cases import unet_forward_inputs
    from tests.harness_names import hidden_size_of
    from . import processors as OP
    from . import sd15
    torch.set_num_threads(8)
    cases = {}
    boc = E.SD15_CONFIG["block_out_channels"]
    for key, (lh, lw) in (("latent_64x64", (64, 64)), ("latent_80x64", (80, 64))):
        d = unet_forward_inputs(lh, lw)
        o = sd15.UNet2DConditionModel({})
        o.load_state_dict(d["sd"], strict=True)
        assert list(o.attn_processors.keys()) == __import__("tests.harness_names", fromlist=["x"]).attn_processor_names(E.SD15_CONFIG)
        o.set_attn_processor({n: (OP.RefSAttn(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                                  else OP.CAttn(n, hidden_size_of(n, boc), 768)) for n in o.attn_processors.keys()})
        for n in d["names"]:
            o.attn_processors[n].to_k_ref.weight.copy_(d["rw"][n]["k"]); o.attn_processors[n].to_v_ref.weight.copy_(d["rw"][n]["v"])
        unc = o(d["x"], 481, d["ehs"])
        cond = o(d["x"], 481, d["ehs"], cross_attention_kwargs={"sa_hidden_states": d["sa"]})
        cases[key] = dict(kind="unet_forward", lh=lh, lw=lw, t=481, out_uncond=unc.clone(), out_cond=cond.clone(), digests=d["digests"],
                          torch_version=torch.__version__)
        print(key, "uncond std", unc.std().item(), "cond-uncond rel rms", ((cond - unc).pow(2).mean().sqrt() / unc.pow(2).mean().sqrt()).item())
        del o
    torch.save(cases, os.path.join(OUT, "unet_forward_full.pt"))
    print("unet_forward_full.pt", os.path.getsize(os.path.join(OUT, "unet_forward_full.pt")) // 1024, "KiB")


@torch.no_grad()
def main_timesteps():
    """Fifth fixture file: full-width fp32-oracle UNet forwards at the two ENDS of the DDIM schedule (t = 981: eps ~ z, the first step of
    the 50-step run; t = 1: the last) for the configs[1] processors (t = 481 is in unet_forward_full.pt), and at t = 981 / 481 / 1 for the
    configs[2] stack -- LoraRefS + LoRAIP processors (rank 128, 77 text + 4 face tokens) with the pose ControlNet's residuals added,
    the ControlNet run on the CFG pair in the reference's order (..._ipa_controlnet.py:651-666: row [1] -> cond, row [0] -> uncond).
    bench.py's `parity` leg and tests/test_fullsize_gpu.py measure the HIP engine against these (tests/unet_fixture.py)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from imagdressing_amd import unet as E
    from tests.harness_names import hidden_size_of
    from tests.unet_fixture import fill_ipa_processors, ipa_controlnet_forward_inputs, unet_forward_inputs
    from . import processors as OP
    from . import sd15
    torch.set_num_threads(8)
    boc = E.SD15_CONFIG["block_out_channels"]
    cases = {}
    # ---- configs[1] processors at the ends of the schedule ----
    d = unet_forward_inputs(64, 64)
    o = sd15.UNet2DConditionModel({})
    o.load_state_dict(d["sd"], strict=True)
    o.set_attn_processor({n: (OP.RefSAttn(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                              else OP.CAttn(n, hidden_size_of(n, boc), 768)) for n in o.attn_processors.keys()})
    for n in d["names"]:
        o.attn_processors[n].to_k_ref.weight.copy_(d["rw"][n]["k"]); o.attn_processors[n].to_v_ref.weight.copy_(d["rw"][n]["v"])
    refs = dict(kind="unet_forward_timesteps", lh=64, lw=64, digests=d["digests"], torch_version=torch.__version__)
    for t in (981, 1):
        unc = o(d["x"], t, d["ehs"])
        cond = o(d["x"], t, d["ehs"], cross_attention_kwargs={"sa_hidden_states": d["sa"]})
        refs[f"t{t}"] = dict(out_uncond=unc.clone(), out_cond=cond.clone())
        print("refs t", t, "uncond std", unc.std().item())
    cases["refs"] = refs
    del o
    # ---- configs[2] stack at t = 981 / 481 / 1 ----
    d = ipa_controlnet_forward_inputs()
    o = sd15.UNet2DConditionModel({})
    o.load_state_dict(d["sd"], strict=True)
    procs = {n: (OP.LoraRefSAttn(n, hidden_size_of(n, boc), scale=d["ref_scale"], rank=d["rank"], lora_scale=d["lora_scale"])
                 if n.endswith("attn1.processor") else
                 OP.LoRAIPAttn(hidden_size_of(n, boc), 768, rank=d["rank"], lora_scale=d["lora_scale"], scale=d["ip_scale"], num_tokens=4))
             for n in o.attn_processors.keys()}
    fill_ipa_processors(procs, d)
    o.set_attn_processor(procs)
    c = sd15.ControlNetModel({})
    c.load_state_dict(d["ctrl_sd"], strict=True)
    ipa = dict(kind="unet_forward_ipa_controlnet", lh=64, lw=64, digests=d["digests"], torch_version=torch.__version__)
    lmi = torch.cat([d["x"]] * 2)
    pec = torch.cat([d["ehs_u_text"], d["ehs"]])                 # [negative; prompt], text only (..._ipa_controlnet.py:550)
    for t in (981, 481, 1):
        down, mid = c(lmi, t, pec, d["pose"], 1.0)
        cond = o(d["x"], t, d["ehs_c"], cross_attention_kwargs={"sa_hidden_states": d["sa"]},
                 down_block_additional_residuals=[r[1:2] for r in down], mid_block_additional_residual=mid[1:2])
        unc = o(d["x"], t, d["ehs_u"], down_block_additional_residuals=[r[0:1] for r in down], mid_block_additional_residual=mid[0:1])
        plain = o(d["x"], t, d["ehs_u"])
        ipa[f"t{t}"] = dict(out_uncond=unc.clone(), out_cond=cond.clone())
        print("ipa t", t, "uncond std", unc.std().item(), "controlnet effect rel rms",
              ((unc - plain).pow(2).mean().sqrt() / plain.pow(2).mean().sqrt()).item())
    cases["ipa_controlnet"] = ipa
    torch.save(cases, os.path.join(OUT, "unet_forward_timesteps.pt"))
    print("unet_forward_timesteps.pt", os.path.getsize(os.path.join(OUT, "unet_forward_timesteps.pt")) // 1024, "KiB")


@torch.no_grad()
def main_trajectory(which=None):
    """Sixth fixture file (tests/golden/trajectory.pt): the reference's sampling loop END TO END (oracle/pipeline.py::denoise restating
    IMAGDressing_v1_pipeline.py:463-541, ..._ipa_controlnet.py:595-736, ..._controlnet_inpainting.py:387-517) on the full-width fp32
    oracle: Resampler (pinned) -> garment UNet pass at t = 0 -> every DDIM step with the two batch-1 UNet calls and the custom CFG ->
    final latent, one image at a time.  Cases and seeds: tests/trajectory_fixture.py::CASES (BASELINE configs[0] 20 steps, configs[1]
    50 steps x seeds 42 / 43, configs[2] and configs[4] 10 steps).  Stored: the final latent and the latents after a few steps
    (64 KB each) plus input digests; the GPU tests and bench.py's ``parity.trajectory`` leg rebuild the inputs from seeds.
    oracle/sd15.py + ddim.py are the UNPINNED restatement of diffusers 0.24; processors / resampler are pinned on the reference."""
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from imagdressing_amd import unet as E
    from tests.harness_names import hidden_size_of
    from tests.trajectory_fixture import CASES, FILE, initial_latents, trajectory_inputs
    from tests.unet_fixture import fill_ipa_processors
    from . import processors as OP
    from . import sd15
    from .ddim import DDIMOracle
    from .pipeline import denoise
    from .resampler import resampler_forward
    torch.set_num_threads(8)
    boc = E.SD15_CONFIG["block_out_channels"]
    out = torch.load(FILE, weights_only=False) if os.path.isfile(FILE) else {}
    built = {}
    for name, spec in CASES.items():
        if which and name not in which:
            continue
        kind, lh, lw = spec["kind"], spec["lh"], spec["lw"]
        key = (kind, lh, lw)
        if key not in built:
            built.clear()
            d = trajectory_inputs(kind, lh, lw)
            o = sd15.UNet2DConditionModel({}); o.load_state_dict(d["sd"], strict=True)
            r = sd15.UNet2DConditionModel({}); r.load_state_dict(d["sd_ref"], strict=True)
            r.set_attn_processor({n: OP.CacheAttn() for n in r.attn_processors.keys()})
            if kind == "ipa_controlnet":
                procs = {n: (OP.LoraRefSAttn(n, hidden_size_of(n, boc), scale=d["ref_scale"], rank=d["rank"], lora_scale=d["lora_scale"])
                             if n.endswith("attn1.processor") else
                             OP.LoRAIPAttn(hidden_size_of(n, boc), 768, rank=d["rank"], lora_scale=d["lora_scale"], scale=d["ip_scale"], num_tokens=4))
                         for n in o.attn_processors.keys()}
                fill_ipa_processors(procs, d)
            else:
                procs = {n: (OP.RefSAttn(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                             else OP.CAttn(n, hidden_size_of(n, boc), 768)) for n in o.attn_processors.keys()}
                for n in d["names"]:
                    procs[n].to_k_ref.weight.copy_(d["rw"][n]["k"]); procs[n].to_v_ref.weight.copy_(d["rw"][n]["v"])
            o.set_attn_processor(procs)
            c = None
            if kind != "refs":
                c = sd15.ControlNetModel({}); c.load_state_dict(d["ctrl_sd"], strict=True)
            built[key] = (d, o, r, c)
        d, o, r, c = built[key]
        rsd = d["resampler_sd"]
        cloth = torch.cat([resampler_forward(rsd, torch.zeros_like(d["clip"]), 12), resampler_forward(rsd, d["clip"], 12)])   # [null; proj] (:409-433)
        lat = initial_latents(spec["seeds"], lh, lw)
        finals, kept = [], {k: [] for k in spec["keep"]}
        for si, seed in enumerate(spec["seeds"]):
            t0 = time.time()
            trace = []
            kw = {}
            pe, ne = d["pe"], d["ne"]
            if kind == "ipa_controlnet":
                pe, ne = d["ehs_c"], d["ehs_u"]
                kw = dict(controlnet=c, control_image=d["pose"], prompt_embeds_control=torch.cat([d["ne"], d["pe"]]),
                          conditioning_scale=spec["conditioning_scale"])
            elif kind == "inpaint":
                kw = dict(controlnet=c, control_image=d["control_image"], prompt_embeds_control=torch.cat([d["ne"], d["pe"]]),
                          conditioning_scale=spec["conditioning_scale"],
                          inpaint=dict(mask=d["mask"], image_latents=d["img_lat"], noise=lat[si:si + 1]))
            fin = denoise(o, r, DDIMOracle(), lat[si:si + 1], pe, ne, cloth, d["refl"], spec["steps"], spec["guidance"], trace=trace, **kw)
            finals.append(fin.clone())
            for k in spec["keep"]:
                kept[k].append(trace[k].clone())
            print(name, "seed", seed, "final std", fin.std().item(), "finite", bool(torch.isfinite(fin).all()), f"{time.time() - t0:.0f} s", flush=True)
        out[name] = dict(kind="trajectory", spec=dict(spec), final=torch.cat(finals), steps={k: torch.cat(v) for k, v in kept.items()},
                         digests=d["traj_digests"], torch_version=torch.__version__)
        torch.save(out, FILE)
    print("trajectory.pt", os.path.getsize(FILE) // 1024, "KiB")


if __name__ == "__main__":
    import sys
    what = sys.argv[1] if len(sys.argv) > 1 else "all"          # base | full | geometry | unet | timesteps | all
    if what in ("base", "all"):
        main()
    if what in ("full", "all"):
        main_full()
    if what in ("geometry", "all"):
        main_geometry()
    if what in ("legacy", "all"):
        main_legacy()
    if what in ("unet", "all"):
        main_unet()
    if what in ("timesteps", "all"):
        main_timesteps()
    if what == "trajectory":          # ~30 min of 8 cores; not part of "all"
        main_trajectory(sys.argv[2:] or None)
