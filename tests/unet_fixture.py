"""The committed full-width UNet-forward fixture (tests/golden/unet_forward_full.pt, made by oracle/make_golden.py::main_unet):
seeded-input builder and the parity probe that bench.py and the GPU tests run against it.  TEST INFRASTRUCTURE; imports no
oracle code (the fixture file holds the fp32 oracle's outputs), so bench.py may use it outside its cpu_baseline leg."""
import hashlib
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(t: torch.Tensor) -> str:          # == oracle.seeds.digest
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]

def unet_forward_inputs(lh, lw, case=None):
    """Seeded inputs of the committed full-width UNet-forward fixture: SD1.5 state dict (seed 0, the synthetic init every
    full-size test uses), latent [1, 4, lh, lw], text states [1, 77, 768], garment tokens of the 16 attn1 layers
    ([1, M_l, C_l], garment at the generation resolution) and to_k_ref / to_v_ref (seed 7).  torch CPU generators ->
    identical on every host; when ``case`` (the fixture entry) is given its digests are checked.  Imports no oracle code."""
    from imagdressing_amd import unet as E
    from tests.harness_names import attn1_names, hidden_size_of

    def rnd(seed, *shape, scale=1.0):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
    cfg = E.SD15_CONFIG
    boc = cfg["block_out_channels"]
    sd = E.random_state_dict(E.unet_param_shapes(cfg), 0)
    names = attn1_names(cfg)
    g = torch.Generator().manual_seed(7)
    rw = {}
    for n in names:
        c = hidden_size_of(n, boc)
        rw[n] = dict(k=torch.randn(c, c, generator=g) * c ** -0.5, v=torch.randn(c, c, generator=g) * c ** -0.5)
    tokens = {320: lh * lw, 640: lh * lw // 4, 1280: lh * lw // 16}
    sa = {}
    for j, n in enumerate(names):
        c = hidden_size_of(n, boc)
        m = lh * lw // 64 if n.startswith("mid_block") else tokens[c]
        sa[n] = rnd(100 + j, 1, m, c)
    d = dict(sd=sd, x=rnd(1, 1, 4, lh, lw), ehs=rnd(2, 1, 77, 768, scale=0.5), rw=rw, sa=sa, names=names)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(digest(sd[k]).encode())
    d["digests"] = dict(x=digest(d["x"]), ehs=digest(d["ehs"]), sd=h.hexdigest()[:16], sa0=digest(sa[names[0]]),
                        rw0=digest(rw[names[0]]["k"]))
    if case is not None:
        assert d["digests"] == case["digests"], ("regenerated inputs differ from the fixture's", d["digests"], case["digests"])
    return d


@torch.no_grad()
def measure_unet_parity(device, dtype, key="latent_64x64", inputs=None):
    """One full-width cond + uncond UNet forward of the HIP engine (CFG layout: [cond; uncond] rows in one call, garment switched
    per row) in ``dtype`` against the committed fp32-oracle outputs.  -> dict(max_abs, rel_rms, ref_std, meets_atol_1e-2) per
    pass.  The bar the north star names is atol 1e-2 on the UNet output (eps, std 0.56)."""
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter import attention_processor as AP
    from tests.harness_names import hidden_size_of
    case = torch.load(os.path.join(GOLDEN, "unet_forward_full.pt"), weights_only=False)[key]
    d = inputs if inputs is not None else unet_forward_inputs(case["lh"], case["lw"], case)
    boc = E.SD15_CONFIG["block_out_channels"]
    e = E.UNet2DConditionModel(d["sd"], {}, str(device), dtype)
    e.set_attn_processor({n: (AP.RefSAttnProcessor2_0(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                              else AP.CAttnProcessor2_0(n, hidden_size_of(n, boc), 768)) for n in e.attn_processors.keys()})
    for n in d["names"]:
        p = e.attn_processors[n]
        p.to_k_ref.weight.copy_(d["rw"][n]["k"]); p.to_v_ref.weight.copy_(d["rw"][n]["v"])
    sa = {n: t.to(device) for n, t in d["sa"].items()}
    x2 = torch.cat([d["x"], d["x"]]).to(device)
    both = e(x2, case["t"], d["ehs"].to(device), cross_attention_kwargs={
        "sa_hidden_states": sa, "sa_batch_mask": torch.tensor([1.0, 0.0], device=device)})[0].float().cpu()
    out = {}
    for nm, got, ref in (("cond", both[0:1], case["out_cond"]), ("uncond", both[1:2], case["out_uncond"])):
        err = (got - ref).abs()
        out[nm] = dict(max_abs=round(err.max().item(), 5), rel_rms=round((err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), 5),
                       ref_std=round(ref.std().item(), 4), frac_within_1e2=round((err <= 1e-2).float().mean().item(), 5))
    out["meets_atol_1e-2"] = bool(max(out["cond"]["max_abs"], out["uncond"]["max_abs"]) <= 1e-2)
    out["finite"] = bool(torch.isfinite(both).all())
    del e
    torch.cuda.empty_cache()
    return out
