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


def ipa_controlnet_forward_inputs(case=None, base=None):
    """Seeded inputs of the configs[2]-shaped full-width forward fixture (tests/golden/unet_forward_timesteps.pt, 64x64 latent):
    everything :func:`unet_forward_inputs` builds, plus the pose-ControlNet state dict (seed 2, zero-convs at 0.1 of fan-in scale so the
    residuals matter), rank-128 LoRA factors for q / k / v / out of all 32 processors and to_k_ip / to_v_ip of the 16 cross-attention
    ones (one generator, seed 107, walked in processor-name order), the 77 + 4 token encoder states of both CFG halves and the pose
    image.  Imports no oracle code."""
    from imagdressing_amd import unet as E
    from tests.harness_names import attn_processor_names, hidden_size_of

    def rnd(seed, *shape, scale=1.0):
        return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
    cfg = E.SD15_CONFIG
    boc, cd, rank = cfg["block_out_channels"], cfg["cross_attention_dim"], 128
    d = dict(base) if base is not None else unet_forward_inputs(64, 64)      # (the 859.5 M-parameter state dict is shared, not copied)
    d["ctrl_sd"] = E.random_state_dict(E.controlnet_param_shapes(cfg), 2, zero_convs=True)
    g = torch.Generator().manual_seed(107)
    d["all_names"] = attn_processor_names(cfg)
    d["lora"], d["ip"] = {}, {}
    for n in d["all_names"]:
        c = hidden_size_of(n, boc)
        kd = c if n.endswith("attn1.processor") else cd
        d["lora"][n] = {nm: (torch.randn(rank, cin, generator=g) * cin ** -0.5, torch.randn(c, rank, generator=g) * rank ** -0.5)
                        for nm, cin in (("q", c), ("k", kd), ("v", kd), ("out", c))}
        if not n.endswith("attn1.processor"):
            d["ip"][n] = (torch.randn(c, cd, generator=g) * cd ** -0.5, torch.randn(c, cd, generator=g) * cd ** -0.5)
    d["ehs_c"] = torch.cat([d["ehs"], rnd(14, 1, 4, 768, scale=0.5)], dim=1)              # prompt + 4 face tokens (..._ipa_controlnet.py:550-557)
    d["ehs_u_text"] = rnd(3, 1, 77, 768, scale=0.5)
    d["ehs_u"] = torch.cat([d["ehs_u_text"], rnd(15, 1, 4, 768, scale=0.5)], dim=1)
    d["pose"] = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(16))
    d["rank"], d["lora_scale"], d["ip_scale"], d["ref_scale"] = rank, 0.2, 0.9, 0.9
    first2 = d["all_names"][1]
    d["digests"] = dict(d["digests"], ctrl=digest(d["ctrl_sd"]["controlnet_mid_block.weight"]), lora0=digest(d["lora"][d["all_names"][0]]["q"][0]),
                        ip0=digest(d["ip"][first2][0]), ehs_c=digest(d["ehs_c"]), ehs_u=digest(d["ehs_u"]), pose=digest(d["pose"]))
    if case is not None:
        assert d["digests"] == case["digests"], ("regenerated inputs differ from the fixture's", d["digests"], case["digests"])
    return d


@torch.no_grad()
def fill_ipa_processors(procs, d):
    """Copy the fixture's garment / LoRA / IP weights into a name -> processor dict (the engine's classes and the oracle's carry the
    same attribute names: to_k_ref, to_{q,k,v,out}_lora.{down,up}, to_{k,v}_ip)."""
    for n, p in procs.items():
        for nm, (down, up) in d["lora"][n].items():
            layer = getattr(p, f"to_{nm}_lora")
            layer.down.weight.copy_(down); layer.up.weight.copy_(up)
        if n.endswith("attn1.processor"):
            p.to_k_ref.weight.copy_(d["rw"][n]["k"]); p.to_v_ref.weight.copy_(d["rw"][n]["v"])
        else:
            p.to_k_ip.weight.copy_(d["ip"][n][0]); p.to_v_ip.weight.copy_(d["ip"][n][1])


def _err_entry(got, ref):
    err = (got - ref).abs()
    return dict(max_abs=round(err.max().item(), 5), rel_rms=round((err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), 5),
                ref_std=round(ref.std().item(), 4), frac_within_1e2=round((err <= 1e-2).float().mean().item(), 5))


@torch.no_grad()
def measure_unet_parity_timesteps(device, dtype, inputs=None, ipa_inputs=None, which=("refs", "ipa_controlnet")):
    """Full-width cond + uncond forwards of the HIP engine at t = 981, 481 and 1 against the committed fp32-oracle outputs:
      refs            RefS + CAttn processors (configs[1]; t = 481 from unet_forward_full.pt, 981 / 1 from unet_forward_timesteps.pt)
      ipa_controlnet  LoraRefS + LoRAIP processors (rank 128, 77 + 4 tokens) with the pose ControlNet's residuals added (configs[2]),
                      the ControlNet itself run by the engine on the CFG pair.
    -> {case: {"t981": {cond, uncond}, ..., "max_abs": worst over timesteps and passes, "meets_atol_1e-2": bool}}"""
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter import attention_processor as AP
    from tests.harness_names import hidden_size_of
    full = torch.load(os.path.join(GOLDEN, "unet_forward_full.pt"), weights_only=False)["latent_64x64"]
    gold = torch.load(os.path.join(GOLDEN, "unet_forward_timesteps.pt"), weights_only=False)
    boc = E.SD15_CONFIG["block_out_channels"]
    res = {}
    mask = torch.tensor([1.0, 0.0], device=device)
    if "refs" in which:
        d = inputs if inputs is not None else unet_forward_inputs(64, 64, full)
        e = E.UNet2DConditionModel(d["sd"], {}, str(device), dtype)
        e.set_attn_processor({n: (AP.RefSAttnProcessor2_0(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                                  else AP.CAttnProcessor2_0(n, hidden_size_of(n, boc), 768)) for n in e.attn_processors.keys()})
        for n in d["names"]:
            p = e.attn_processors[n]
            p.to_k_ref.weight.copy_(d["rw"][n]["k"]); p.to_v_ref.weight.copy_(d["rw"][n]["v"])
        sa = {n: t.to(device) for n, t in d["sa"].items()}
        x2 = torch.cat([d["x"], d["x"]]).to(device)
        out = {}
        for t in (981, 481, 1):
            ref = full if t == 481 else gold["refs"][f"t{t}"]
            both = e(x2, t, d["ehs"].to(device), cross_attention_kwargs={"sa_hidden_states": sa, "sa_batch_mask": mask})[0].float().cpu()
            out[f"t{t}"] = dict(cond=_err_entry(both[0:1], ref["out_cond"]), uncond=_err_entry(both[1:2], ref["out_uncond"]),
                                finite=bool(torch.isfinite(both).all()))
        res["refs"] = out
        del e, sa
        torch.cuda.empty_cache()
    if "ipa_controlnet" in which:
        d = ipa_inputs if ipa_inputs is not None else ipa_controlnet_forward_inputs(gold["ipa_controlnet"])
        e = E.UNet2DConditionModel(d["sd"], {}, str(device), dtype)
        procs = {n: (AP.LoraRefSAttnProcessor2_0(n, hidden_size_of(n, boc), scale=d["ref_scale"], rank=d["rank"], lora_scale=d["lora_scale"])
                     if n.endswith("attn1.processor") else
                     AP.LoRAIPAttnProcessor2_0(hidden_size_of(n, boc), 768, rank=d["rank"], lora_scale=d["lora_scale"], scale=d["ip_scale"], num_tokens=4))
                 for n in e.attn_processors.keys()}
        fill_ipa_processors(procs, d)
        e.set_attn_processor(procs)
        ctrl = E.ControlNetModel(d["ctrl_sd"], {}, str(device), dtype)
        sa = {n: t.to(device) for n, t in d["sa"].items()}
        x2 = torch.cat([d["x"], d["x"]]).to(device)
        ehs = torch.cat([d["ehs_c"], d["ehs_u"]]).to(device)                       # rows: [cond; uncond], like PipelineBase.denoise
        ehs_ctrl = torch.cat([d["ehs"], d["ehs_u_text"]]).to(device)                # text-only states for the ControlNet
        out = {}
        for t in (981, 481, 1):
            ref = gold["ipa_controlnet"][f"t{t}"]
            down, mid = ctrl(x2, t, ehs_ctrl, d["pose"].to(device), 1.0)
            both = e(x2, t, ehs, cross_attention_kwargs={"sa_hidden_states": sa, "sa_batch_mask": mask},
                     down_block_additional_residuals=down, mid_block_additional_residual=mid)[0].float().cpu()
            out[f"t{t}"] = dict(cond=_err_entry(both[0:1], ref["out_cond"]), uncond=_err_entry(both[1:2], ref["out_uncond"]),
                                finite=bool(torch.isfinite(both).all()))
        res["ipa_controlnet"] = out
        del e, ctrl, sa
        torch.cuda.empty_cache()
    for case, out in res.items():
        worst = max(max(v["cond"]["max_abs"], v["uncond"]["max_abs"]) for v in out.values())
        out["max_abs"] = worst
        out["meets_atol_1e-2"] = bool(worst <= 1e-2)
    return res


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
