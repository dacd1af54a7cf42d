"""Build matching (CPU oracle, HIP engine) model pairs from one seeded state dict."""
import torch

from imagdressing_amd import unet as E
from oracle import processors as OP
from oracle import sd15

SMALL = dict(block_out_channels=(80, 160, 320, 320), attention_head_dim=2, norm_num_groups=8, cross_attention_dim=64)


def oracle_cfg(cfg):
    c = dict(cfg)
    if "attention_head_dim" in c:
        c["heads"] = c.pop("attention_head_dim")
    return c


def hidden_size_of(name, boc):
    if name.startswith("mid_block"):
        return boc[-1]
    if name.startswith("up_blocks"):
        return list(reversed(boc))[int(name[len("up_blocks.")])]
    return boc[int(name[len("down_blocks.")])]


def ref_weights(names, boc, seed):
    """seeded to_k_ref / to_v_ref (and friends) per attn1 processor name"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n in names:
        c = hidden_size_of(n, boc)
        out[n] = dict(k=torch.randn(c, c, generator=g) * c ** -0.5, v=torch.randn(c, c, generator=g) * c ** -0.5)
    return out


def build_pair(cfg=None, seed=0, device="cuda", kind="refs", ref_seed=7, with_controlnet=False, cross_dim=None,
               dtype=torch.bfloat16, rank=16):
    """-> dict(oracle_unet, engine_unet, oracle_ref_unet, engine_ref_unet[, controlnets])
    with processors installed on both sides carrying identical weights."""
    from imagdressing_amd.adapter import attention_processor as AP
    cfg = dict(cfg or {})
    full = dict(E.SD15_CONFIG, **cfg)
    boc = full["block_out_channels"]
    sd_u = E.random_state_dict(E.unet_param_shapes(full), seed)
    sd_r = E.random_state_dict(E.unet_param_shapes(full), seed + 1)
    o_u = sd15.UNet2DConditionModel(oracle_cfg(cfg)); o_u.load_state_dict(sd_u, strict=True)
    o_r = sd15.UNet2DConditionModel(oracle_cfg(cfg)); o_r.load_state_dict(sd_r, strict=True)
    e_u = E.UNet2DConditionModel(sd_u, cfg, device, dtype)
    e_r = E.UNet2DConditionModel(sd_r, cfg, device, dtype)
    names = list(e_u.attn_processors.keys())
    assert names == list(o_u.attn_processors.keys()), "processor name order differs between oracle and engine"
    rw = ref_weights([n for n in names if n.endswith("attn1.processor")], boc, ref_seed)
    o_procs, e_procs = {}, {}
    g = torch.Generator().manual_seed(ref_seed + 100)
    cd = full["cross_attention_dim"]

    def fill_lora(op, ep, c, kd, rank):
        with torch.no_grad():
            for nm, cin in (("q", c), ("k", kd), ("v", kd), ("out", c)):
                down = torch.randn(rank, cin, generator=g) * cin ** -0.5
                up = torch.randn(c, rank, generator=g) * rank ** -0.5
                for p in (op, ep):
                    layer = getattr(p, f"to_{nm}_lora")
                    layer.down.weight.copy_(down); layer.up.weight.copy_(up)
    for n in names:
        c = hidden_size_of(n, boc)
        if n.endswith("attn1.processor"):
            if kind == "ipa":
                op, ep = OP.LoraRefSAttn(n, c, rank=rank, lora_scale=0.2), AP.LoraRefSAttnProcessor2_0(n, c, rank=rank, lora_scale=0.2)
                fill_lora(op, ep, c, c, rank)
            else:
                op, ep = OP.RefSAttn(n, c), AP.RefSAttnProcessor2_0(n, c)
            with torch.no_grad():
                for p in (op, ep):
                    p.to_k_ref.weight.copy_(rw[n]["k"]); p.to_v_ref.weight.copy_(rw[n]["v"])
        elif kind == "ipa":
            op = OP.LoRAIPAttn(c, cd, rank=rank, lora_scale=0.2, scale=0.9, num_tokens=4)
            ep = AP.LoRAIPAttnProcessor2_0(c, cd, rank=rank, lora_scale=0.2, scale=0.9, num_tokens=4)
            fill_lora(op, ep, c, cd, rank)
            kip, vip = torch.randn(c, cd, generator=g) * cd ** -0.5, torch.randn(c, cd, generator=g) * cd ** -0.5
            with torch.no_grad():
                for p in (op, ep):
                    p.to_k_ip.weight.copy_(kip); p.to_v_ip.weight.copy_(vip)
        else:
            op, ep = OP.CAttn(n, c, cd), AP.CAttnProcessor2_0(n, c, cd)
        o_procs[n], e_procs[n] = op, ep
    o_u.set_attn_processor(o_procs); e_u.set_attn_processor(e_procs)
    o_r.set_attn_processor({n: OP.CacheAttn() for n in names})
    e_r.set_attn_processor({n: AP.CacheAttnProcessor2_0() for n in names})
    out = dict(o_unet=o_u, e_unet=e_u, o_ref=o_r, e_ref=e_r, cfg=full, names=names)
    if with_controlnet:
        sd_c = E.random_state_dict(E.controlnet_param_shapes(full), seed + 2, zero_convs=True)
        o_c = sd15.ControlNetModel(oracle_cfg(cfg)); o_c.load_state_dict(sd_c, strict=True)
        out.update(o_ctrl=o_c, e_ctrl=E.ControlNetModel(sd_c, cfg, device, dtype))
    return out


def err_stats(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = (got - ref).abs()
    return dict(max_abs=e.max().item(), mean_abs=e.mean().item(), ref_std=ref.std().item(),
                rel_rms=(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
