"""Checkpoint ingestion for the IMAGDressing-v1 hot path.

The reference's `prepare()` (/root/reference/inference_IMAGdressing.py:40-135) does four things with files:
  * SD1.5 UNet weights  (`UNet2DConditionModel.from_pretrained(..., subfolder="unet")`, :48-50, :90-92) -- a diffusers-layout
    state dict (`diffusion_pytorch_model.safetensors` / `.bin`);
  * the IMAGDressing checkpoint (`torch.load(args.model_ckpt)["module"]`, :97): a DeepSpeed-wrapped state dict whose keys are
    prefixed `ref_unet.` (garment UNet), `unet.` (unused at inference: the base UNet stays frozen), `proj.` (Resampler) and
    `adapter_modules.{i}.` (the attention processors in `unet.attn_processors` order) -- split at :99-112, loaded at :114-116;
  * the VAE (`AutoencoderKL.from_pretrained`, :42);
  * (IPA variant) the FaceID-Plus file with `image_proj.` / `ip_adapter.` groups (`…ipa_controlnet.py:88-101`), handled by
    the pipeline class itself.
This module restates that plumbing for the MI355X engines: weights are repacked ONCE at load time into the layouts the
kernels consume (NHWC-ordered 3x3 filters, interleaved GEGLU rows, concatenated time-embedding projections, 16-bit), so
nothing is cast or permuted per run.  Host-side only; tensors are read on the CPU and land in HBM inside the engines."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

PREFIXES = ("ref_unet", "unet", "proj", "adapter_modules")


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """`.safetensors` or a torch pickle (`.pt` / `.bin` / `.ckpt`), unwrapping DeepSpeed's {"module": ...} level (:97)."""
    if os.path.splitext(path)[-1] == ".safetensors":
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "module" in sd and isinstance(sd["module"], dict):
        sd = sd["module"]
    elif isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return sd


def split_imagdressing_state_dict(model_sd: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """The key routing of inference_IMAGdressing.py:99-112: first matching prefix wins, in the reference's order
    (`ref_unet`, `unet`, `proj`, `adapter_modules`); anything else is returned under "other" (the script prints it)."""
    out: Dict[str, Dict[str, torch.Tensor]] = {p: {} for p in PREFIXES}
    out["other"] = {}
    for k, v in model_sd.items():
        for p in PREFIXES:
            if k.startswith(p):
                out[p][k.replace(p + ".", "")] = v        # the reference's own renaming (str.replace, :101-108)
                break
        else:
            out["other"][k] = v
    return out


def hidden_size_of(name: str, block_out_channels) -> int:
    """inference_IMAGdressing.py:70-77"""
    if name.startswith("mid_block"):
        return block_out_channels[-1]
    if name.startswith("up_blocks"):
        return list(reversed(block_out_channels))[int(name[len("up_blocks.")])]
    return block_out_channels[int(name[len("down_blocks.")])]


def build_engines(unet_sd: Dict[str, torch.Tensor], imagdressing_sd: Dict[str, torch.Tensor], *, device="cuda",
                  dtype=torch.float16, vae_sd: Optional[Dict[str, torch.Tensor]] = None, config: Optional[dict] = None,
                  resampler_kwargs: Optional[dict] = None, strict: bool = True):
    """What `prepare()` builds, on the MI355X engines: -> dict(unet, ref_unet, image_proj, vae, other_keys).

    `unet_sd`: diffusers-layout SD1.5 UNet weights; `imagdressing_sd`: the (unwrapped) IMAGDressing checkpoint.
    The denoising UNet gets `RefSAttnProcessor2_0` / `CAttnProcessor2_0` (:79-83) loaded from `adapter_modules.*` through
    `ModuleList(unet.attn_processors.values()).load_state_dict` exactly as :86 / :116; the garment UNet is built from
    `ref_unet.*` with `CacheAttnProcessor2_0` (:93-94, :114); the Resampler from `proj.*` (:55-64, :115)."""
    from . import unet as E
    from .adapter import attention_processor as AP
    from .adapter.resampler import Resampler
    parts = split_imagdressing_state_dict(imagdressing_sd)
    full = dict(E.SD15_CONFIG, **(config or {}))
    boc, cd = full["block_out_channels"], full["cross_attention_dim"]
    unet = E.UNet2DConditionModel(unet_sd, config, device, dtype)
    procs = {}
    for name in unet.attn_processors.keys():
        hs = hidden_size_of(name, boc)
        procs[name] = AP.RefSAttnProcessor2_0(name, hs) if name.endswith("attn1.processor") else \
            AP.CAttnProcessor2_0(name, hidden_size=hs, cross_attention_dim=cd)
    unet.set_attn_processor(procs)
    adapter_modules = torch.nn.ModuleList(unet.attn_processors.values())
    adapter_modules.load_state_dict(parts["adapter_modules"], strict=strict)
    adapter_modules.to(device=device, dtype=dtype)
    ref_unet = E.UNet2DConditionModel(parts["ref_unet"], config, device, dtype)
    ref_unet.set_attn_processor({n: AP.CacheAttnProcessor2_0() for n in ref_unet.attn_processors.keys()})
    rk = dict(dim=cd, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=cd, ff_mult=4)
    rk.update(resampler_kwargs or {})
    image_proj = Resampler(**rk)
    image_proj.load_state_dict(parts["proj"], strict=strict)
    image_proj.to(device=device, dtype=dtype)
    vae = None
    if vae_sd is not None:
        from .vae import AutoencoderKL
        vae = AutoencoderKL(vae_sd, None, device, dtype)
    return dict(unet=unet, ref_unet=ref_unet, image_proj=image_proj, vae=vae, other_keys=sorted(parts["other"]),
                unused_unet_keys=len(parts["unet"]))
