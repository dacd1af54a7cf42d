"""The reference's sampling-loop semantics, from embeddings/latents to final latents
(TEST INFRASTRUCTURE).

Restates, for one image (the reference hard-codes batch_size = 1,
IMAGDressing_v1_pipeline.py:389):

* base loop                ``IMAGDressing_v1_pipeline.py:463-541``
* + ControlNet / IP tokens ``..._pipeline_ipa_controlnet.py:595-736``
* + inpaint blend          ``..._pipeline_controlnet_inpainting.py:387-517``

Everything before the loop that is NOT on the hot path (CLIP text/vision encoders, VAE encode,
SURVEY.md section 8f "next") enters as tensors: ``prompt_embeds`` [1,T,768],
``negative_prompt_embeds`` [1,T,768], ``cloth_embeds`` = cat([cloth_null, cloth_proj]) [2,16,768]
(:433), ``ref_latents`` [1,4,h,w] (:457-458), initial ``latents`` [1,4,h,w] (:440-448).
Batched generation (B>1) is DEFINED as B independent runs of this function.
"""
from __future__ import annotations

from typing import Optional

import torch


@torch.no_grad()
def garment_features(reference_unet, ref_latents, cloth_embeds):
    """Garment UNet once at t=0 on the CFG-doubled garment latent (:466-473); keep the cond
    half ``[1]`` of each ``CacheAttnProcessor2_0`` input (:476-479)."""
    reference_unet(ref_latents.repeat(2, 1, 1, 1), torch.zeros((), dtype=torch.long), cloth_embeds)
    return {name: proc.cache["hidden_states"][1].unsqueeze(0)
            for name, proc in reference_unet.attn_processors.items()}


@torch.no_grad()
def denoise(
    unet, reference_unet, scheduler, latents, prompt_embeds, negative_prompt_embeds, cloth_embeds,
    ref_latents, num_inference_steps: int, guidance_scale: float,
    controlnet=None, control_image=None, prompt_embeds_control=None, conditioning_scale: float = 1.0,
    inpaint: Optional[dict] = None, trace: Optional[list] = None,
    eta: float = 0.0, variance_noise: Optional[list] = None, strength: float = 1.0,
):
    """Returns the final latents [1,4,h,w].  ``inpaint`` = dict(mask [1,1,h,w], image_latents,
    noise) enables the per-step blend (..._controlnet_inpainting.py:487-500).
    ``prompt_embeds_control`` = cat([negative, prompt]) *text-only* embeds [2,77,768]
    (..._ipa_controlnet.py:550).  ``trace`` collects latents after every step.
    ``eta`` > 0 with ``variance_noise`` = one [1,4,h,w] tensor per executed step: the stochastic DDIM step the reference reaches
    through ``prepare_extra_step_kwargs`` (IMAGDressing_v1_pipeline.py:102-119, :530).  ``strength`` < 1 (inpainting only,
    ..._controlnet_inpainting.py:316-341, diffusers' ``get_timesteps`` / ``prepare_latents``): the last
    int(steps * strength) timesteps are run, starting from add_noise(image_latents, noise, first of them)."""
    timesteps = scheduler.set_timesteps(num_inference_steps)
    if strength < 1.0:
        assert inpaint is not None
        init = min(int(num_inference_steps * strength), num_inference_steps)
        timesteps = timesteps[max(num_inference_steps - init, 0):]
        latents = scheduler.add_noise(inpaint["image_latents"], inpaint["noise"], timesteps[0])
    sa = None
    for i, t in enumerate(timesteps):
        if i == 0:
            sa = garment_features(reference_unet, ref_latents, cloth_embeds)
        lmi = scheduler.scale_model_input(torch.cat([latents] * 2), t)       # :483-488
        kw_c, kw_u = {}, {}
        if controlnet is not None:
            down, mid = controlnet(lmi, t, prompt_embeds_control, control_image, conditioning_scale)
            kw_c = dict(down_block_additional_residuals=[d[1] for d in down],      # [1] -> cond
                        mid_block_additional_residual=mid[1])
            kw_u = dict(down_block_additional_residuals=[d[0] for d in down],      # [0] -> uncond
                        mid_block_additional_residual=mid[0])
        eps_c = unet(lmi[0:1], t, prompt_embeds, cross_attention_kwargs={"sa_hidden_states": sa}, **kw_c)
        eps_u = unet(lmi[1:2], t, negative_prompt_embeds, **kw_u)               # no garment branch
        eps = eps_u + guidance_scale * (eps_c - eps_u)                            # :521-527
        latents = scheduler.step(eps, t, latents, **({} if eta == 0 else dict(eta=eta, variance_noise=variance_noise[i])))   # :530-532
        if inpaint is not None:
            proper = inpaint["image_latents"]
            if i < len(timesteps) - 1:
                proper = scheduler.add_noise(proper, inpaint["noise"], timesteps[i + 1])
            m = inpaint["mask"]
            latents = (1 - m) * proper + m * latents
        if trace is not None:
            trace.append(latents.clone())
    return latents
