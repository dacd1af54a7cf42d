"""ControlNet-inpainting pipeline: per-step masked latent blend
(mirrors /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline_controlnet_inpainting.py:13-40, 117-548)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import (PipelineBase, RefSAttnProcessor2_0, controlnet_keep, first, randn_tensor, set_scale_by_type,
                    to_image_tensor)


class IMAGDressing_v1(PipelineBase):
    _optional_components: list = []

    def __init__(self, vae, reference_unet, unet, tokenizer, text_encoder, controlnet, image_encoder, ImgProj, scheduler,
                 safety_checker=None, feature_extractor=None, requires_safety_checker: bool = True):
        self._init_common(vae=vae, reference_unet=reference_unet, unet=unet, tokenizer=tokenizer, text_encoder=text_encoder,
                          image_encoder=image_encoder, ImgProj=ImgProj, scheduler=scheduler, safety_checker=safety_checker,
                          feature_extractor=feature_extractor, controlnet=controlnet)

    def set_scale(self, scale):
        set_scale_by_type(self.unet, RefSAttnProcessor2_0, scale=scale)

    def _image_latents(self, image, device, generator, size=None):
        """VAE-encode the person image (inherited ``prepare_latents(..., return_image_latents=True)``, :330-346)."""
        p = next(self.vae.parameters())
        x = to_image_tensor(image, p.device, normalize=True, size=size, multiple=self.vae_scale_factor).to(p.dtype)
        return self.vae.encode(x).latent_dist.sample(generator) * self.vae.config.scaling_factor

    @torch.no_grad()
    def __call__(self, prompt, null_prompt, negative_prompt, ref_image, width, height, num_inference_steps, guidance_scale,
                 ref_clip_image=None, num_images_per_prompt=1, image_scale=1.0, num_samples=1, strength: float = 1.0,
                 image=None, mask_image=None, control_image=None, padding_mask_crop: Optional[int] = None,
                 latents: Optional[torch.Tensor] = None, timesteps: List[int] = None,
                 callback_on_step_end: Optional[Callable] = None, callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 eta: float = 0.0, generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, clip_skip: Optional[int] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 ref_clip_hidden_states: Optional[torch.Tensor] = None, ref_image_latents: Optional[torch.Tensor] = None,
                 image_latents: Optional[torch.Tensor] = None, mask_latents: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None, shard_over_ranks: bool = False, trace: Optional[list] = None, **kwargs):
        if guess_mode or guidance_scale <= 1.0 or padding_mask_crop is not None or timesteps:
            raise NotImplementedError("guess_mode, guidance_scale <= 1, padding_mask_crop and custom timesteps are not implemented "
                                      "(the reference script uses none of them)")
        if not 0.0 < float(strength) <= 1.0:
            raise ValueError(f"The value of strength should in (0.0, 1.0] but is {strength}")           # diffusers check_inputs (0.0 leaves no step)
        callback = kwargs.pop("callback", None)
        callback_steps = kwargs.pop("callback_steps", None) or 1
        self.set_scale(image_scale)
        device = self.device
        self._cross_attention_kwargs = cross_attention_kwargs
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, num_images_per_prompt, True, negative_prompt, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, clip_skip=clip_skip)
        if ref_clip_image is None and ref_clip_hidden_states is None:
            cloth_tokens, _ = self.encode_prompt(null_prompt, device, 1, False)
        else:
            cloth_tokens = self._cloth_tokens(ref_clip_image, ref_clip_hidden_states, device)
        steps_run = min(int(num_inference_steps * float(strength)), num_inference_steps)          # the gate is over the timesteps actually run (:376-381)
        control = dict(image=to_image_tensor(control_image, device, normalize=False, size=(height, width), multiple=self.vae_scale_factor),
                       prompt_embeds=prompt_embeds,
                       negative_prompt_embeds=negative_prompt_embeds, scale=float(first(controlnet_conditioning_scale)),
                       keep=controlnet_keep(max(steps_run, 1), float(first(control_guidance_start)), float(first(control_guidance_end))))
        B = num_images_per_prompt
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        # strength == 1.0: start from pure noise; the SAME noise re-noises the original latents in the blend (:496-498).
        # strength < 1.0 (:316-341 -> diffusers get_timesteps / prepare_latents): run the last int(steps * strength) timesteps,
        # starting from add_noise(image_latents, noise, first of them).  Explicit ``latents`` are taken as the noise, as diffusers does.
        if image_latents is None:                                             # (before the noise draw, like diffusers' prepare_latents)
            image_latents = self._image_latents(image, device, generator, size=(height, width))
        if noise is None:
            noise = latents if latents is not None else randn_tensor((B, 4, h, w), generator=generator, device=device, dtype=torch.float32)
        init_steps = min(int(num_inference_steps * float(strength)), num_inference_steps)
        t_start = max(num_inference_steps - init_steps, 0)
        if init_steps < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline "
                             f"steps is {init_steps} which is < 1 and not appropriate for this pipeline.")
        if latents is not None or float(strength) == 1.0:
            lat = (noise if latents is None else latents).to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma
        else:
            self.scheduler.set_timesteps(num_inference_steps, device=device)
            t0 = int(self.scheduler.timesteps[t_start * getattr(self.scheduler, "order", 1)])
            il = image_latents.to(device=device, dtype=torch.float32)
            lat = self.scheduler.add_noise(il.expand(B, -1, -1, -1) if il.shape[0] != B else il, noise.to(device=device, dtype=torch.float32), t0)
        if mask_latents is None:                                              # prepare_mask_latents: nearest resize to h x w
            m = to_image_tensor(mask_image, device, normalize=False)[:, :1]
            m = (m >= 0.5).float()
            mask_latents = torch.nn.functional.interpolate(m, size=(h, w))
        lat, noise_s = self._shard(lat, shard_over_ranks), self._shard(noise.to(device), shard_over_ranks)
        ref_lat = self._ref_latents(ref_image, ref_image_latents)
        sa = self._sa_states(ref_lat, cloth_tokens, shard_over_ranks)
        inpaint = dict(mask=mask_latents, image_latents=image_latents, noise=noise_s)
        out = self.denoise(latents=lat, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                           sa_hidden_states=sa, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                           control=control, inpaint=inpaint, callback=callback, callback_steps=callback_steps, trace=trace,
                           eta=eta, generator=generator, variance_noise=kwargs.get("variance_noise"), t_start=t_start)
        return self._decode(out, output_type, generator)
