"""Pipeline with an optional pose ControlNet
(mirrors /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline_controlnet.py:20-45, 352-677)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import (PipelineBase, RefSAttnProcessor2_0, controlnet_keep, first, set_scale_by_type, to_image_tensor)


class IMAGDressing_v1(PipelineBase):
    _optional_components: list = []

    def __init__(self, vae, reference_unet, unet, tokenizer, text_encoder, controlnet, image_encoder, ImgProj, scheduler,
                 safety_checker=None, feature_extractor=None):
        self._init_common(vae=vae, reference_unet=reference_unet, unet=unet, tokenizer=tokenizer, text_encoder=text_encoder,
                          image_encoder=image_encoder, ImgProj=ImgProj, scheduler=scheduler, safety_checker=safety_checker,
                          feature_extractor=feature_extractor, controlnet=controlnet)

    def set_scale(self, scale):                                              # :352-355
        set_scale_by_type(self.unet, RefSAttnProcessor2_0, scale=scale)

    def _control(self, pose_image, prompt_embeds, negative_prompt_embeds, num_inference_steps, scale, start, end, device, size=None):
        """ControlNet inputs: the pose image (shared by the CFG halves, :497-498) and the TEXT-ONLY embeddings
        (``prompt_embeds_control``, ..._ipa_controlnet.py:550)."""
        if pose_image is None:
            return None
        return dict(image=to_image_tensor(pose_image, device, normalize=False, size=size, multiple=self.vae_scale_factor),
                    prompt_embeds=prompt_embeds,
                    negative_prompt_embeds=negative_prompt_embeds, scale=float(first(scale)),
                    keep=controlnet_keep(num_inference_steps, float(first(start)), float(first(end))))

    @torch.no_grad()
    def __call__(self, prompt, null_prompt, negative_prompt, ref_image, width, height, num_inference_steps, guidance_scale,
                 pose_image=None, ref_clip_image=None, num_images_per_prompt=1, image_scale=1.0, num_samples=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, clip_skip: Optional[int] = None, callback: Optional[Callable] = None,
                 callback_steps: Optional[int] = 1, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 ref_clip_hidden_states: Optional[torch.Tensor] = None, ref_image_latents: Optional[torch.Tensor] = None,
                 latents: Optional[torch.Tensor] = None, shard_over_ranks: bool = False, trace: Optional[list] = None, **kwargs):
        if guess_mode or guidance_scale <= 1.0:
            # neither runs in the reference: with guess_mode its ControlNet sees the cond half only and the loop then indexes
            # down_block[1] of a batch-1 tensor (..._ipa_controlnet.py:634-639, :662-665; the zero-padding lines are commented out);
            # without CFG latent_model_input[1] does not exist (:672, :690)
            raise NotImplementedError("guess_mode / guidance_scale <= 1: the reference's loop indexes the CFG pair of the ControlNet "
                                      "residuals and of the latents unconditionally (..._ipa_controlnet.py:662-690)")
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
        control = self._control(pose_image, prompt_embeds, negative_prompt_embeds, num_inference_steps,
                                controlnet_conditioning_scale, control_guidance_start, control_guidance_end, device,
                                size=(height, width))
        if control is not None:
            height, width = control["image"].shape[-2:]                     # :501
        lat = self._shard(self.prepare_latents(num_images_per_prompt, 4, width, height, torch.float32, device, generator, latents),
                          shard_over_ranks)
        ref_lat = self._ref_latents(ref_image, ref_image_latents)
        sa = self._sa_states(ref_lat, cloth_tokens, shard_over_ranks)
        out = self.denoise(latents=lat, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                           sa_hidden_states=sa, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                           control=control, callback=callback, callback_steps=callback_steps or 1, trace=trace,
                           eta=eta, generator=generator, variance_noise=kwargs.get("variance_noise"))
        return self._decode(out, output_type, generator)
