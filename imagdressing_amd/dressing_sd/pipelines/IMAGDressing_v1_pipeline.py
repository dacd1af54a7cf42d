"""Base pipeline: text + garment conditioning, no ControlNet
(mirrors /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:18-40, 342-547)."""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import PipelineBase, RefSAttnProcessor2_0, StableDiffusionPipelineOutput, set_scale_by_type


class IMAGDressing_v1(PipelineBase):
    _optional_components: list = []

    def __init__(self, vae, reference_unet, unet, tokenizer, text_encoder, image_encoder, ImgProj, scheduler,
                 safety_checker=None, feature_extractor=None):
        self._init_common(vae=vae, reference_unet=reference_unet, unet=unet, tokenizer=tokenizer, text_encoder=text_encoder,
                          image_encoder=image_encoder, ImgProj=ImgProj, scheduler=scheduler, safety_checker=safety_checker,
                          feature_extractor=feature_extractor)

    def set_scale(self, scale):                                            # :342-345
        set_scale_by_type(self.unet, RefSAttnProcessor2_0, scale=scale)

    @torch.no_grad()
    def __call__(self, prompt, null_prompt, negative_prompt, ref_image, width, height, num_inference_steps, guidance_scale,
                 ref_clip_image=None, num_images_per_prompt=1, image_scale=1.0, num_samples=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, clip_skip: Optional[int] = None, callback: Optional[Callable] = None,
                 callback_steps: Optional[int] = 1, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 # --- extensions: bypass the out-of-scope encoders / inject latents / shard over ranks ---
                 ref_clip_hidden_states: Optional[torch.Tensor] = None, ref_image_latents: Optional[torch.Tensor] = None,
                 latents: Optional[torch.Tensor] = None, shard_over_ranks: bool = False, trace: Optional[list] = None, **kwargs):
        if guidance_scale <= 1.0:
            # the reference cannot run this either: its loop indexes the CFG pair unconditionally (cache["hidden_states"][1] :476-479,
            # latent_model_input[1] :511) and null_prompt_embeds is only bound under do_classifier_free_guidance (:431-435)
            raise NotImplementedError("guidance_scale <= 1: the reference's loop indexes the CFG pair unconditionally "
                                      "(IMAGDressing_v1_pipeline.py:476-479, :511); sample with guidance_scale > 1")
        self.set_scale(image_scale)                                        # :374
        device = self.device
        self._cross_attention_kwargs = cross_attention_kwargs
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, num_images_per_prompt, True, negative_prompt, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, clip_skip=clip_skip)                     # :395-405
        if ref_clip_image is None and ref_clip_hidden_states is None:
            # the reference falls back to text "null prompt" tokens as garment-UNet context (:416-427)
            cloth_tokens, _ = self.encode_prompt(null_prompt, device, 1, False)
        else:
            cloth_tokens = self._cloth_tokens(ref_clip_image, ref_clip_hidden_states, device)      # :409-415
        lat = self.prepare_latents(num_images_per_prompt, 4, width, height, torch.float32, device, generator, latents)
        lat = self._shard(lat, shard_over_ranks)
        ref_lat = self._ref_latents(ref_image, ref_image_latents)                                   # :454-458
        sa = self._sa_states(ref_lat, cloth_tokens, shard_over_ranks)                               # :465-480
        out = self.denoise(latents=lat, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                           sa_hidden_states=sa, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                           callback=callback, callback_steps=callback_steps or 1, trace=trace,
                           eta=eta, generator=generator, variance_noise=kwargs.get("variance_noise"))           # eta: :451, :530
        return self._decode(out, output_type, generator)                                            # :544-547
