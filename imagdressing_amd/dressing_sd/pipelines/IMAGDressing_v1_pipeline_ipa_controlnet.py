"""Pipeline with IP-Adapter-FaceID-Plus face tokens + pose ControlNet
(mirrors /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline_ipa_controlnet.py:22-101, 366-742)."""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ...adapter.resampler import ProjPlusModel
from ._base import (IPAttnProcessor2_0, LoRAIPAttnProcessor2_0, LoraRefSAttnProcessor2_0, PipelineBase, controlnet_keep, first,
                    set_scale_by_type, to_image_tensor)


class IMAGDressing_v1(PipelineBase):
    _optional_components: list = []

    def __init__(self, vae, reference_unet, unet, tokenizer, text_encoder, controlnet, image_encoder, ImgProj, ip_ckpt, scheduler,
                 safety_checker=None, feature_extractor=None):
        self._init_common(vae=vae, reference_unet=reference_unet, unet=unet, tokenizer=tokenizer, text_encoder=text_encoder,
                          image_encoder=image_encoder, ImgProj=ImgProj, scheduler=scheduler, safety_checker=safety_checker,
                          feature_extractor=feature_extractor, controlnet=controlnet)
        self.ip_ckpt = ip_ckpt
        self.num_tokens = 4
        self.image_proj_model = self.init_proj()
        if ip_ckpt is not None:
            self.load_ip_adapter()

    def init_proj(self):                                                     # :79-86
        clip_dim = 1280
        if self.image_encoder is not None and hasattr(self.image_encoder, "config"):
            clip_dim = self.image_encoder.config.hidden_size
        return ProjPlusModel(cross_attention_dim=self.unet.config.cross_attention_dim, id_embeddings_dim=512,
                             clip_embeddings_dim=clip_dim, num_tokens=self.num_tokens)

    def load_ip_adapter(self):                                               # :88-101
        if isinstance(self.ip_ckpt, dict):
            state_dict = self.ip_ckpt
        elif os.path.splitext(self.ip_ckpt)[-1] == ".safetensors":
            from safetensors import safe_open
            state_dict = {"image_proj": {}, "ip_adapter": {}}
            with safe_open(self.ip_ckpt, framework="pt", device="cpu") as f:
                for key in f.keys():
                    if key.startswith("image_proj."):
                        state_dict["image_proj"][key.replace("image_proj.", "")] = f.get_tensor(key)
                    elif key.startswith("ip_adapter."):
                        state_dict["ip_adapter"][key.replace("ip_adapter.", "")] = f.get_tensor(key)
        else:
            state_dict = torch.load(self.ip_ckpt, map_location="cpu")
        self.image_proj_model.load_state_dict(state_dict["image_proj"])
        ip_layers = torch.nn.ModuleList([p for p in self.unet.attn_processors.values()])
        ip_layers.load_state_dict(state_dict["ip_adapter"], strict=False)

    def get_image_embeds(self, clip_image=None, faceid_embeds=None, clip_hidden_states=None, uncond_clip_hidden_states=None):
        """(face tokens, uncond face tokens) [1, 4, 768] each (:366-377).  ``shortcut`` stays False (:375)."""
        dev = self.device
        if clip_hidden_states is None:
            clip_hidden_states = self._clip_hidden(clip_image, dev)
            uncond_clip_hidden_states = self._clip_hidden(torch.zeros_like(clip_image), dev)
        faceid_embeds = faceid_embeds.to(dev)
        pos = self.image_proj_model(faceid_embeds, clip_hidden_states.to(dev))
        neg = self.image_proj_model(torch.zeros_like(faceid_embeds), uncond_clip_hidden_states.to(dev))
        return pos, neg

    def set_scale(self, scale, lora_scale):                                  # :379-383
        set_scale_by_type(self.unet, LoraRefSAttnProcessor2_0, scale=scale, lora_scale=lora_scale)

    def set_ipa_scale(self, ipa_scale, lora_scale):                          # :386-393
        set_scale_by_type(self.unet, LoRAIPAttnProcessor2_0, scale=ipa_scale, lora_scale=lora_scale)
        set_scale_by_type(self.unet, IPAttnProcessor2_0, scale=ipa_scale, lora_scale=lora_scale)

    @torch.no_grad()
    def __call__(self, prompt, null_prompt, negative_prompt, ref_image, width, height, num_inference_steps, guidance_scale,
                 pose_image=None, ref_clip_image=None, face_clip_image=None, faceid_embeds=None, num_images_per_prompt=1,
                 image_scale=1.0, ipa_scale=0.0, s_lora_scale=0.0, c_lora_scale=0.0, num_samples=1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, clip_skip: Optional[int] = None, callback: Optional[Callable] = None,
                 callback_steps: Optional[int] = 1, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 ref_clip_hidden_states: Optional[torch.Tensor] = None, ref_image_latents: Optional[torch.Tensor] = None,
                 face_clip_hidden_states: Optional[torch.Tensor] = None, face_uncond_clip_hidden_states: Optional[torch.Tensor] = None,
                 latents: Optional[torch.Tensor] = None, shard_over_ranks: bool = False, trace: Optional[list] = None, **kwargs):
        if guess_mode or guidance_scale <= 1.0:
            # neither runs in the reference: with guess_mode its ControlNet sees the cond half only and the loop then indexes
            # down_block[1] of a batch-1 tensor (..._ipa_controlnet.py:634-639, :662-665; the zero-padding lines are commented out);
            # without CFG latent_model_input[1] does not exist (:672, :690)
            raise NotImplementedError("guess_mode / guidance_scale <= 1: the reference's loop indexes the CFG pair of the ControlNet "
                                      "residuals and of the latents unconditionally (..._ipa_controlnet.py:662-690)")
        has_face = face_clip_image is not None or face_clip_hidden_states is not None
        if not has_face:                                                      # :432-437
            self.set_scale(image_scale, lora_scale=0.0)
            self.set_ipa_scale(ipa_scale=0.0, lora_scale=0.0)
        else:
            self.set_scale(image_scale, lora_scale=s_lora_scale)
            self.set_ipa_scale(ipa_scale, lora_scale=c_lora_scale)
        device = self.device
        self._cross_attention_kwargs = cross_attention_kwargs
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(
            prompt, device, num_images_per_prompt, True, negative_prompt, prompt_embeds=prompt_embeds,
            negative_prompt_embeds=negative_prompt_embeds, clip_skip=clip_skip)
        control = None
        if pose_image is not None:                                            # ControlNet sees the 77 text tokens only (:550)
            control = dict(image=to_image_tensor(pose_image, device, normalize=False, size=(height, width), multiple=self.vae_scale_factor),
                           prompt_embeds=prompt_embeds,
                           negative_prompt_embeds=negative_prompt_embeds, scale=float(first(controlnet_conditioning_scale)),
                           keep=controlnet_keep(num_inference_steps, float(first(control_guidance_start)),
                                                float(first(control_guidance_end))))
            height, width = control["image"].shape[-2:]
        if has_face:                                                          # :513-521, :555-557
            pos, neg = self.get_image_embeds(face_clip_image, faceid_embeds, face_clip_hidden_states, face_uncond_clip_hidden_states)
            prompt_embeds = torch.cat([prompt_embeds.to(device), pos.to(prompt_embeds.dtype)], dim=1)
            negative_prompt_embeds = torch.cat([negative_prompt_embeds.to(device), neg.to(negative_prompt_embeds.dtype)], dim=1)
        if ref_clip_image is None and ref_clip_hidden_states is None:
            cloth_tokens, _ = self.encode_prompt(null_prompt, device, 1, False)
        else:
            cloth_tokens = self._cloth_tokens(ref_clip_image, ref_clip_hidden_states, device)
        lat = self._shard(self.prepare_latents(num_images_per_prompt, 4, width, height, torch.float32, device, generator, latents),
                          shard_over_ranks)
        ref_lat = self._ref_latents(ref_image, ref_image_latents)
        sa = self._sa_states(ref_lat, cloth_tokens, shard_over_ranks)
        out = self.denoise(latents=lat, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                           sa_hidden_states=sa, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                           control=control, callback=callback, callback_steps=callback_steps or 1, trace=trace,
                           eta=eta, generator=generator, variance_noise=kwargs.get("variance_noise"))
        return self._decode(out, output_type, generator)
