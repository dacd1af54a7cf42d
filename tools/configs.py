"""Builders for the BASELINE.json configurations that are NOT the bench line (parity / tuning / timing cases):

  build(3, ...)  configs[2]: IP-Adapter FaceID-Plus (4 face tokens, T = 81) + rank-128 LoRA hybrid processors + pose ControlNet,
                 512x512, batch 8 on one GPU  (inference_IMAGdressing_ipa_controlnetpose.py)
  build(5, ...)  configs[4]: ControlNet-inpainting at 768x576 (latent 96x72: N = 6912 / 1728 / 432 / 108), 4 images per GPU
                 (inference_IMAGdressing_controlnetinpainting.py)
  build(1, ...)  configs[1]: the bench.py workload

-> (pipeline, kwargs) with full-width random-init SD1.5 weights and seeded synthetic inputs resident in HBM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse

import torch

import bench
from imagdressing_amd import unet as E
from imagdressing_amd.adapter import attention_processor as AP


def build(config: int, dev, dt, batch=None, steps=6, width=512, height=512):
    g = torch.Generator().manual_seed(2)
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale        # noqa: E731
    if config == 1:          # (width / height: also the reference scripts' default geometry 512 x 640 and batch 1, for the tuning table)
        batch = batch or 4
        pipe = bench.build_pipeline(dev, dt, 0)
        inp = bench.synthetic_inputs(width, height, batch, dev, dt, 0, 1)
        kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=width, height=height,
                  num_inference_steps=steps, guidance_scale=7.5, num_images_per_prompt=batch, output_type="latent", **inp)
        return pipe, kw
    base = bench.build_pipeline(dev, dt, 0)
    ctrl = E.ControlNetModel.random_init(seed=9, device=dev, dtype=dt)
    if config == 3:
        from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_ipa_controlnet import IMAGDressing_v1 as IPAPipe
        batch = batch or 8
        unet = base.unet
        boc = unet.cfg["block_out_channels"]
        procs = {}
        for name in unet.attn_processors.keys():           # inference_IMAGdressing_ipa_controlnetpose.py:74-96
            hs = boc[-1] if name.startswith("mid_block") else (list(reversed(boc))[int(name[len("up_blocks.")])] if name.startswith("up_blocks")
                                                                else boc[int(name[len("down_blocks.")])])
            p = AP.LoraRefSAttnProcessor2_0(name, hs, rank=128) if name.endswith("attn1.processor") else \
                AP.LoRAIPAttnProcessor2_0(hs, 768, rank=128, num_tokens=4)
            with torch.no_grad():
                for _, q in p.named_parameters():
                    q.copy_(torch.randn(q.shape, generator=g) * (q.shape[-1] ** -0.5))
            procs[name] = p
        unet.set_attn_processor(procs)
        pipe = IPAPipe(vae=None, reference_unet=base.reference_unet, unet=unet, tokenizer=None, text_encoder=None, controlnet=ctrl,
                       image_encoder=None, ImgProj=base.ImgProj, ip_ckpt=None, scheduler=base.scheduler, safety_checker=None,
                       feature_extractor=None)
        kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=512, height=512, num_inference_steps=steps,
                  guidance_scale=7.0, num_images_per_prompt=batch, image_scale=0.9, ipa_scale=0.9, s_lora_scale=0.2, c_lora_scale=0.2,
                  pose_image=torch.rand(1, 3, 512, 512, generator=g).to(dev), faceid_embeds=rn(1, 512).to(dev),
                  face_clip_hidden_states=rn(1, 257, 1280, scale=0.5).to(device=dev, dtype=dt),
                  face_uncond_clip_hidden_states=rn(1, 257, 1280, scale=0.5).to(device=dev, dtype=dt),
                  prompt_embeds=rn(1, 77, 768, scale=0.5).to(dev), negative_prompt_embeds=rn(1, 77, 768, scale=0.5).to(dev),
                  ref_clip_hidden_states=rn(1, 257, 1280, scale=0.5).to(device=dev, dtype=dt),
                  ref_image_latents=rn(1, 4, 64, 64).to(dev), latents=rn(batch, 4, 64, 64).to(dev), output_type="latent")
        return pipe, kw
    if config == 5:
        from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1 as Inpaint
        batch = batch or 4
        pipe = Inpaint(vae=None, reference_unet=base.reference_unet, unet=base.unet, tokenizer=None, text_encoder=None, controlnet=ctrl,
                       image_encoder=None, ImgProj=base.ImgProj, scheduler=base.scheduler, safety_checker=None, feature_extractor=None)
        W, H = 576, 768
        h, w = H // 8, W // 8
        mask = torch.zeros(1, 1, h, w)
        mask[:, :, int(h * 0.184): int(h * 0.816), int(w * 0.184): int(w * 0.816)] = 1.0     # centred rectangle, 40 % of the area
        kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=W, height=H, num_inference_steps=steps,
                  guidance_scale=5.0, num_images_per_prompt=batch, prompt_embeds=rn(1, 77, 768, scale=0.5).to(dev),
                  negative_prompt_embeds=rn(1, 77, 768, scale=0.5).to(dev),
                  ref_clip_hidden_states=rn(1, 257, 1280, scale=0.5).to(device=dev, dtype=dt),
                  ref_image_latents=rn(1, 4, h, w).to(dev), control_image=torch.rand(1, 3, H, W, generator=g).to(dev),
                  image_latents=rn(1, 4, h, w).to(dev), mask_latents=mask.to(dev), noise=rn(batch, 4, h, w).to(dev), output_type="latent")
        return pipe, kw
    raise ValueError(f"config {config}")


if __name__ == "__main__":
    import json
    import time
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--fp8-attention", action="store_true", help="level-0 attention on the MX-FP8 MFMA (ops.ATTN_FP8)")
    a = ap.parse_args()
    if a.fp8_attention:
        from imagdressing_amd import ops
        ops.ATTN_FP8 = True
    dev = torch.device("cuda", 0)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    pipe, kw = build(a.config, dev, dt, a.batch or None, a.steps)
    out = pipe(**kw).images
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe(**kw).images
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    B = kw["num_images_per_prompt"]
    print(json.dumps(dict(config=f"BASELINE configs[{a.config - 1}]", dtype=a.dtype, fp8_attention=a.fp8_attention, batch=B, steps=a.steps,
                          ms_per_step=round(dtm / a.steps * 1e3, 2), images_per_s_at_50_steps=round(B / (dtm / a.steps * 50), 3),
                          finite=bool(torch.isfinite(out).all()), shape=list(out.shape))))
