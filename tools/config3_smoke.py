"""BASELINE configs[2] at full width on one GPU: IP-Adapter FaceID-Plus (4 face tokens, T = 81) + LoRA-augmented hybrid processors
(rank 128) + pose ControlNet, 512x512, batch 8, bf16, random-init weights, a few DDIM steps.  Prints the step time."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import unet as E
from imagdressing_amd.adapter import attention_processor as AP
from imagdressing_amd.adapter.resampler import Resampler
from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_ipa_controlnet import IMAGDressing_v1 as IPAPipe
from imagdressing_amd.scheduler import DDIMScheduler

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev, dt = torch.device("cuda", 0), torch.bfloat16
unet = E.UNet2DConditionModel.random_init(seed=0, device=dev, dtype=dt)
ref_unet = E.UNet2DConditionModel.random_init(seed=1, device=dev, dtype=dt)
ctrl = E.ControlNetModel.random_init(seed=9, device=dev, dtype=dt)
boc = unet.cfg["block_out_channels"]
g = torch.Generator().manual_seed(2)
procs = {}
for name in unet.attn_processors.keys():           # inference_IMAGdressing_ipa_controlnetpose.py:74-96
    hs = boc[-1] if name.startswith("mid_block") else (list(reversed(boc))[int(name[len("up_blocks.")])] if name.startswith("up_blocks")
                                                        else boc[int(name[len("down_blocks.")])])
    if name.endswith("attn1.processor"):
        p = AP.LoraRefSAttnProcessor2_0(name, hs, rank=128)
    else:
        p = AP.LoRAIPAttnProcessor2_0(hs, 768, rank=128, num_tokens=4)
    with torch.no_grad():
        for n_, q in p.named_parameters():
            q.copy_(torch.randn(q.shape, generator=g) * (q.shape[-1] ** -0.5))
    procs[name] = p
unet.set_attn_processor(procs)
ref_unet.set_attn_processor({n: AP.CacheAttnProcessor2_0() for n in ref_unet.attn_processors.keys()})
torch.manual_seed(3)
proj = Resampler(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4)
sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                    set_alpha_to_one=False, steps_offset=1)
pipe = IPAPipe(vae=None, reference_unet=ref_unet, unet=unet, tokenizer=None, text_encoder=None, controlnet=ctrl, image_encoder=None,
               ImgProj=proj, ip_ckpt=None, scheduler=sch, safety_checker=None, feature_extractor=None)
B = a.batch
kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=512, height=512, num_inference_steps=a.steps,
          guidance_scale=7.0, num_images_per_prompt=B, image_scale=0.9, ipa_scale=0.9, s_lora_scale=0.2, c_lora_scale=0.2,
          pose_image=torch.rand(1, 3, 512, 512, generator=g).to(dev), faceid_embeds=torch.randn(1, 512, generator=g).to(dev),
          face_clip_hidden_states=(torch.randn(1, 257, 1280, generator=g) * 0.5).to(device=dev, dtype=dt),
          face_uncond_clip_hidden_states=(torch.randn(1, 257, 1280, generator=g) * 0.5).to(device=dev, dtype=dt),
          prompt_embeds=(torch.randn(1, 77, 768, generator=g) * 0.5).to(dev), negative_prompt_embeds=(torch.randn(1, 77, 768, generator=g) * 0.5).to(dev),
          ref_clip_hidden_states=(torch.randn(1, 257, 1280, generator=g) * 0.5).to(device=dev, dtype=dt),
          ref_image_latents=torch.randn(1, 4, 64, 64, generator=g).to(dev), latents=torch.randn(B, 4, 64, 64, generator=g).to(dev),
          output_type="latent")
out = pipe(**kw).images
torch.cuda.synchronize()
t0 = time.perf_counter()
out = pipe(**kw).images
torch.cuda.synchronize()
dtm = time.perf_counter() - t0
print(json.dumps(dict(config="configs[2]: IPA FaceID-Plus + LoRA(128) + pose ControlNet, 512x512, bf16", batch=B, steps=a.steps,
                      ms_per_step=round(dtm / a.steps * 1e3, 2), images_per_s_at_50_steps=round(B / (dtm / a.steps * 50), 3),
                      finite=bool(torch.isfinite(out).all()), shape=list(out.shape))))
