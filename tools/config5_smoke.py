"""BASELINE configs[4] geometry on one GPU: the ControlNet-inpainting pipeline at 768x576 (latent 96x72: N = 6912 / 1728 / 432 /
108 tokens per level), full-width SD1.5 UNet + ControlNet with random-init weights, 4 images, a few DDIM steps.  Prints the
step time and checks the result is finite and that the un-masked region follows the re-noised original latents."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagdressing_amd import unet as E
from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1 as Inpaint

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
dev, dt = torch.device("cuda", 0), torch.bfloat16
base = bench.build_pipeline(dev, dt, 0)
ctrl = E.ControlNetModel.random_init(seed=9, device=dev, dtype=dt)
pipe = Inpaint(vae=None, reference_unet=base.reference_unet, unet=base.unet, tokenizer=None, text_encoder=None, controlnet=ctrl,
               image_encoder=None, ImgProj=base.ImgProj, scheduler=base.scheduler, safety_checker=None, feature_extractor=None)
W, H = 576, 768
h, w = H // 8, W // 8
g = torch.Generator().manual_seed(0)
B = a.batch
mask = torch.zeros(1, 1, h, w); mask[:, :, h // 4: h * 3 // 4, w // 4: w * 3 // 4] = 1.0     # centred rectangle
kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=W, height=H, num_inference_steps=a.steps,
          guidance_scale=5.0, num_images_per_prompt=B, prompt_embeds=(torch.randn(1, 77, 768, generator=g) * 0.5).to(dev),
          negative_prompt_embeds=(torch.randn(1, 77, 768, generator=g) * 0.5).to(dev),
          ref_clip_hidden_states=(torch.randn(1, 257, 1280, generator=g) * 0.5).to(device=dev, dtype=dt),
          ref_image_latents=torch.randn(1, 4, h, w, generator=g).to(dev), control_image=torch.rand(1, 3, H, W, generator=g).to(dev),
          image_latents=torch.randn(1, 4, h, w, generator=g).to(dev), mask_latents=mask.to(dev),
          noise=torch.randn(B, 4, h, w, generator=g).to(dev), output_type="latent")
out = pipe(**kw).images
torch.cuda.synchronize()
t0 = time.perf_counter()
out = pipe(**kw).images
torch.cuda.synchronize()
dtm = time.perf_counter() - t0
keep = (1 - mask.to(dev)).bool().expand(B, 4, h, w)
# after the last step the un-masked region is the original image latents (the blend of the final step uses them un-noised)
err = (out - kw["image_latents"].expand(B, -1, -1, -1))[keep].abs().max().item()
print(json.dumps(dict(config="768x576 inpaint + ControlNet, bf16", batch=B, steps=a.steps, ms_per_step=round(dtm / a.steps * 1e3, 2),
                      finite=bool(torch.isfinite(out).all()), unmasked_max_dev=err, shape=list(out.shape))))
