"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 checked against the CPU oracle
(reference loop semantics): 4 DDIM steps, 2 images sharing a garment, reduced-width SD1.5-shaped UNets
(head dims 40 / 80 / 160 like the real model), fp16 and bf16."""
import torch


@torch.no_grad()
def run_smoke(device):
    from imagdressing_amd import _lib
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    from tests.harness import SMALL, build_pair, err_stats
    _lib.load()
    g = lambda s, *shape, scale=1.0: torch.randn(*shape, generator=torch.Generator().manual_seed(s)) * scale
    steps, gs = 4, 7.5
    lat = torch.stack([g(42 + i, 4, 16, 16) for i in range(2)])
    pe, ne, cloth, refl = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5), g(12, 2, 16, 64, scale=0.5), g(13, 1, 4, 16, 16)
    ref = None
    for dtype, bar, rms in ((torch.float16, 3e-2, 5e-3), (torch.bfloat16, 0.15, 3e-2)):
        p = build_pair(SMALL, seed=0, device=str(device), dtype=dtype)
        if ref is None:
            ref = torch.cat([denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat[i:i + 1], pe, ne, cloth, refl, steps, gs)
                             for i in range(2)])
        sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                            clip_sample=False, set_alpha_to_one=False, steps_offset=1)
        pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                               image_encoder=None, ImgProj=lambda h: h, scheduler=sch)
        out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128,
                   num_inference_steps=steps, guidance_scale=gs, num_images_per_prompt=2, prompt_embeds=pe.to(device),
                   negative_prompt_embeds=ne.to(device), ref_clip_hidden_states=cloth[1:2].to(device),
                   ref_image_latents=refl.to(device), latents=lat.to(device), output_type="latent").images
        st = err_stats(out, ref)
        assert torch.isfinite(out).all(), "non-finite latents"
        assert st["max_abs"] < bar * max(st["ref_std"], 1.0) and st["rel_rms"] < rms, (str(dtype), st)
        print(f"smoke {dtype}: max_abs={st['max_abs']:.4g} rel_rms={st['rel_rms']:.4g} (ref std {st['ref_std']:.3g})")
