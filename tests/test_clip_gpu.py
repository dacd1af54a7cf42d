"""CLIP text / vision encoders on the HIP kernels against the `transformers` implementation itself
(the library the reference calls, inference_IMAGdressing.py:44-47), instantiated from config with seeded random weights
and run in fp32 on the host.  Bars: fp16 rms 0.5 %, bf16 2.5 % of the output scale (the per-format bars of the UNet)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DTS = pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])


def bars(dt):
    return dict(rel_rms=5e-3, max_rel=4e-2) if dt == torch.float16 else dict(rel_rms=2.5e-2, max_rel=0.2)


def stats(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = (got - ref).abs()
    return dict(max_abs=e.max().item(), ref_std=ref.std().item(), rel_rms=(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())


def check(got, ref, dt, what):
    st = stats(got, ref)
    b = bars(dt)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert st["rel_rms"] < b["rel_rms"] and st["max_abs"] < b["max_rel"] * st["ref_std"], (what, st)


@pytest.fixture(scope="module")
def hf():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    transformers = pytest.importorskip("transformers")
    return transformers


def seeded(model, seed):
    """re-draw every parameter with O(1)-preserving scales (HF's default init gives near-zero attention logits)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n or "layrnorm" in n:
                p.copy_((1.0 + 0.1 * torch.randn(p.shape, generator=g)) if n.endswith("weight") else 0.05 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif "embedding" in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.5 / fan_in ** 0.5))
    return model.eval()


TEXT_SMALL = dict(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                  max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999, projection_dim=64)
VIS_SMALL = dict(hidden_size=160, intermediate_size=320, num_hidden_layers=3, num_attention_heads=2, image_size=56, patch_size=14,
                 projection_dim=64, hidden_act="gelu")


def ids_with_eos(B, T, vocab, eos, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, vocab - 1, (B, T), generator=g)
    for b in range(B):
        ids[b, 5 + 9 * b:] = eos          # prompt of length 5 + 9b, then end-of-text padding (as the CLIP tokenizer pads)
    return ids


@DTS
@torch.no_grad()
def test_clip_text_small_vs_transformers(hf, dt):
    from imagdressing_amd.clip import CLIPTextModel
    ref_m = seeded(hf.CLIPTextModel(hf.CLIPTextConfig(**TEXT_SMALL)), 0)
    eng = CLIPTextModel(ref_m.state_dict(), TEXT_SMALL, "cuda", dt)
    ids = ids_with_eos(3, 77, 1000, 999, 1)
    ref = ref_m(ids, output_hidden_states=True)
    out = eng(ids.cuda(), output_hidden_states=True)
    check(out[0], ref[0], dt, "last_hidden_state")
    check(out.hidden_states[1], ref.hidden_states[1], dt, "hidden_states[1]")
    check(out.pooler_output, ref.pooler_output, dt, "pooler_output")
    # causality: changing a later token must not change earlier positions (bit-exact)
    ids2 = ids.clone(); ids2[:, 40:] = 7
    out2 = eng(ids2.cuda())
    assert torch.equal(out2[0][:, :40], out[0][:, :40]) and not torch.equal(out2[0][:, 40:], out[0][:, 40:])


@DTS
@torch.no_grad()
def test_clip_vision_small_vs_transformers(hf, dt):
    from imagdressing_amd.clip import CLIPVisionModelWithProjection
    ref_m = seeded(hf.CLIPVisionModelWithProjection(hf.CLIPVisionConfig(**VIS_SMALL)), 2)
    eng = CLIPVisionModelWithProjection(ref_m.state_dict(), VIS_SMALL, "cuda", dt)
    px = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(3))
    ref = ref_m(px, output_hidden_states=True)
    out = eng(px.cuda(), output_hidden_states=True)
    assert len(out.hidden_states) == len(ref.hidden_states) == 4
    check(out.hidden_states[0], ref.hidden_states[0], dt, "embeddings + pre-LN")
    check(out.hidden_states[-2], ref.hidden_states[-2], dt, "hidden_states[-2]")
    check(out.image_embeds, ref.image_embeds, dt, "image_embeds")


@torch.no_grad()
def test_clip_text_full_size_vs_transformers(hf):
    """the SD1.5 text tower (CLIP ViT-L/14: 123 M parameters, 12 layers x 12 heads x d 64, quick-GELU), fp16"""
    from imagdressing_amd.clip import TEXT_CONFIG, CLIPTextModel
    cfg = {k: v for k, v in TEXT_CONFIG.items() if k != "layer_norm_eps"}
    ref_m = seeded(hf.CLIPTextModel(hf.CLIPTextConfig(**cfg)), 4)
    assert sum(p.numel() for p in ref_m.parameters()) == 123_060_480
    eng = CLIPTextModel(ref_m.state_dict(), None, "cuda", torch.float16)
    ids = ids_with_eos(2, 77, 49408, 49407, 5)
    check(eng(ids.cuda())[0], ref_m(ids)[0], torch.float16, "text last_hidden_state")


@torch.no_grad()
def test_clip_vision_full_width_vs_transformers(hf):
    """the IP-Adapter image encoder's geometry (OpenCLIP ViT-H/14: hidden 1280, 16 heads x d 80, MLP 5120, 257 tokens, GELU),
    fp16; 6 of the 32 layers so that building the host-side reference does not dominate the GPU session (the layers are
    identical; the full tower has 632,076,800 parameters).  Checks the penultimate hidden state the pipeline consumes
    (IMAGDressing_v1_pipeline.py:404-411) and the projected embedding."""
    from imagdressing_amd.clip import VISION_CONFIG, CLIPVisionModelWithProjection
    cfg = {k: v for k, v in VISION_CONFIG.items() if k not in ("layer_norm_eps", "num_channels")}
    cfg["num_hidden_layers"] = 6
    ref_m = seeded(hf.CLIPVisionModelWithProjection(hf.CLIPVisionConfig(**cfg)), 6)
    per_layer = 4 * (1280 * 1280 + 1280) + 2 * 1280 * 5120 + 5120 + 1280 + 4 * 1280
    assert sum(p.numel() for p in ref_m.parameters()) == 632_076_800 - 26 * per_layer
    eng = CLIPVisionModelWithProjection(ref_m.state_dict(), cfg, "cuda", torch.float16)
    px = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    ref = ref_m(px, output_hidden_states=True)
    out = eng(px.cuda(), output_hidden_states=True)
    assert out.hidden_states[-2].shape == (1, 257, 1280)
    check(out.hidden_states[-2], ref.hidden_states[-2], torch.float16, "vision hidden_states[-2]")
    check(out.image_embeds, ref.image_embeds, torch.float16, "image_embeds")


@torch.no_grad()
def test_pipeline_with_hip_clip_encoders(hf):
    """prompt -> tokenizer -> HIP text encoder and garment CLIP image -> HIP image encoder -> Resampler, through the
    reference pipeline surface (IMAGDressing_v1_pipeline.py:246-262, :395-415): equals passing the embeddings directly"""
    from imagdressing_amd.adapter.resampler import Resampler
    from imagdressing_amd.clip import CLIPTextModel, CLIPVisionModelWithProjection
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from tests.harness import SMALL, build_pair
    dt = torch.float16
    p = build_pair(SMALL, seed=0, dtype=dt)                    # cross_attention_dim 64
    tcfg = dict(TEXT_SMALL, hidden_size=64, num_attention_heads=1, intermediate_size=128)
    text = CLIPTextModel(seeded(hf.CLIPTextModel(hf.CLIPTextConfig(**tcfg)), 8).state_dict(), tcfg, "cuda", dt)
    vis = CLIPVisionModelWithProjection(seeded(hf.CLIPVisionModelWithProjection(hf.CLIPVisionConfig(**VIS_SMALL)), 9).state_dict(),
                                        VIS_SMALL, "cuda", dt)
    torch.manual_seed(3)
    proj = Resampler(dim=64, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=160, output_dim=64, ff_mult=2).to(device="cuda", dtype=dt)

    class Tok:                                    # the CLIPTokenizer surface the pipeline touches
        model_max_length = 77

        def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
            text = [text] if isinstance(text, str) else text
            ids = torch.full((len(text), max_length), 999, dtype=torch.int64)
            for i, s in enumerate(text):
                toks = [1 + (ord(c) % 900) for c in s][:max_length - 1]
                ids[i, :len(toks)] = torch.tensor(toks, dtype=torch.int64)
            import types
            return types.SimpleNamespace(input_ids=ids)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=Tok(), text_encoder=text,
                           image_encoder=vis, ImgProj=proj, scheduler=sch, safety_checker=None, feature_extractor=None)
    g = lambda s, *sh: torch.randn(*sh, generator=torch.Generator().manual_seed(s))     # noqa: E731
    clip_px = g(1, 1, 3, 56, 56).cuda()
    common = dict(width=128, height=128, num_inference_steps=3, guidance_scale=7.5, num_images_per_prompt=1,
                  ref_image_latents=g(2, 1, 4, 16, 16).cuda(), latents=g(3, 1, 4, 16, 16).cuda(), output_type="latent", ref_image=None)
    a = pipe(prompt="a red dress", null_prompt="", negative_prompt="blurry", ref_clip_image=clip_px, **common).images
    pe = text(Tok()("a red dress").input_ids.cuda())[0]
    ne = text(Tok()("blurry").input_ids.cuda())[0]
    hid = vis(clip_px.to(dt), output_hidden_states=True).hidden_states[-2]
    b = pipe(prompt=None, null_prompt=None, negative_prompt=None, prompt_embeds=pe, negative_prompt_embeds=ne,
             ref_clip_hidden_states=hid, **common).images
    assert torch.isfinite(a).all() and torch.equal(a, b)


@torch.no_grad()
def test_main_call_sequence_of_the_reference_script(hf):
    """The ``__main__`` body of /root/reference/inference_IMAGdressing.py:148-189, statement for statement, on the small config: PIL
    garment -> ``resize_img`` (:25-37) -> the torchvision transform (:158-162, restated with PIL + torch: torchvision is not in this
    image) -> ``CLIPImageProcessor`` pixel values (:171; the real transformers class) -> ``pipe(prompt=..., null_prompt=...,
    negative_prompt=..., ref_image=vae_clothes, ref_clip_image=..., width, height, num_images_per_prompt, guidance_scale,
    image_scale, generator, num_inference_steps)`` (:175-187) with NO embeddings / latents injected -> ``.images`` is a list of PIL
    images of the requested size (:189-193 pastes them into a grid).  VAE encode of the garment, CLIP encoders, resampler, garment
    UNet, denoising loop and VAE decode all run on the HIP engines.  Deterministic for a seeded CPU generator."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor
    from imagdressing_amd.adapter.resampler import Resampler
    from imagdressing_amd.clip import CLIPTextModel, CLIPVisionModelWithProjection
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from imagdressing_amd.vae import AutoencoderKL
    from oracle import vae as OV
    from tests.harness import SMALL, build_pair
    dt = torch.float16
    p = build_pair(SMALL, seed=0, dtype=dt)
    tcfg = dict(TEXT_SMALL, hidden_size=64, num_attention_heads=1, intermediate_size=128)
    text = CLIPTextModel(seeded(hf.CLIPTextModel(hf.CLIPTextConfig(**tcfg)), 8).state_dict(), tcfg, "cuda", dt)
    vis = CLIPVisionModelWithProjection(seeded(hf.CLIPVisionModelWithProjection(hf.CLIPVisionConfig(**VIS_SMALL)), 9).state_dict(),
                                        VIS_SMALL, "cuda", dt)
    vcfg = dict(block_out_channels=(64, 128, 128, 128), norm_num_groups=8)
    vae = AutoencoderKL(OV.seeded_state_dict(vcfg, seed=0), vcfg, "cuda", dt)
    torch.manual_seed(3)
    proj = Resampler(dim=64, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=160, output_dim=64, ff_mult=2).to(device="cuda", dtype=dt)

    class Tok:                                    # (no vocabulary files offline) the CLIPTokenizer surface the pipeline touches
        model_max_length = 77

        def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
            import types
            text = [text] if isinstance(text, str) else text
            ids = torch.full((len(text), max_length), 999, dtype=torch.int64)
            for i, s in enumerate(text):
                toks = [1 + (ord(c) % 900) for c in s][:max_length - 1]
                ids[i, :len(toks)] = torch.tensor(toks, dtype=torch.int64)
            return types.SimpleNamespace(input_ids=ids)
    noise_scheduler = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                    clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = IMAGDressing_v1(unet=p["e_unet"], reference_unet=p["e_ref"], vae=vae, tokenizer=Tok(), text_encoder=text, image_encoder=vis,
                           ImgProj=proj, scheduler=noise_scheduler, safety_checker=None, feature_extractor=CLIPImageProcessor)

    def resize_img(input_image, max_side=192, min_side=128, mode=Image.BILINEAR, base_pixel_number=64):      # :25-37 (640 x 512 -> 192 x 128 for the small config: SD latents need image sides divisible by 64)
        w, h = input_image.size
        ratio = min_side / min(h, w)
        w, h = round(ratio * w), round(ratio * h)
        ratio = max_side / max(h, w)
        input_image = input_image.resize([round(ratio * w), round(ratio * h)], mode)
        w_resize_new = (round(ratio * w) // base_pixel_number) * base_pixel_number
        h_resize_new = (round(ratio * h) // base_pixel_number) * base_pixel_number
        return input_image.resize([w_resize_new, h_resize_new], mode)

    def img_transform(img):                       # transforms.Compose([Resize([H, W], BILINEAR), ToTensor(), Normalize([0.5], [0.5])]) (:158-162)
        img = img.resize((128, 192), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - 0.5) / 0.5

    # ================= body of __main__, inference_IMAGdressing.py:148-189 =================
    num_samples = 1
    clip_image_processor = CLIPImageProcessor(size={"shortest_edge": 56}, crop_size={"height": 56, "width": 56})      # (the small vision tower's input size)
    prompt = 'A beautiful woman, best quality, high quality'
    null_prompt = ''
    negative_prompt = 'bare, naked, nude, undressed, monochrome, lowres, bad anatomy, worst quality, low quality'
    rng = np.random.RandomState(0)
    clothes_img = Image.fromarray(rng.randint(0, 255, (300, 240, 3), dtype=np.uint8)).convert("RGB")      # Image.open(args.cloth_path).convert("RGB")
    clothes_img = resize_img(clothes_img)
    vae_clothes = img_transform(clothes_img).unsqueeze(0)
    ref_clip_image = clip_image_processor(images=clothes_img, return_tensors="pt").pixel_values

    def run():
        generator = torch.Generator(device="cpu").manual_seed(42)
        return pipe(
            ref_image=vae_clothes,
            prompt=prompt,
            ref_clip_image=ref_clip_image,
            null_prompt=null_prompt,
            negative_prompt=negative_prompt,
            width=128,
            height=192,
            num_images_per_prompt=num_samples,
            guidance_scale=7.5,
            image_scale=1.0,
            generator=generator,
            num_inference_steps=6,
        ).images
    output = run()
    # ================= end =================
    assert isinstance(output, list) and len(output) == num_samples and isinstance(output[0], Image.Image)
    assert output[0].size == (128, 192) and output[0].mode == "RGB"
    save_output = [clothes_img.resize((128, 192), Image.BICUBIC), output[0]]          # :190-193 (image_grid pastes them side by side)
    grid = Image.new("RGB", size=(2 * 128, 192))
    for i, img in enumerate(save_output):
        grid.paste(img, box=(i * 128, 0))
    arr = np.asarray(output[0])
    assert arr.std() > 1.0                        # not a constant image
    assert np.array_equal(arr, np.asarray(run()[0]))      # same seed -> same picture (no atomics, fixed-order reductions)
