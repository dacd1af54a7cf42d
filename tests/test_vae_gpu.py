"""SD1.5 VAE on the HIP kernels against the fp32 CPU restatement (oracle/vae.py) on identical
seeded weights, plus the two kernels it adds (row softmax, bottom/right-only padded stride-2 conv).
Bars: fp16 atol 1e-2 x output std (rms 0.5 %), bf16 rms 2.5 % -- the same per-format bars as the UNet."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SMALL = dict(block_out_channels=(64, 128, 128, 128), norm_num_groups=8)
DTS = pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])


def bars(dt):
    return dict(rel_rms=5e-3, max_rel=2.5e-2) if dt == torch.float16 else dict(rel_rms=2.5e-2, max_rel=0.15)


def stats(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = (got - ref).abs()
    return dict(max_abs=e.max().item(), ref_std=ref.std().item(), rel_rms=(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())


def rnd(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd import ops
    return ops


@DTS
@torch.no_grad()
def test_vae_small_encode_decode_vs_oracle(gpu, dt):
    from imagdressing_amd.vae import AutoencoderKL
    from oracle import vae as OV
    sd = OV.seeded_state_dict(SMALL, seed=0)
    o = OV.AutoencoderKL(SMALL); o.load_state_dict(sd, strict=True)
    e = AutoencoderKL(sd, SMALL, "cuda", dt)
    img = rnd(1, 2, 3, 64, 64).clamp(-1, 1)
    mean_o, logvar_o = o.encode_moments(img)
    dist = e.encode(img.cuda()).latent_dist
    b = bars(dt)
    for name, got, ref in (("mean", dist.mean, mean_o), ("logvar", dist.logvar, logvar_o)):
        st = stats(got, ref)
        assert got.shape == ref.shape and st["rel_rms"] < b["rel_rms"] and st["max_abs"] < b["max_rel"] * st["ref_std"], (name, st)
    z = rnd(2, 2, 4, 8, 8)
    ref = o.decode(z)
    got = e.decode(z.cuda(), return_dict=False)[0]
    assert got.shape == ref.shape == (2, 3, 64, 64) and got.dtype == dt
    st = stats(got, ref)
    assert st["rel_rms"] < b["rel_rms"] and st["max_abs"] < b["max_rel"] * st["ref_std"], st
    # slicing is a memory knob, not a numerical one
    e.enable_slicing()
    assert torch.equal(e.decode(z.cuda(), return_dict=False)[0], got)
    # sampling surface: mode() is the mean, sample() adds std * noise
    assert torch.equal(dist.mode(), dist.mean)
    s = dist.sample(generator=torch.Generator(device="cuda").manual_seed(0))
    assert s.shape == dist.mean.shape and torch.isfinite(s).all()


@pytest.fixture(scope="module")
def full_vae_oracle(gpu):
    """(state dict, latent, fp32 oracle decode) of the real 49.5 M-parameter decoder, computed once for both dtypes"""
    from oracle import vae as OV
    sd = OV.seeded_state_dict(None, seed=3)
    o = OV.AutoencoderKL(); o.load_state_dict(sd, strict=True)
    z = rnd(4, 1, 4, 32, 32)
    with torch.no_grad():
        ref = o.decode(z)
    return sd, z, ref


@DTS
@torch.no_grad()
def test_vae_full_width_decode_vs_oracle(full_vae_oracle, dt):
    """the real decoder (channels 512/512/256/128, d = 512 mid attention) on a 32x32 latent -> 256x256"""
    from imagdressing_amd.vae import AutoencoderKL
    sd, z, ref = full_vae_oracle
    e = AutoencoderKL(sd, None, "cuda", dt)
    got = e.decode(z.cuda(), return_dict=False)[0]
    st = stats(got, ref)
    b = bars(dt)
    assert got.shape == (1, 3, 256, 256) and torch.isfinite(got).all()
    assert st["rel_rms"] < b["rel_rms"] and st["max_abs"] < b["max_rel"] * st["ref_std"], st


@DTS
def test_softmax_rows(gpu, dt):
    ops = gpu
    for rows, cols in ((64, 64), (300, 1000), (128, 4096), (32, 6912)):
        s = (rnd(5, rows, cols) * 4.0).cuda()
        s[0, :] = 50.0 * torch.arange(cols, device="cuda") / cols          # a spiky row
        p = ops.softmax_rows(s, dtype=dt)
        ref = torch.softmax(s.float(), dim=-1)
        assert p.dtype == dt and (p.float() - ref).abs().max() < (4e-3 if dt == torch.bfloat16 else 5e-4)
        assert (p.float().sum(-1) - 1).abs().max() < (2e-2 if dt == torch.bfloat16 else 2e-3)
    wide = torch.zeros(64, 128, dtype=dt, device="cuda")
    ops.softmax_rows((rnd(6, 64, 100)).cuda(), out=wide)
    assert (wide[:, 100:] == 0).all() and (wide[:, :100].float().sum(-1) - 1).abs().max() < 2e-2
    with pytest.raises(ops.L.ImdError):
        ops.softmax_rows(torch.zeros(2, 20000, device="cuda"))


@DTS
def test_conv_stride2_bottom_right_padding(gpu, dt):
    """F.pad(x, (0, 1, 0, 1)) + conv(stride 2, padding 0): the VAE encoder's Downsample2D"""
    ops = gpu
    B, H, W, Cin, Cout = 2, 16, 24, 64, 128
    x = rnd(7, B, Cin, H, W).to(dt); w = rnd(8, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt); b = rnd(9, Cout)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b, stride=2).permute(0, 2, 3, 1)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    got = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wp.cuda(), b.cuda(), stride=2, pad_br_only=True)
    assert got.shape == ref.shape
    tol = 1e-2 if dt == torch.bfloat16 else 2e-3
    assert (got.float().cpu() - ref).abs().max() < tol + tol * ref.abs().max()
    sym = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().cuda(), wp.cuda(), b.cuda(), stride=2)
    assert (sym.float() - got.float()).abs().max() > 0.1         # and it is not the symmetric-padding conv


@torch.no_grad()
def test_pipeline_with_hip_vae(gpu):
    """reference call sites IMAGDressing_v1_pipeline.py:457-458 (garment encode) and :544 (decode) through the HIP VAE"""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from imagdressing_amd.vae import AutoencoderKL
    from tests.harness import SMALL as USMALL, build_pair
    dt = torch.float16
    p = build_pair(USMALL, seed=0, dtype=dt)
    vae = AutoencoderKL.random_init(seed=5, config=SMALL, device="cuda", dtype=dt)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)

    class Proj:
        def __call__(self, h):
            return h
    pipe = IMAGDressing_v1(vae=vae, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=Proj(), scheduler=sch, safety_checker=None, feature_extractor=None)
    assert pipe.vae_scale_factor == 8
    garment = rnd(10, 1, 3, 128, 128).clamp(-1, 1).cuda()
    kw = dict(prompt=None, null_prompt=None, negative_prompt=None, width=128, height=128, num_inference_steps=4,
              guidance_scale=7.5, num_images_per_prompt=2, prompt_embeds=rnd(11, 1, 77, 64, scale=0.5).cuda(),
              negative_prompt_embeds=rnd(12, 1, 77, 64, scale=0.5).cuda(), ref_clip_hidden_states=rnd(13, 1, 16, 64, scale=0.5).cuda(),
              latents=rnd(14, 2, 4, 16, 16).cuda())
    out = pipe(ref_image=garment, output_type="pt", **kw).images
    assert out.shape == (2, 3, 128, 128) and torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    # the garment latent the pipeline used is encode(garment).mean * 0.18215
    lat = vae.encode(garment).latent_dist.mean * 0.18215
    out2 = pipe(ref_image=None, ref_image_latents=lat, output_type="pt", **kw).images
    assert torch.equal(out, out2)
