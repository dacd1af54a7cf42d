"""Parity at BASELINE.json's FULL sizes (SD1.5 widths 320/640/1280, 512x512 -> 64x64 latent, batch 4 sharing a garment).

Two kinds of checks:
  * one full-width UNet forward against the fp32 CPU oracle (the oracle finishes it in seconds at batch 1);
  * size-independent properties of the path, which need no oracle at all: determinism, batched == sharded == single
    image generation, softmax rows summing to one, key-order invariance, exact power-of-two linearity of the
    convolutions, tile-config independence of the GEMM, unit statistics after GroupNorm, DDIM / CFG identities.
Tolerances are written in each test (bit-exact where the property is exact in floating point)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd import ops as o
    return o


def rnd(seed, *shape, scale=1.0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(device)


# ----------------------------------------------------------------------------------------------------------------
# full-width UNet forward vs the fp32 CPU oracle
# ----------------------------------------------------------------------------------------------------------------
# latent (rows, cols): 512x512 -> 64x64 (BASELINE configs[1]); the reference scripts' own default, width 512 x height 640 with a
# 640x512 garment (inference_IMAGdressing.py:182-183) -> 80x64, N = M = 5120 / 1280 / 320 / 80
@pytest.fixture(scope="module", params=[(64, 64), (80, 64)], ids=["512x512", "512x640"])
def full_oracle(request):
    """(state dict, inputs, fp32 oracle output) of ONE full-width SD1.5 UNet forward, batch 1."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lh, lw = request.param
    from imagdressing_amd import unet as E
    from oracle import processors as OP
    from oracle import sd15
    sd = E.random_state_dict(E.unet_param_shapes(E.SD15_CONFIG), 0)
    o = sd15.UNet2DConditionModel({})
    o.load_state_dict(sd, strict=True)
    boc = E.SD15_CONFIG["block_out_channels"]
    from tests.harness import hidden_size_of
    o.set_attn_processor({n: (OP.RefSAttn(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                              else OP.CAttn(n, hidden_size_of(n, boc), 768)) for n in o.attn_processors.keys()})
    x = rnd(1, 1, 4, lh, lw)
    ehs = rnd(2, 1, 77, 768, scale=0.5)
    # garment tokens of every attn1 layer ([1, M_l, C_l], M_l = N_l: garment at the generation resolution) and seeded to_k_ref / to_v_ref
    names = [n for n in o.attn_processors.keys() if n.endswith("attn1.processor")]
    from tests.harness import ref_weights
    rw = ref_weights(names, boc, 7)
    tokens = {320: lh * lw, 640: lh * lw // 4, 1280: lh * lw // 16}
    sa = {}
    for j, n in enumerate(names):
        c = hidden_size_of(n, boc)
        m = lh * lw // 64 if n.startswith("mid_block") else tokens[c]
        sa[n] = rnd(100 + j, 1, m, c)
        with torch.no_grad():
            o.attn_processors[n].to_k_ref.weight.copy_(rw[n]["k"]); o.attn_processors[n].to_v_ref.weight.copy_(rw[n]["v"])
    with torch.no_grad():
        ref = o(x, 481, ehs)
        ref_cond = o(x, 481, ehs, cross_attention_kwargs={"sa_hidden_states": sa})
    del o
    return dict(sd=sd, x=x, ehs=ehs, ref=ref, ref_cond=ref_cond, sa=sa, rw=rw)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_full_width_unet_forward_vs_oracle(full_oracle, dtype):
    """859.5 M-parameter UNet, 64x64 latent (BASELINE configs[1]) and 80x64 (the reference scripts' default 512x640): the HIP
    engine against the fp32 oracle on identical seeded weights.
    Bars: fp16 rms 0.5 % and worst element 2e-2 x output std; bf16 rms 2.5 % / 0.12 x std (8 mantissa bits)."""
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter import attention_processor as AP
    from tests.harness import err_stats, hidden_size_of
    fo = full_oracle
    e = E.UNet2DConditionModel(fo["sd"], {}, "cuda", dtype)
    boc = E.SD15_CONFIG["block_out_channels"]
    e.set_attn_processor({n: (AP.RefSAttnProcessor2_0(n, hidden_size_of(n, boc)) if n.endswith("attn1.processor")
                              else AP.CAttnProcessor2_0(n, hidden_size_of(n, boc), 768)) for n in e.attn_processors.keys()})
    got = e(fo["x"].cuda(), 481, fo["ehs"].cuda())[0]
    st = err_stats(got, fo["ref"])
    assert torch.isfinite(got).all()
    bar = dict(rel_rms=5e-3, max_rel=2e-2) if dtype == torch.float16 else dict(rel_rms=2.5e-2, max_rel=0.12)
    assert st["rel_rms"] < bar["rel_rms"] and st["max_abs"] < bar["max_rel"] * st["ref_std"], st
    # ---- the COND pass: garment branch on in all 16 hybrid blocks (N = M = 4096 / 1024 / 256 / 64 or 5120 / 1280 / 320 / 80), and the pipeline's
    # CFG layout -- [cond; uncond] rows in one call, garment switched per row (sa_batch_mask) -- against the two oracle passes
    for n, p in e.attn_processors.items():
        if n.endswith("attn1.processor"):
            p.to_k_ref.weight.copy_(fo["rw"][n]["k"]); p.to_v_ref.weight.copy_(fo["rw"][n]["v"])
    sa = {n: t.cuda() for n, t in fo["sa"].items()}
    x2 = torch.cat([fo["x"], fo["x"]]).cuda()
    both = e(x2, 481, fo["ehs"].cuda(), cross_attention_kwargs={"sa_hidden_states": sa,
                                                                  "sa_batch_mask": torch.tensor([1.0, 0.0], device="cuda")})[0]
    st_c, st_u = err_stats(both[0:1], fo["ref_cond"]), err_stats(both[1:2], fo["ref"])
    for st2 in (st_c, st_u):
        assert st2["rel_rms"] < bar["rel_rms"] and st2["max_abs"] < bar["max_rel"] * st2["ref_std"], (st_c, st_u)
    # the garment branch matters at this size (the cond and uncond oracle passes differ by 11 % rms, several times the error bar)
    gap = err_stats(fo["ref_cond"], fo["ref"])
    assert gap["rel_rms"] > 3 * bar["rel_rms"] and gap["rel_rms"] > 0.05, gap
    del e
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------------------------
# full-width forwards at the ENDS of the schedule and with the configs[2] stack, against the committed fp32-oracle fixture
# ----------------------------------------------------------------------------------------------------------------
_TS_INPUTS = {}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_full_width_unet_forward_at_three_timesteps_vs_committed_oracle(dtype):
    """tests/golden/unet_forward_timesteps.pt (oracle/make_golden.py timesteps): t = 981 (first step of the 50-step schedule,
    eps ~ z), 481, 1 (last step), for (a) the configs[1] processors and (b) the configs[2] stack -- LoraRefS + LoRAIP (rank 128,
    77 + 4 tokens) with the residuals of the engine's own pose ControlNet -- CFG layout, full width, 64x64 latent.
    Bars: fp16 the north-star atol 1e-2 on every element of eps at every timestep; bf16 rms 2.5 % / worst element 0.12 x std
    (8 mantissa bits)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    from tests.unet_fixture import GOLDEN, ipa_controlnet_forward_inputs, measure_unet_parity_timesteps, unet_forward_inputs
    if "base" not in _TS_INPUTS:
        full = torch.load(os.path.join(GOLDEN, "unet_forward_full.pt"), weights_only=False)["latent_64x64"]
        gold = torch.load(os.path.join(GOLDEN, "unet_forward_timesteps.pt"), weights_only=False)
        _TS_INPUTS["base"] = unet_forward_inputs(64, 64, full)
        _TS_INPUTS["ipa"] = ipa_controlnet_forward_inputs(gold["ipa_controlnet"], base=_TS_INPUTS["base"])
    res = measure_unet_parity_timesteps(torch.device("cuda"), dtype, inputs=_TS_INPUTS["base"], ipa_inputs=_TS_INPUTS["ipa"])
    for case in ("refs", "ipa_controlnet"):
        for t in (981, 481, 1):
            for half in ("cond", "uncond"):
                st = res[case][f"t{t}"][half]
                assert res[case][f"t{t}"]["finite"]
                if dtype == torch.float16:
                    assert st["max_abs"] <= 1e-2 and st["rel_rms"] < 5e-3, (case, t, half, st)
                else:
                    assert st["rel_rms"] < 2.5e-2 and st["max_abs"] < 0.12 * st["ref_std"], (case, t, half, st)
    if dtype == torch.float16:
        assert res["refs"]["meets_atol_1e-2"] and res["ipa_controlnet"]["meets_atol_1e-2"]


# ----------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] at FULL width: IP-Adapter FaceID-Plus tokens + rank-128 LoRA on every attention + pose ControlNet
# ----------------------------------------------------------------------------------------------------------------
_IPA_ORACLE = {}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_full_width_ipa_lora_controlnet_pipeline_vs_oracle(dtype):
    """inference_IMAGdressing_ipa_controlnetpose.py's path at the real SD1.5 widths on a 64x64 latent: LoraRefS (rank 128,
    folded into the weights) + LoRAIP processors (77 text + 4 face tokens), ControlNet-OpenPose residuals split into the
    cond / uncond halves, custom CFG (g = 7.0) and DDIM -- two images sharing the garment through TWO denoising steps of the
    HIP pipeline against the reference loop semantics on the fp32 oracle (oracle/pipeline.py, one image at a time).
    Bars as for the single full-width UNet forward, relative to the oracle's final latent."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_ipa_controlnet import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    from tests.harness import build_pair, err_stats
    p = build_pair({}, seed=3, kind="ipa", with_controlnet=True, dtype=dtype, rank=128)
    steps, gs = 2, 7.0
    lat = torch.stack([rnd(42 + i, 4, 64, 64) for i in range(2)])
    pe, ne = rnd(10, 1, 77, 768, scale=0.5), rnd(11, 1, 77, 768, scale=0.5)
    face_p, face_n = rnd(14, 1, 4, 768, scale=0.5), rnd(15, 1, 4, 768, scale=0.5)
    cloth = rnd(12, 2, 16, 768, scale=0.5); refl = rnd(13, 1, 4, 64, 64)
    pose = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(16))
    if "ref" not in _IPA_ORACLE:       # the fp32 oracle trajectory does not depend on the engine's element type: once per session
        _IPA_ORACLE["ref"] = torch.cat([
            denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat[i:i + 1], torch.cat([pe, face_p], 1), torch.cat([ne, face_n], 1),
                    cloth, refl, steps, gs, controlnet=p["o_ctrl"], control_image=pose,
                    prompt_embeds_control=torch.cat([ne, pe]), conditioning_scale=0.8) for i in range(2)])
    ref = _IPA_ORACLE["ref"]
    for k in ("o_unet", "o_ref", "o_ctrl"):
        p[k] = None

    class FaceProj:      # image_proj_model stand-in returning the given face tokens
        def __call__(self, idv, clip):
            return (face_p if float(idv.abs().sum()) > 0 else face_n).cuda()
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           controlnet=p["e_ctrl"], image_encoder=None, ImgProj=lambda h: h, ip_ckpt=None, scheduler=sch)
    pipe.image_proj_model = FaceProj()
    out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=512, height=512,
               num_inference_steps=steps, guidance_scale=gs, pose_image=pose.cuda(), faceid_embeds=torch.ones(1, 512),
               face_clip_hidden_states=torch.zeros(1, 257, 1280), face_uncond_clip_hidden_states=torch.zeros(1, 257, 1280),
               image_scale=1.0, ipa_scale=0.9, s_lora_scale=0.2, c_lora_scale=0.2, controlnet_conditioning_scale=0.8,
               num_images_per_prompt=2, prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(),
               ref_clip_hidden_states=cloth[1:2].cuda(), ref_image_latents=refl.cuda(), latents=lat.cuda(), output_type="latent").images
    st = err_stats(out, ref)
    assert torch.isfinite(out).all()
    bar = dict(rel_rms=5e-3, max_rel=2.5e-2) if dtype == torch.float16 else dict(rel_rms=2.5e-2, max_rel=0.15)
    assert st["rel_rms"] < bar["rel_rms"] and st["max_abs"] < bar["max_rel"] * st["ref_std"], st
    assert (ref[0] - ref[1]).pow(2).mean().sqrt() > 0.2 * ref.std()         # two different images
    del pipe, p
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] geometry at FULL width: ControlNet-inpainting at 768 x 576 (latent 96 x 72, N = 6912 / 1728 / 432 / 108)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fp8", [False, True], ids=["attn16", "attn-fp8"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_full_width_inpaint_768x576_pipeline_vs_oracle(dtype, fp8):
    """inference_IMAGdressing_controlnetinpainting.py's path at the real widths and the real geometry: level-0 hybrid attention
    over N = M = 6912 tokens (108 key blocks: the d = 40 kernel's odd / ragged-unit tail), 3x3 convolutions on 96 x 72, 48 x 36,
    24 x 18 and 12 x 9 maps (ragged halo-patch tiles), ControlNet residuals, custom CFG (g = 5.0), DDIM and the per-step
    masked blend with the re-noised original latents -- one image, two steps, against the reference loop on the fp32 oracle."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    from imagdressing_amd import ops as O
    from tests.harness import build_pair, err_stats
    p = build_pair({}, seed=5, with_controlnet=True, dtype=dtype)
    steps, gs = 2, 5.0
    h, w = 96, 72
    noise = rnd(42, 1, 4, h, w)
    pe, ne = rnd(10, 1, 77, 768, scale=0.5), rnd(11, 1, 77, 768, scale=0.5)
    cloth = rnd(12, 2, 16, 768, scale=0.5); refl = rnd(13, 1, 4, h, w)
    img_lat = rnd(17, 1, 4, h, w)
    mask = torch.zeros(1, 1, h, w); mask[:, :, 18:78, 13:59] = 1.0                # centred rectangle, 40 % of the area
    ctrl = torch.rand(1, 3, 8 * h, 8 * w, generator=torch.Generator().manual_seed(18))
    if "ref" not in _INPAINT_ORACLE:
        _INPAINT_ORACLE["ref"] = denoise(p["o_unet"], p["o_ref"], DDIMOracle(), noise, pe, ne, cloth, refl, steps, gs,
                                         controlnet=p["o_ctrl"], control_image=ctrl, prompt_embeds_control=torch.cat([ne, pe]),
                                         inpaint=dict(mask=mask, image_latents=img_lat, noise=noise))
    ref = _INPAINT_ORACLE["ref"]
    for k in ("o_unet", "o_ref", "o_ctrl"):
        p[k] = None
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           controlnet=p["e_ctrl"], image_encoder=None, ImgProj=lambda x: x, scheduler=sch)
    O.ATTN_FP8 = fp8            # configs[4] proper: the level-0 (d = 40) hybrid attention on the MX-FP8 MFMA
    try:
        out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=8 * w, height=8 * h,
                   num_inference_steps=steps, guidance_scale=gs, num_images_per_prompt=1, control_image=ctrl.cuda(),
                   prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(),
                   ref_image_latents=refl.cuda(), image_latents=img_lat.cuda(), mask_latents=mask.cuda(), noise=noise.cuda(),
                   output_type="latent").images
    finally:
        O.ATTN_FP8 = False
    st = err_stats(out, ref)
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_stats.jsonl", "a") as f:
        f.write(json.dumps(dict(test=f"full_width_inpaint_768x576[{dtype},fp8_attention={fp8}]", **st)) + "\n")
    assert torch.isfinite(out).all() and tuple(out.shape) == (1, 4, h, w)
    # bf16: the rms bar is the statistic that matters (2.0 % in rounds 2 and 3); the single worst of the 27,648 values at the end of the
    # trajectory moves with the fp32 summation order of any kernel (0.31 ... 0.36 x sigma = 0.14 ... 0.16 of the latent scale across
    # the tile / K-split choices of the tuning table), hence 0.2
    bar = dict(rel_rms=5e-3, max_rel=2.5e-2) if dtype == torch.float16 else dict(rel_rms=2.5e-2, max_rel=0.2)
    if fp8:     # e4m3 in the five level-0 hybrid blocks (x 2 UNets: the garment UNet stays 16-bit): budget measured, see DESIGN.md
        bar = dict(rel_rms=4e-2, max_rel=0.3)
    assert st["rel_rms"] < bar["rel_rms"] and st["max_abs"] < bar["max_rel"] * st["ref_std"], st
    keep = (mask == 0).expand(1, 4, -1, -1)       # outside the mask: exactly the original latents after the last step (:494-500)
    assert torch.allclose(out.cpu()[keep], img_lat[keep], atol=1e-5)
    del pipe, p
    torch.cuda.empty_cache()


_INPAINT_ORACLE = {}


# ----------------------------------------------------------------------------------------------------------------
# the BASELINE configs[1] pipeline: determinism, batched == sharded generation
# ----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def full_pipe(request):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import argparse

    import bench
    dev = torch.device("cuda", 0)
    pipe = bench.build_pipeline(dev, request.param, 0)
    inp = bench.synthetic_inputs(512, 512, 4, dev, request.param, 0, 1)
    yield pipe, inp, request.param
    del pipe
    torch.cuda.empty_cache()


def run_pipe(pipe, inp, sel, steps=50):
    kw = dict(inp)
    kw["latents"] = inp["latents"][sel]
    return pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=512, height=512,
                num_inference_steps=steps, guidance_scale=7.5, num_images_per_prompt=kw["latents"].shape[0],
                output_type="latent", **kw).images.float()


@torch.no_grad()
def test_full_pipeline_deterministic(full_pipe):
    """50 DDIM steps, 4 images at 512x512: no atomics, fixed-order split-K -> two runs are bit-identical."""
    pipe, inp, _ = full_pipe
    a = run_pipe(pipe, inp, slice(0, 4))
    b = run_pipe(pipe, inp, slice(0, 4))
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)


@torch.no_grad()
def test_full_pipeline_first_hybrid_block_once_per_image_is_bit_identical(full_pipe):
    """Round 6: in ``down_blocks.0.attentions.0`` the cond and uncond rows of an image still carry identical hidden states, so norm ->
    proj_in -> norm1 -> q / k / v -> the self-attention phase run once per image and the attention launch stores that phase twice
    (``Transformer2D.call_pair_half``, ``imd_attn_params.out_dup``: bit-identical at the kernel level,
    ``test_full_attention_duplicated_first_phase_is_bit_identical``).  At the pipeline level the two forms run the SAME arithmetic through
    different launch geometries (half-batch GroupNorm statistics come from the producing convolution's epilogue instead of a statistics
    pass, the half-batch projections pick other tile configurations): they differ in fp32 summation order only, i.e. like two
    realisations of the 16-bit rounding noise -- the bars of ``test_full_pipeline_batched_equals_sharded``.  The path is really taken
    (the first level-0 attention launch runs B instead of 2B rows: seen through the hook)."""
    from imagdressing_amd import ops
    pipe, inp, _ = full_pipe
    assert ops.CFG_PAIR_ATTN
    seen = []
    ops.ATTN_EVENT_HOOK = {"match": lambda **kw: (seen.append((kw["B"], kw["N"], kw["L2"])) or False), "events": []}
    try:
        a = run_pipe(pipe, inp, slice(0, 4), steps=5)
    finally:
        ops.ATTN_EVENT_HOOK = None
    assert (4, 4096, 4096) in seen and (8, 4096, 4096) in seen            # the de-duplicated first block and the ordinary ones
    ops.CFG_PAIR_ATTN = False
    seen.clear()
    ops.ATTN_EVENT_HOOK = {"match": lambda **kw: (seen.append((kw["B"], kw["N"], kw["L2"])) or False), "events": []}
    try:
        b = run_pipe(pipe, inp, slice(0, 4), steps=5)
    finally:
        ops.ATTN_EVENT_HOOK = None
    assert (4, 4096, 4096) not in seen
    assert torch.isfinite(a).all()
    bar = 4e-2 if full_pipe[2] == torch.bfloat16 else 6e-3
    scale = b.pow(2).mean().sqrt()
    assert (a - b).pow(2).mean().sqrt() < bar * scale, ((a - b).pow(2).mean().sqrt() / scale).item()


@torch.no_grad()
def test_full_pipeline_batched_equals_sharded(full_pipe):
    """Batched generation is DEFINED as independent runs: the 4-image batch, two 2-image shards
    (what 2 ranks compute, imagdressing_amd/dist.py::shard_bounds) and a single-image run agree.  They differ only in
    fp32 summation order (tile / split-K choices depend on M), which re-rolls the 16-bit output roundings: two such
    runs differ like two independent realisations of the format's rounding noise, i.e. by ~sqrt(2) x (engine vs fp32
    oracle).  Bars (rms of the final latent after 20 steps): fp16 6e-3, bf16 4e-2 (measured 2.0e-2)."""
    from imagdressing_amd.dist import shard_bounds
    pipe, inp, dtype = full_pipe
    steps = 20
    bar = 4e-2 if dtype == torch.bfloat16 else 6e-3
    full = run_pipe(pipe, inp, slice(0, 4), steps=steps)
    shards = [run_pipe(pipe, inp, slice(*shard_bounds(4, r, 2)), steps=steps) for r in range(2)]
    both = torch.cat(shards)
    one = run_pipe(pipe, inp, slice(2, 3), steps=steps)
    scale = full.pow(2).mean().sqrt()
    assert (both - full).pow(2).mean().sqrt() < bar * scale, ((both - full).pow(2).mean().sqrt() / scale).item()
    assert (one - full[2:3]).pow(2).mean().sqrt() < bar * scale, ((one - full[2:3]).pow(2).mean().sqrt() / scale).item()
    # different seeds really give different images (the comparison above is not vacuous)
    assert (full[0] - full[1]).pow(2).mean().sqrt() > 0.2 * scale


# ----------------------------------------------------------------------------------------------------------------
# kernel-level properties at the level-0 shapes of the CFG batch
# ----------------------------------------------------------------------------------------------------------------
def attn_operands(ops, dt, B=8, H=8, N=4096, M=4096, D=40, seed=0):
    dpk, dpv = ops.attn_padded_dims(D)
    g = torch.Generator(device="cuda").manual_seed(seed)

    def r(*s):
        return torch.randn(*s, generator=g, device="cuda").to(dt)
    q = torch.zeros(B, H, N, dpk, dtype=dt, device="cuda"); q[..., :D] = r(B, H, N, D) * (D ** -0.5 * math.log2(math.e))
    k = ops.k_buffer((B, H, N, dpk), D, dt, "cuda"); k[..., :D] = r(B, H, N, D)       # pad column D = 1 (k_pad_one)
    vt = torch.zeros(B, H, dpv, ops.pad64(N), dtype=dt, device="cuda"); vt[:, :, :D, :N] = r(B, H, D, N)
    kr = ops.k_buffer((1, H, M, dpk), D, dt, "cuda"); kr[..., :D] = r(1, H, M, D)
    vr = torch.zeros(1, H, dpv, ops.pad64(M), dtype=dt, device="cuda"); vr[:, :, :D, :M] = r(1, H, D, M)
    s2 = torch.cat([torch.ones(B // 2), torch.zeros(B // 2)]).cuda()
    return q, k, vt, kr, vr, s2


def run_attn(ops, q, k, vt, kr, vr, s2, D=40, k_pad_one=True):
    """k_pad_one=True: the LDS-DMA staging of the d = 40 kernel (what the processors run); False: its register staging."""
    B, H, N = q.shape[:3]
    M = kr.shape[2]
    out = torch.empty(B, N, H * D, dtype=q.dtype, device="cuda")
    ops.attention(q, k, vt, out, B=B, H=H, N=N, D=D, L1=N, L1P=ops.pad64(N), k2=kr, v2t=vr, scale2=s2, L2=M,
                  L2P=ops.pad64(M), kv2_bdiv=B, k_pad_one=k_pad_one)
    return out


def attn_oracle(q, k, vt, kr, vr, s2, D, dt):
    """fp32 CPU evaluation of the kernel's contract on the SAME 16-bit operands:
    out[b, n, h*D:(h+1)*D] = softmax2(q k^T) v  (rounded to the element type, like the reference's first SDPA output,
    adapter/attention_processor.py:589-594)  +  s2[b] * softmax2(q k_ref^T) v_ref   (:607-612); softmax2 = base-2 softmax
    (q carries d^-1/2 log2 e)."""
    B, H, N, _ = q.shape
    M = kr.shape[2]
    ln2 = math.log(2.0)
    qf, kf, vf = q.float().cpu(), k.float().cpu(), vt.float().cpu()
    krf, vrf, s2c = kr.float().cpu(), vr.float().cpu(), s2.cpu()
    out = torch.empty(B, N, H * D)
    for b in range(B):
        for h in range(H):
            qq = qf[b, h, :, :D] * ln2
            o = torch.softmax(qq @ kf[b, h, :, :D].t(), dim=-1) @ vf[b, h, :D, :N].t()
            o = o.to(dt).float()
            if s2c[b] != 0:
                o = o + s2c[b] * (torch.softmax(qq @ krf[0, h, :M, :D].t(), dim=-1) @ vrf[0, h, :D, :M].t())
            out[b, :, h * D:(h + 1) * D] = o
    return out


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("N,M,spike", [(4096, 4096, False), (4096, 4096, True), (1000, 700, True)],
                         ids=["baseline-shape", "baseline-shape-spiked", "ragged-spiked"])
def test_full_attention_benchmarked_instantiation_vs_oracle(ops, dt, N, M, spike):
    """THE kernel bench.py's roofline times -- head dim 40, N >= 512 (two query blocks per wave, speculative exp), two phases
    on the cond rows of the B = 8 CFG batch, XCD-aware work list -- against the fp32 oracle on identical operands, every
    output element.  Scores have std 3 (peaked rows, outputs O(1)), not the flat softmax of N(0,1) scores.  The spiked
    variants plant keys whose scores exceed everything before them by 2^20 and more, late in both key sets (self: key
    N-130, garment: key M-70; the ragged case ends both in partial tiles), so the speculative block fails its bound and the
    exact redo path + O rescale run in the middle of the sequence for the rows that look at them.
    Bars: |err| <= atol + rtol |ref| with (fp16) 3e-3 / 2e-3 and (bf16) 2e-2 / 1.6e-2: P is rounded to the element type
    before P.V (as the reference's fp16 SDPA does) and phase 0 is rounded once more before the add."""
    q, k, vt, kr, vr, s2 = attn_operands(ops, dt, N=N, M=M, seed=11)
    D = 40
    q.mul_(3.0)
    if spike:
        g = torch.Generator().manual_seed(5)
        qrows = torch.randint(0, N, (40,), generator=g)
        # key N-130 is aligned with a set of query rows (score ~ +40 in base-2 units on those rows), garment key M-70 likewise
        k[:, :, N - 130, :D] = (q[:, :, qrows[0], :D].float() * 2.0).to(dt)
        kr[:, :, M - 70, :D] = (q[:1, :, qrows[1], :D].float() * 2.5).to(dt)
        for r in qrows[2:].tolist():
            q[:, :, r, :D] = q[:, :, qrows[0], :D] * (0.5 + (r % 7) * 0.25)
    ref = attn_oracle(q, k, vt, kr, vr, s2, D, dt)
    atol, rtol = (3e-3, 2e-3) if dt == torch.float16 else (2e-2, 1.6e-2)
    assert ref.abs().max() > 1.0
    for pad_one in (True, False):           # both staging paths of the kernel: LDS-DMA (what the processors run) and registers
        out = run_attn(ops, q, k, vt, kr, vr, s2, k_pad_one=pad_one).float().cpu()
        err = (out - ref).abs()
        bad = err > atol + rtol * ref.abs()
        assert torch.isfinite(out).all()
        assert not bad.any(), f"k_pad_one={pad_one}: {int(bad.sum())} of {bad.numel()} off; max err {err.max().item():.4g}; ref max {ref.abs().max().item():.3g}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("N,M", [(4096, 4096), (1000, 700), (640, 5120)])
def test_full_attention_duplicated_first_phase_is_bit_identical(ops, dt, N, M):
    """imd_attn_params.out_dup (ABI v9): a launch over the B cond rows of a CFG batch's first hybrid block that also stores each row's
    first-phase result == the 2B-row launch in which rows [B, 2B) repeat the same Q / K / V without the garment branch
    (RefSAttnProcessor2_0 with / without sa_hidden_states on identical hidden states, attention_processor.py:589-612).  Bit for bit,
    both element types, ragged N, a row whose garment weight is 0 (single-phase row: both outputs equal), both default variants."""
    B, H, D = 4, 8, 40
    q, k, vt, kr, vr, _ = attn_operands(ops, dt, B=B, N=N, M=M, seed=5)
    s2h = torch.tensor([1.0, 0.5, 0.0, 2.0], device="cuda")                  # (row 2: garment branch off -> one phase)
    s2 = torch.cat([s2h, torch.zeros(B, device="cuda")])
    ref = run_attn(ops, torch.cat([q, q]), torch.cat([k, k]), torch.cat([vt, vt]), kr, vr, s2)
    for variant in (13, 12):
        with ops.tuning_scope(attn_variant=variant):
            out = torch.empty(2 * B, N, H * D, dtype=dt, device="cuda")
            ops.attention(q, k, vt, out[:B], B=B, H=H, N=N, D=D, L1=N, L1P=ops.pad64(N), k2=kr, v2t=vr, scale2=s2, L2=M, L2P=ops.pad64(M),
                          kv2_bdiv=B, k_pad_one=True, out_dup=out[B:])
            ref_v = run_attn(ops, torch.cat([q, q]), torch.cat([k, k]), torch.cat([vt, vt]), kr, vr, s2)
        assert torch.equal(out, ref_v), f"variant {variant}"
        assert torch.equal(ref_v, ref) or variant == 12                          # (12 / 13 differ only under overflow: not here)
    assert torch.equal(out[2], out[B + 2]) and not torch.equal(out[0], out[B])
    assert ops.attention_dup_supported(H, N, D) and not ops.attention_dup_supported(H, 256, D) and not ops.attention_dup_supported(H, N, 80)
    with pytest.raises(ops.L.ImdError):                                          # register staging (no k_pad_one) has no duplicated store
        ops.attention(q, k, vt, out[:B], B=B, H=H, N=N, D=D, L1=N, L1P=ops.pad64(N), k_pad_one=False, out_dup=out[B:])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_full_attention_softmax_rows_sum_to_one(ops, dt):
    """V = 1 everywhere -> softmax(QK^T) V = 1 exactly (numerator and denominator come from the same rounded P), so the
    hybrid output is exactly 1 + s2[b]: 2 on the cond rows (garment branch on), 1 on the uncond rows."""
    q, k, vt, kr, vr, s2 = attn_operands(ops, dt)
    D = 40
    vt[:, :, :D, :4096] = 1.0
    vr[:, :, :D, :4096] = 1.0
    out = run_attn(ops, q, k, vt, kr, vr, s2).float()
    assert torch.equal(out[:4], torch.full_like(out[:4], 2.0))
    assert torch.equal(out[4:], torch.full_like(out[4:], 1.0))


def test_full_attention_key_order_invariance(ops):
    """Attention does not depend on the order of the keys: permuting the 4096 image tokens and the 4096 garment tokens
    (K rows and V^T columns together) changes the result only through summation order and the deferred-max path.
    Bar: the north-star atol 1e-2 (outputs are O(0.05): sums of 4096 N(0,1) values weighted by a near-flat softmax)."""
    dt = torch.bfloat16
    q, k, vt, kr, vr, s2 = attn_operands(ops, dt, seed=3)
    base = run_attn(ops, q, k, vt, kr, vr, s2).float()
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(5)).cuda()
    k2 = k[:, :, perm].contiguous(); vt2 = vt.clone(); vt2[..., :4096] = vt[..., perm]
    kr2 = kr[:, :, perm].contiguous(); vr2 = vr.clone(); vr2[..., :4096] = vr[..., perm]
    got = run_attn(ops, q, k2, vt2, kr2, vr2, s2).float()
    assert base.abs().max() > 0.05
    assert (got - base).abs().max() < 1e-2, (got - base).abs().max().item()


@pytest.mark.parametrize("cfg", [5, 4, 0], ids=["halo-patch", "gather128x128x32", "gather128x128x64"])
@pytest.mark.parametrize("Cin,Cout", [(320, 320), (960, 320)])
def test_full_conv_power_of_two_linearity(ops, cfg, Cin, Cout):
    """conv(2x) == 2 conv(x) and conv(x/4) == conv(x)/4 BIT FOR BIT (scaling by a power of two commutes with every
    rounding on the path) on the level-0 ResNet convolutions of the CFG batch (8 x 64 x 64 pixels)."""
    x = rnd(1, 8, 64, 64, Cin, device="cuda").to(bf16)
    w = rnd(2, Cout, 9 * Cin, scale=(9 * Cin) ** -0.5, device="cuda").to(bf16)
    y = ops.conv2d_nhwc(x, w, None, cfg=cfg, split_k=1)
    y2 = ops.conv2d_nhwc((x.float() * 2).to(bf16), w, None, cfg=cfg, split_k=1)
    yq = ops.conv2d_nhwc((x.float() * 0.25).to(bf16), w, None, cfg=cfg, split_k=1)
    assert torch.isfinite(y).all() and y.float().abs().max() > 1.0
    assert torch.equal(y2.float(), y.float() * 2)
    assert torch.equal(yq.float(), y.float() * 0.25)


def test_full_conv_kernels_agree(ops):
    """The halo-patch kernel and the gather kernel are two schedules of the same sum: same bf16 output up to fp32
    summation order (bar: 1 bf16 ulp of the output scale), and both match F.conv2d on a sampled set of pixels."""
    Cin, Cout = 640, 320
    x = rnd(3, 8, 64, 64, Cin, device="cuda").to(bf16)
    w = rnd(4, Cout, 9 * Cin, scale=(9 * Cin) ** -0.5, device="cuda").to(bf16)
    b = rnd(5, Cout, device="cuda")
    ya = ops.conv2d_nhwc(x, w, b, cfg=5, split_k=1).float()
    yb = ops.conv2d_nhwc(x, w, b, cfg=0, split_k=1).float()
    assert (ya - yb).abs().max() <= 2 ** -7 * max(1.0, ya.abs().max().item()) * 1.01
    ref = F.conv2d(x[:1].float().permute(0, 3, 1, 2), w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    assert (ya[:1] - ref).abs().max() < 2e-2 + 1e-2 * ref.abs().max()


@pytest.mark.parametrize("M,N,K", [(32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280), (32768, 320, 1280)])
def test_full_linear_tile_configs_agree(ops, M, N, K):
    """Every tile configuration (and the tuned default) of the projection GEMMs of one transformer block computes the
    same matrix: fp32 accumulation order is the only difference (bar: 1 bf16 ulp of the output scale)."""
    x = rnd(6, M, K, device="cuda").to(bf16)
    w = rnd(7, N, K, scale=K ** -0.5, device="cuda").to(bf16)
    b = rnd(8, N, device="cuda")
    base = ops.linear(x, w, b, cfg=0, split_k=1).float()
    ref = (x[:256].float() @ w.float().t() + b)
    assert (base[:256] - ref).abs().max() < 1e-2 + 1e-2 * ref.abs().max()
    tol = 2 ** -7 * max(1.0, base.abs().max().item()) * 1.01
    for cfg in (1, 2, 3, 4, 6, 7, 8, 9, 10, 11, -1):
        got = ops.linear(x, w, b, cfg=cfg, split_k=(0 if cfg == -1 else 1)).float()
        assert (got - base).abs().max() <= tol, (cfg, (got - base).abs().max().item())


@pytest.mark.parametrize("Cc,HW", [(320, 4096), (960, 4096), (640, 1024), (1280, 64)])
def test_full_groupnorm_unit_statistics(ops, Cc, HW):
    """After GroupNorm(32) with gamma = 1, beta = 0 every (batch, group) slab has mean 0 and variance 1
    (tolerance 2e-2: bf16 output rounding of O(1) values + eps)."""
    x = (rnd(9, 8, HW, Cc, device="cuda") * 3.0 + 1.5).to(bf16)
    y = ops.group_norm(x, torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), groups=32, eps=1e-5).float()
    g = y.view(8, HW, 32, Cc // 32)
    mean = g.mean(dim=(1, 3)); var = g.var(dim=(1, 3), unbiased=False)
    assert mean.abs().max() < 2e-2 and (var - 1).abs().max() < 2e-2, (mean.abs().max().item(), (var - 1).abs().max().item())


def test_full_ddim_cfg_identities(ops):
    """DDIM (eta = 0) + CFG on the [4, 4096, 4] fp32 latent of the BASELINE batch:
    eps = 0 -> z' = sqrt(a_prev / a_t) z;  eps_c == eps_u -> guidance drops out;  the emitted 16-bit UNet input holds
    the same z' in both CFG halves with channels 4..7 zero."""
    from imagdressing_amd.scheduler import DDIMScheduler
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    sch.set_timesteps(50)
    ac = sch.alphas_cumprod
    t = int(sch.timesteps[10]); tp = t - 20
    a_t, a_prev = float(ac[t]), float(ac[tp])
    B, HW = 4, 4096
    z0 = rnd(10, B, HW, 4, device="cuda")
    # (1) eps = 0
    z = z0.clone(); eps = torch.zeros(2 * B, HW, 4, device="cuda")
    xn = torch.empty(2 * B, HW, 8, dtype=bf16, device="cuda")
    ops.ddim_cfg_step(z, eps, xn, guidance=7.5, a_t=a_t, a_prev=a_prev)
    assert torch.allclose(z, z0 * math.sqrt(a_prev / a_t), rtol=2e-6, atol=1e-7)
    assert torch.equal(xn[:B, :, :4], z.to(bf16)) and torch.equal(xn[B:, :, :4], z.to(bf16)) and (xn[..., 4:] == 0).all()
    # (2) equal cond / uncond predictions: any guidance scale gives the same step
    e = rnd(11, B, HW, 4, device="cuda")
    za, zb = z0.clone(), z0.clone()
    ops.ddim_cfg_step(za, torch.cat([e, e]), None, guidance=7.5, a_t=a_t, a_prev=a_prev)
    ops.ddim_cfg_step(zb, torch.cat([e, e]), None, guidance=1.0, a_t=a_t, a_prev=a_prev)
    assert torch.allclose(za, zb, rtol=1e-5, atol=1e-6)
    x0 = (z0 - math.sqrt(1 - a_t) * e) / math.sqrt(a_t)
    assert torch.allclose(za, math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * e, rtol=1e-4, atol=1e-5)
