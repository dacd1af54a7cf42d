"""End-to-end parity of the HIP engine against the CPU oracle (reference loop semantics,
oracle/pipeline.py) on seeded synthetic weights.  Tolerances are written per test; the north-star
bar is fp16-class atol 1e-2 on O(1) outputs."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.harness import SMALL, build_pair, err_stats  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def record(name, stats):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_stats.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **stats)) + "\n")


# per-dtype bars.  fp16 (the reference's own compute dtype) is held to the north-star atol 1e-2 on one UNet
# forward; bf16 (8 mantissa bits) carries an inherent ~1.2 % rms error through ~60 layers (measured identically
# by rounding the fp32 CPU oracle's GEMM inputs to bf16), so its bar is 5e-2 abs / 2e-2 rms.
BARS = {torch.float16: dict(max_abs=1e-2, rel_rms=4e-3), torch.bfloat16: dict(max_abs=5e-2, rel_rms=2e-2)}


@pytest.fixture(scope="module", params=[torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def small_pair(request):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    torch.manual_seed(0)
    p = build_pair(SMALL, seed=0, dtype=request.param)
    p["dtype"] = request.param
    return p


def g(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@torch.no_grad()
def test_unet_forward_small(small_pair):
    p = small_pair
    x = g(1, 2, 4, 16, 16); ehs = g(2, 2, 77, 64, scale=0.5)
    ref = p["o_unet"](x, 481, ehs)
    got = p["e_unet"](x.cuda(), 481, ehs.cuda())[0]
    st = err_stats(got, ref); record(f"unet_forward_small[{p['dtype']}]", st)
    bar = BARS[p["dtype"]]
    assert st["max_abs"] < bar["max_abs"] and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_unet_forward_small_fused_feed_forward(small_pair, monkeypatch):
    """the 320-channel blocks with norm3 -> GEGLU feed-forward -> + residual as ONE launch (ff_fused.hip; the engines use it from
    16384 rows up, here forced on) against the oracle, and against the four-launch path"""
    from imagdressing_amd import ops
    p = small_pair
    x = g(1, 2, 4, 16, 16); ehs = g(2, 2, 77, 64, scale=0.5)
    ref = p["o_unet"](x, 481, ehs)
    monkeypatch.setattr(ops, "FUSED_FF_MIN_ROWS", 0)
    got = p["e_unet"](x.cuda(), 481, ehs.cuda())[0]
    monkeypatch.setattr(ops, "FUSED_FF", False)
    plain = p["e_unet"](x.cuda(), 481, ehs.cuda())[0]
    assert not torch.equal(got, plain)                   # the fused kernel really ran
    st = err_stats(got, ref); record(f"unet_forward_small_fused_ff[{p['dtype']}]", st)
    bar = BARS[p["dtype"]]
    assert st["max_abs"] < bar["max_abs"] and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_unet_forward_small_with_garment(small_pair):
    from oracle.pipeline import garment_features
    p = small_pair
    x = g(3, 1, 4, 16, 16); ehs = g(4, 1, 77, 64, scale=0.5)
    refl = g(5, 1, 4, 16, 16); cloth = g(6, 2, 16, 64, scale=0.5)
    sa_o = garment_features(p["o_ref"], refl, cloth)
    ref = p["o_unet"](x, 301, ehs, cross_attention_kwargs={"sa_hidden_states": sa_o})
    ref_plain = p["o_unet"](x, 301, ehs)
    assert (ref - ref_plain).abs().max() > 1e-2          # the garment branch matters in this setup
    # engine: garment features from the engine's own garment UNet
    from imagdressing_amd.unet import nchw_to_nhwc8
    p["e_ref"].forward_nhwc(nchw_to_nhwc8(refl.cuda(), p["dtype"]), 0, cloth[1:2].cuda().to(p["dtype"]).contiguous())
    sa_e = {n: pr.cache["hidden_states"] for n, pr in p["e_ref"].attn_processors.items()}
    n0 = [n for n in p["names"] if n.endswith("attn1.processor")][0]
    st0 = err_stats(sa_e[n0], sa_o[n0]); record(f"garment_feature_first_layer[{p['dtype']}]", st0)
    got = p["e_unet"](x.cuda(), 301, ehs.cuda(), cross_attention_kwargs={"sa_hidden_states": sa_e})[0]
    st = err_stats(got, ref); record(f"unet_forward_small_garment[{p['dtype']}]", st)
    bar = BARS[p["dtype"]]
    assert st["max_abs"] < bar["max_abs"] and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_time_embeddings_of_a_whole_schedule_at_once(small_pair):
    """The pipelines run the time-embedding chain once over all timesteps of a call and every forward picks its row, handed to the resnets as ONE
    vector for the whole batch (rowvec_stride 0): the rows equal the per-forward chain's (same GEMMs on more rows: fp32 summation order only), a
    forward with the table equals a forward without it, and the table is gone after clear_time_embeddings()."""
    p = small_pair
    unet = p["e_unet"]
    ts = [981, 721, 481, 1]
    table = unet.precompute_time_embeddings(ts, "cuda")
    try:
        assert table.shape[0] == len(ts) and table.dtype == torch.float32
        for i, t in enumerate(ts):
            row = unet._time_embed_rows(torch.full((2,), float(t), dtype=torch.float32, device="cuda"))
            assert torch.equal(row[0], row[1])
            st = err_stats(table[i:i + 1], row[:1])
            assert st["rel_rms"] < 2e-3 and st["max_abs"] < 2e-2 * max(st["ref_std"], 1e-3), (t, st)
            got = unet._time_embed(t, 2, "cuda")
            assert got.shape[0] == 1 and got.data_ptr() == table[i:i + 1].data_ptr()
        x = g(1, 2, 4, 16, 16).cuda(); ehs = g(2, 2, 77, 64, scale=0.5).cuda()
        with_table = unet(x, 481, ehs)[0]
    finally:
        unet.clear_time_embeddings()
    assert unet._time_embed(481, 2, "cuda").shape[0] == 2
    without = unet(x, 481, ehs)[0]
    st = err_stats(with_table, without); record(f"unet_forward_small_schedule_time_embeddings[{p['dtype']}]", st)
    assert st["rel_rms"] < 2e-3 and st["max_abs"] < 1e-2, st
    ref = p["o_unet"](x.cpu(), 481, ehs.cpu())
    st = err_stats(with_table, ref)
    bar = BARS[p["dtype"]]
    assert st["max_abs"] < bar["max_abs"] and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_pipeline_small_20_steps(small_pair):
    """B=2 images sharing a garment == 2 independent runs of the reference loop (oracle)."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    p = small_pair
    steps, gs = 20, 7.5
    lat = torch.stack([torch.randn(4, 16, 16, generator=torch.Generator().manual_seed(42 + i)) for i in range(2)])
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    refs = []
    for i in range(2):
        tr = []
        refs.append(denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat[i:i + 1], pe, ne, cloth, refl, steps, gs, trace=tr))
    ref = torch.cat(refs)
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)

    class Proj:   # ImgProj stand-in: the tokens are given
        def __call__(self, h):
            return h
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=Proj(), scheduler=sch, safety_checker=None, feature_extractor=None)
    out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128,
               num_inference_steps=steps, guidance_scale=gs, num_images_per_prompt=2, prompt_embeds=pe.cuda(),
               negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(), ref_image_latents=refl.cuda(),
               latents=lat.cuda(), output_type="latent").images
    st = err_stats(out, ref); record(f"pipeline_small_20_steps[{p['dtype']}]", st)
    assert torch.isfinite(out).all()
    # Seeded random weights are not a trained denoiser: x0 = (z - sqrt(1-a)eps)/sqrt(a) does not cancel, so the
    # latent grows ~14x over the trajectory (ref_std ~ 17).  The bar is therefore RELATIVE to the oracle's final
    # latent scale: atol 1e-2 x ref_std for fp16 (1e-1 for bf16: the largest of 2048 values at the end of a 20-step
    # trajectory moves between 7e-2 and 9e-2 with the summation order of any one kernel) and rms 4e-3 (2.5e-2).
    scale = st["ref_std"]
    bar = dict(max_abs=1e-2, rel_rms=4e-3) if p["dtype"] == torch.float16 else dict(max_abs=1e-1, rel_rms=2.5e-2)
    assert st["max_abs"] < bar["max_abs"] * scale and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_pipeline_small_teacher_forced_eps_absolute(small_pair):
    """Per-step error in ABSOLUTE terms, without the compounding of a non-contractive synthetic model: the engine's CFG-batched
    UNet call is fed the ORACLE's latent z_i of a 20-step trajectory (teacher forcing) at steps 0, 1, 5, 10, 15, 19 and its
    eps_c / eps_u are compared with the oracle's at the same z_i.  Seeded weights are not a denoiser (z grows ~14x =
    sqrt(abar_end / abar_start), for ANY eps uncorrelated with z), so late inputs are far outside the O(1) regime a trained model
    keeps its latents in; the bar is the north-star atol 1e-2 for fp16 wherever |z| is O(1) (step 0, 1) and 1e-2 x the eps scale
    relative to step 0 elsewhere; bf16: 6e-2 (8 mantissa bits)."""
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise, garment_features
    p = small_pair
    steps, gs = 20, 7.5
    lat = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(42))
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    tr = []
    sch = DDIMOracle()
    denoise(p["o_unet"], p["o_ref"], sch, lat, pe, ne, cloth, refl, steps, gs, trace=tr)
    ts = sch.set_timesteps(steps)
    zs = [lat] + tr[:-1]                                   # latent going INTO step i
    sa_o = garment_features(p["o_ref"], refl, cloth)
    from imagdressing_amd.unet import nchw_to_nhwc8
    e_ref = p["e_ref"]
    e_ref.forward_nhwc(nchw_to_nhwc8(refl.cuda(), p["dtype"]), 0, cloth[1:2].cuda().to(p["dtype"]).contiguous())
    sa_e = {n: pr.cache["hidden_states"] for n, pr in e_ref.attn_processors.items()}
    mask = torch.tensor([1.0, 0.0], device="cuda")
    ehs2 = torch.cat([pe, ne]).cuda()
    base = None
    for i in (0, 1, 5, 10, 15, 19):
        z, t = zs[i], int(ts[i])
        ec = p["o_unet"](z, t, pe, cross_attention_kwargs={"sa_hidden_states": sa_o})
        eu = p["o_unet"](z, t, ne)
        got = p["e_unet"](torch.cat([z, z]).cuda(), t, ehs2, cross_attention_kwargs={"sa_hidden_states": sa_e, "sa_batch_mask": mask})[0]
        ref = torch.cat([ec, eu])
        st = err_stats(got, ref)
        st.update(step=i, z_std=z.std().item())
        record(f"teacher_forced_eps[{p['dtype']}]", st)
        base = base or st["ref_std"]
        bar = (1e-2 if p["dtype"] == torch.float16 else 6e-2) * max(1.0, st["ref_std"] / base)
        assert st["max_abs"] <= bar, st


def _sched():
    from imagdressing_amd.scheduler import DDIMScheduler
    return DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         clip_sample=False, set_alpha_to_one=False, steps_offset=1)


@torch.no_grad()
def test_step_graph_replay_is_bit_identical(small_pair):
    """``enable_step_graph``: step 1 of the DDIM loop captured once as a HIP graph (timestep + schedule coefficients read from
    device memory, imd_ddim_params.coefs) and replayed for steps 1..S-1 gives EXACTLY the eager loop's latents (same kernels, same
    order), at batch 1 -- the reference's literal usage (IMAGDressing_v1_pipeline.py:389) -- and batch 2, twice in a row."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    p = small_pair
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=lambda h: h, scheduler=_sched(), safety_checker=None, feature_extractor=None)
    for nimg in (1, 2):
        lat = torch.stack([torch.randn(4, 16, 16, generator=torch.Generator().manual_seed(42 + i)) for i in range(nimg)])

        def run():
            return pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128,
                        num_inference_steps=8, guidance_scale=7.5, num_images_per_prompt=nimg, prompt_embeds=pe.cuda(),
                        negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(), ref_image_latents=refl.cuda(),
                        latents=lat.cuda(), output_type="latent").images
        pipe.enable_step_graph(False)
        eager = run()
        pipe.enable_step_graph(True)
        g1 = run()
        g2 = run()
        assert getattr(pipe, "_last_step_graph", None) is not None      # the graph path really ran
        side = pipe._graph_stream.cuda_stream
        from imagdressing_amd import ops as _ops
        assert any(k[-1] == side for k in _ops._ws), "the replayed step keeps its scratch buffers keyed by the side stream"
        pipe.enable_step_graph(False)                                   # releases the graph, its stream and that stream's scratch buffers
        assert getattr(pipe, "_last_step_graph", None) is None and not any(k[-1] == side for k in _ops._ws)
        assert torch.isfinite(eager).all()
        assert torch.equal(eager, g1) and torch.equal(eager, g2), (nimg, (eager - g1).abs().max().item())


def _traj_bar(dtype):
    """Bars for multi-step latent trajectories, RELATIVE to the oracle's final latent scale (seeded random weights
    are not a trained denoiser: the latent grows ~6-17x, so an absolute 1e-2 is meaningless there).  fp16: rms error
    < 0.4 % and worst element < 2 % of the latent std (1e-2 x std is met on the plain / IPA paths; the inpaint blend
    concentrates the error in the masked region whose own std is ~2x the global one).  bf16: 2.5 % / 15 %."""
    return dict(max_abs=2e-2, rel_rms=4e-3) if dtype == torch.float16 else dict(max_abs=0.15, rel_rms=2.5e-2)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_pipeline_ipa_controlnet_small(dtype):
    """config-3 path: LoraRefS + LoRAIP processors (LoRA folded into weights), 4 face tokens appended to the text
    tokens, pose ControlNet residuals (cond / uncond halves) -- vs the oracle's reference-loop semantics."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_ipa_controlnet import IMAGDressing_v1
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    p = build_pair(SMALL, seed=3, kind="ipa", with_controlnet=True, dtype=dtype)
    steps, gs = 8, 7.0
    lat = g(42, 1, 4, 16, 16)
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    face_p, face_n = g(14, 1, 4, 64, scale=0.5), g(15, 1, 4, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    pose = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(16))
    # oracle: prompt embeds with face tokens appended (:555-557); ControlNet gets text-only [neg, pos] (:550)
    ref = denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat, torch.cat([pe, face_p], 1), torch.cat([ne, face_n], 1), cloth, refl,
                  steps, gs, controlnet=p["o_ctrl"], control_image=pose, prompt_embeds_control=torch.cat([ne, pe]),
                  conditioning_scale=0.8)

    class FaceProj:      # image_proj_model stand-in returning the given face tokens
        def __call__(self, idv, clip):
            return (face_p if float(idv.abs().sum()) > 0 else face_n).cuda()
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           controlnet=p["e_ctrl"], image_encoder=None, ImgProj=lambda h: h, ip_ckpt=None, scheduler=_sched())
    pipe.image_proj_model = FaceProj()
    out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128,
               num_inference_steps=steps, guidance_scale=gs, pose_image=pose.cuda(), faceid_embeds=torch.ones(1, 512),
               face_clip_hidden_states=torch.zeros(1, 257, 1280), face_uncond_clip_hidden_states=torch.zeros(1, 257, 1280),
               image_scale=1.0, ipa_scale=0.9, s_lora_scale=0.2, c_lora_scale=0.2, controlnet_conditioning_scale=0.8,
               prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(),
               ref_image_latents=refl.cuda(), latents=lat.cuda(), output_type="latent").images
    st = err_stats(out, ref); record(f"pipeline_ipa_controlnet_small[{dtype}]", st)
    bar = _traj_bar(dtype)
    assert torch.isfinite(out).all()
    assert st["max_abs"] < bar["max_abs"] * max(st["ref_std"], 1.0) and st["rel_rms"] < bar["rel_rms"], st


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_pipeline_inpaint_small(dtype):
    """config-5 path: ControlNet-inpaint + per-step masked blend with re-noised original latents."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    p = build_pair(SMALL, seed=5, with_controlnet=True, dtype=dtype)
    steps, gs = 8, 5.0
    noise = g(42, 2, 4, 16, 24)
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    img_lat = g(17, 1, 4, 16, 24)
    mask = torch.zeros(1, 1, 16, 24); mask[:, :, 4:12, 6:18] = 1.0              # centred rectangle
    ctrl = torch.rand(1, 3, 128, 192, generator=torch.Generator().manual_seed(18))
    refs = [denoise(p["o_unet"], p["o_ref"], DDIMOracle(), noise[i:i + 1], pe, ne, cloth, refl, steps, gs, controlnet=p["o_ctrl"],
                    control_image=ctrl, prompt_embeds_control=torch.cat([ne, pe]), conditioning_scale=1.0,
                    inpaint=dict(mask=mask, image_latents=img_lat, noise=noise[i:i + 1])) for i in range(2)]
    ref = torch.cat(refs)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           controlnet=p["e_ctrl"], image_encoder=None, ImgProj=lambda h: h, scheduler=_sched())
    out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=192, height=128,
               num_inference_steps=steps, guidance_scale=gs, num_images_per_prompt=2, control_image=ctrl.cuda(),
               prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(),
               ref_image_latents=refl.cuda(), image_latents=img_lat.cuda(), mask_latents=mask.cuda(), noise=noise.cuda(),
               output_type="latent").images
    st = err_stats(out, ref); record(f"pipeline_inpaint_small[{dtype}]", st)
    bar = _traj_bar(dtype)
    assert torch.isfinite(out).all()
    assert st["max_abs"] < bar["max_abs"] * max(st["ref_std"], 1.0) and st["rel_rms"] < bar["rel_rms"], st
    # outside the mask the result is exactly the original latents (last step: no re-noising, :494-500)
    keep = (mask == 0).expand(2, 4, -1, -1)
    assert torch.allclose(out.cpu()[keep], img_lat.expand(2, -1, -1, -1)[keep], atol=1e-5)


@torch.no_grad()
def test_pipeline_small_stochastic_ddim(small_pair):
    """eta = 0.6 through the pipeline call (eta -> prepare_extra_step_kwargs -> DDIMScheduler.step, IMAGDressing_v1_pipeline.py:451,:530)
    with the per-step noise passed in == the oracle loop with the same noise; and a generator-driven run is reproducible."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    p = small_pair
    steps, gs, eta = 10, 7.5, 0.6
    lat = g(42, 1, 4, 16, 16)
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    vn = [g(100 + i, 1, 4, 16, 16) for i in range(steps)]
    ref = denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat, pe, ne, cloth, refl, steps, gs, eta=eta, variance_noise=vn)
    ref0 = denoise(p["o_unet"], p["o_ref"], DDIMOracle(), lat, pe, ne, cloth, refl, steps, gs)
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=lambda h: h, scheduler=_sched())
    kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128, num_inference_steps=steps,
              guidance_scale=gs, prompt_embeds=pe.cuda(), negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(),
              ref_image_latents=refl.cuda(), latents=lat.cuda(), output_type="latent")
    out = pipe(eta=eta, variance_noise=[v.cuda() for v in vn], **kw).images
    st = err_stats(out, ref); record(f"pipeline_small_stochastic_ddim[{p['dtype']}]", st)
    bar = _traj_bar(p["dtype"])
    assert torch.isfinite(out).all()
    assert st["max_abs"] < bar["max_abs"] * max(st["ref_std"], 1.0) and st["rel_rms"] < bar["rel_rms"], st
    assert err_stats(out, ref0)["rel_rms"] > 10 * st["rel_rms"]            # it is not the deterministic trajectory
    a = pipe(eta=eta, generator=torch.Generator("cuda").manual_seed(5), **kw).images
    b = pipe(eta=eta, generator=torch.Generator("cuda").manual_seed(5), **kw).images
    c = pipe(eta=eta, generator=torch.Generator("cuda").manual_seed(6), **kw).images
    assert torch.equal(a, b) and not torch.equal(a, c)
    with pytest.raises(NotImplementedError):
        pipe(**dict(kw, guidance_scale=1.0))                               # the reference's loop cannot run without the CFG pair either


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_pipeline_inpaint_small_strength(dtype):
    """strength = 0.6 (..._controlnet_inpainting.py:316-341 -> diffusers get_timesteps / prepare_latents): the last int(10 * 0.6) = 6
    timesteps, from the image latents noised to the first of them; ControlNet gate over the 6 steps that run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1
    from oracle.ddim import DDIMOracle
    from oracle.pipeline import denoise
    p = build_pair(SMALL, seed=5, with_controlnet=True, dtype=dtype)
    steps, gs, strength = 10, 5.0, 0.6
    noise = g(42, 1, 4, 16, 24)
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    img_lat = g(17, 1, 4, 16, 24)
    mask = torch.zeros(1, 1, 16, 24); mask[:, :, 4:12, 6:18] = 1.0
    ctrl = torch.rand(1, 3, 128, 192, generator=torch.Generator().manual_seed(18))
    tr = []
    ref = denoise(p["o_unet"], p["o_ref"], DDIMOracle(), None, pe, ne, cloth, refl, steps, gs, controlnet=p["o_ctrl"],
                  control_image=ctrl, prompt_embeds_control=torch.cat([ne, pe]), conditioning_scale=1.0,
                  inpaint=dict(mask=mask, image_latents=img_lat, noise=noise), strength=strength, trace=tr)
    assert len(tr) == 6
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           controlnet=p["e_ctrl"], image_encoder=None, ImgProj=lambda h: h, scheduler=_sched())
    kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=192, height=128,
              num_inference_steps=steps, guidance_scale=gs, control_image=ctrl.cuda(), prompt_embeds=pe.cuda(),
              negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(), ref_image_latents=refl.cuda(),
              image_latents=img_lat.cuda(), mask_latents=mask.cuda(), noise=noise.cuda(), output_type="latent")
    mine = []
    out = pipe(strength=strength, trace=mine, **kw).images
    assert len(mine) == 6
    st = err_stats(out, ref); record(f"pipeline_inpaint_small_strength[{dtype}]", st)
    bar = _traj_bar(dtype)
    assert torch.isfinite(out).all()
    assert st["max_abs"] < bar["max_abs"] * max(st["ref_std"], 1.0) and st["rel_rms"] < bar["rel_rms"], st
    keep = (mask == 0).expand(1, 4, -1, -1)
    assert torch.allclose(out.cpu()[keep], img_lat[keep], atol=1e-5)
    for bad in (0.0, 1.5, 0.05):                       # diffusers: strength outside (0, 1]; fewer than one step left
        with pytest.raises(ValueError):
            pipe(strength=bad, **kw)


@torch.no_grad()
def test_build_engines_from_imagdressing_checkpoint(tmp_path):
    """imagdressing_amd.checkpoint.build_engines (the reference's prepare(), inference_IMAGdressing.py:40-135) from a
    DeepSpeed-style checkpoint FILE == engines built directly from the same tensors: bit-identical UNet outputs."""
    from imagdressing_amd import checkpoint as CK
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter.resampler import Resampler
    dt = torch.float16
    p = build_pair(SMALL, seed=0, dtype=dt)
    full = dict(E.SD15_CONFIG, **SMALL)
    sd_u = E.random_state_dict(E.unet_param_shapes(full), 0)
    sd_r = E.random_state_dict(E.unet_param_shapes(full), 1)
    torch.manual_seed(3)
    rk = dict(dim=64, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=64, ff_mult=2)
    proj = Resampler(**rk)
    adapters = torch.nn.ModuleList(p["e_unet"].attn_processors.values())
    ck = {}
    ck.update({"ref_unet." + k: v for k, v in sd_r.items()})
    ck.update({"unet." + k: v for k, v in sd_u.items()})
    ck.update({"proj." + k: v.detach().cpu() for k, v in proj.state_dict().items()})
    ck.update({"adapter_modules." + k: v.detach().float().cpu() for k, v in adapters.state_dict().items()})
    f = tmp_path / "IMAGDressing-v1_small.pt"
    torch.save({"module": ck}, f)
    eng = CK.build_engines(sd_u, CK.load_state_dict_file(str(f)), device="cuda", dtype=dt, config=SMALL, resampler_kwargs=rk)
    assert eng["other_keys"] == [] and eng["unused_unet_keys"] == len(sd_u)
    x = g(1, 2, 4, 16, 16).cuda(); ehs = g(2, 2, 77, 64, scale=0.5).cuda()
    refl = g(5, 1, 4, 16, 16); cloth = g(6, 1, 16, 64, scale=0.5)
    from imagdressing_amd.unet import nchw_to_nhwc8
    outs = []
    for unet, ref in ((p["e_unet"], p["e_ref"]), (eng["unet"], eng["ref_unet"])):
        ref.forward_nhwc(nchw_to_nhwc8(refl.cuda(), dt), 0, cloth.cuda().to(dt).contiguous())
        sa = {n: pr.cache["hidden_states"] for n, pr in ref.attn_processors.items()}
        outs.append(unet(x, 301, ehs, cross_attention_kwargs={"sa_hidden_states": sa})[0])
    assert torch.equal(outs[0], outs[1])
    clip = g(7, 1, 20, 96, scale=0.5).cuda().to(dt)
    assert torch.equal(eng["image_proj"](clip), proj.to(device="cuda", dtype=dt)(clip))


def _write_hf_dir(d, sd, cfg):
    import json
    import os

    from safetensors.torch import save_file
    os.makedirs(d, exist_ok=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, f)


@torch.no_grad()
def test_prepare_flow_of_the_reference_script(tmp_path):
    """The BODY of the reference's ``prepare()`` (/root/reference/inference_IMAGdressing.py:40-135), statement for
    statement, against the MI355X classes: ``from_pretrained(...).to(dtype=, device=)`` from local Hugging Face-layout
    directories, the ``attn_procs`` loop, ``unet.set_attn_processor``, ``ModuleList(unet.attn_processors.values())
    .to(...)``, ``torch.load(ckpt)["module"]`` and its prefix split, the three ``load_state_dict`` calls, the scheduler and
    the pipeline constructor (safety checker / feature extractor passed as CLASSES, :133-134).  Only the import lines
    differ from the script.  The pipeline it returns generates the same latents as one assembled directly from the same
    tensors (bit-identical), and ``set_scale`` reaches the loaded processors."""
    # --- the imports the script would change ---
    from adapter.attention_processor import CacheAttnProcessor2_0, CAttnProcessor2_0, RefSAttnProcessor2_0
    from adapter.resampler import Resampler
    from dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd import unet as E
    from imagdressing_amd.scheduler import DDIMScheduler
    from imagdressing_amd.unet import UNet2DConditionModel
    from imagdressing_amd.vae import AutoencoderKL

    class StableDiffusionSafetyChecker:      # the script passes these two CLASSES, never instances
        pass

    class CLIPImageProcessor:
        pass
    # --- synthetic "downloads": an SD-layout UNet dir, a VAE dir, and a DeepSpeed-wrapped IMAGDressing checkpoint ---
    full = dict(E.SD15_CONFIG, **SMALL)
    sd_u = E.random_state_dict(E.unet_param_shapes(full), 0)
    sd_r = E.random_state_dict(E.unet_param_shapes(full), 1)
    _write_hf_dir(tmp_path / "rv" / "unet", sd_u, {k: full[k] for k in ("block_out_channels", "attention_head_dim", "norm_num_groups",
                                                                        "cross_attention_dim", "in_channels", "out_channels")})
    from oracle import vae as OV
    vcfg = dict(block_out_channels=(64, 128, 128, 128), norm_num_groups=8)
    sd_v = OV.seeded_state_dict(vcfg, seed=0)
    _write_hf_dir(tmp_path / "vae", sd_v, vcfg)
    pair = build_pair(SMALL, seed=0, dtype=torch.float16)            # directly-assembled twin (same seeds -> same tensors)
    rk = dict(dim=64, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=64, ff_mult=2)
    torch.manual_seed(3)
    proj0 = Resampler(**rk)
    ck = {}
    ck.update({"ref_unet." + k: v for k, v in sd_r.items()})
    ck.update({"unet." + k: v for k, v in sd_u.items()})
    ck.update({"proj." + k: v.detach().cpu() for k, v in proj0.state_dict().items()})
    ck.update({"adapter_modules." + k: v.detach().float().cpu()
               for k, v in torch.nn.ModuleList(pair["e_unet"].attn_processors.values()).state_dict().items()})
    torch.save({"module": ck}, tmp_path / "IMAGDressing-v1_small.pt")

    class args:
        device = "cuda"
        model_ckpt = str(tmp_path / "IMAGDressing-v1_small.pt")

    # ================= body of prepare(), inference_IMAGdressing.py:41-135 (tokenizer / CLIP encoders left out: the test
    # feeds embeddings) =================
    generator = torch.Generator(device="cpu").manual_seed(42)
    vae = AutoencoderKL.from_pretrained(str(tmp_path / "vae")).to(dtype=torch.float16, device=args.device)
    unet = UNet2DConditionModel.from_pretrained(str(tmp_path / "rv"), subfolder="unet").to(
        dtype=torch.float16,
        device=args.device)
    image_proj = Resampler(
        dim=unet.config.cross_attention_dim,
        depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96,
        output_dim=unet.config.cross_attention_dim,
        ff_mult=2
    )
    image_proj = image_proj.to(dtype=torch.float16, device=args.device)
    attn_procs = {}
    st = unet.state_dict()
    for name in unet.attn_processors.keys():
        cross_attention_dim = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
        if name.startswith("mid_block"):
            hidden_size = unet.config.block_out_channels[-1]
        elif name.startswith("up_blocks"):
            block_id = int(name[len("up_blocks."):].split(".")[0])
            hidden_size = list(reversed(unet.config.block_out_channels))[block_id]
        elif name.startswith("down_blocks"):
            block_id = int(name[len("down_blocks."):].split(".")[0])
            hidden_size = unet.config.block_out_channels[block_id]
        if cross_attention_dim is None:
            attn_procs[name] = RefSAttnProcessor2_0(name, hidden_size)
        else:
            attn_procs[name] = CAttnProcessor2_0(name, hidden_size=hidden_size, cross_attention_dim=cross_attention_dim)
    unet.set_attn_processor(attn_procs)
    adapter_modules = torch.nn.ModuleList(unet.attn_processors.values())
    adapter_modules = adapter_modules.to(dtype=torch.float16, device=args.device)
    del st
    ref_unet = UNet2DConditionModel.from_pretrained(str(tmp_path / "rv"), subfolder="unet").to(
        dtype=torch.float16,
        device=args.device)
    ref_unet.set_attn_processor(
        {name: CacheAttnProcessor2_0() for name in ref_unet.attn_processors.keys()})  # set cache
    model_sd = torch.load(args.model_ckpt, map_location="cpu")["module"]
    ref_unet_dict = {}
    unet_dict = {}
    image_proj_dict = {}
    adapter_modules_dict = {}
    for k in model_sd.keys():
        if k.startswith("ref_unet"):
            ref_unet_dict[k.replace("ref_unet.", "")] = model_sd[k]
        elif k.startswith("unet"):
            unet_dict[k.replace("unet.", "")] = model_sd[k]
        elif k.startswith("proj"):
            image_proj_dict[k.replace("proj.", "")] = model_sd[k]
        elif k.startswith("adapter_modules"):
            adapter_modules_dict[k.replace("adapter_modules.", "")] = model_sd[k]
        else:
            raise AssertionError(k)
    ref_unet.load_state_dict(ref_unet_dict)
    image_proj.load_state_dict(image_proj_dict)
    adapter_modules.load_state_dict(adapter_modules_dict)
    noise_scheduler = DDIMScheduler(
        num_train_timesteps=1000,
        beta_start=0.00085,
        beta_end=0.012,
        beta_schedule="scaled_linear",
        clip_sample=False,
        set_alpha_to_one=False,
        steps_offset=1,
    )
    pipe = IMAGDressing_v1(unet=unet, reference_unet=ref_unet, vae=vae, tokenizer=None,
                           text_encoder=None, image_encoder=None,
                           ImgProj=image_proj,
                           scheduler=noise_scheduler,
                           safety_checker=StableDiffusionSafetyChecker,
                           feature_extractor=CLIPImageProcessor)
    # ================= end of prepare() =================
    assert all(isinstance(p, CacheAttnProcessor2_0) for p in ref_unet.attn_processors.values())     # kept across load_state_dict
    twin = IMAGDressing_v1(unet=pair["e_unet"], reference_unet=pair["e_ref"], vae=None, tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=proj0.to(device="cuda", dtype=torch.float16),
                           scheduler=DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                   clip_sample=False, set_alpha_to_one=False, steps_offset=1),
                           safety_checker=None, feature_extractor=None)
    kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128, num_inference_steps=4,
              guidance_scale=7.5, num_images_per_prompt=1, image_scale=1.0, generator=generator,
              prompt_embeds=g(10, 1, 77, 64, scale=0.5).cuda(), negative_prompt_embeds=g(11, 1, 77, 64, scale=0.5).cuda(),
              ref_clip_hidden_states=g(12, 1, 20, 96, scale=0.5).cuda().half(), ref_image_latents=g(13, 1, 4, 16, 16).cuda(),
              latents=g(14, 1, 4, 16, 16).cuda(), output_type="latent")
    a = pipe(**kw).images
    b = twin(**kw).images
    assert torch.isfinite(a).all() and torch.equal(a, b)
    pipe.set_scale(0.25)
    assert all(p.scale == 0.25 for p in unet.attn_processors.values() if isinstance(p, RefSAttnProcessor2_0))
    # the VAE handle built by from_pretrained decodes (output_type="pt") through the script's call sequence (:544-546)
    img = pipe(**dict(kw, output_type="pt")).images
    assert img.shape == (1, 3, 128, 128) and torch.isfinite(img).all() and float(img.min()) >= 0.0 and float(img.max()) <= 1.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@torch.no_grad()
def test_pipeline_unipc_10_steps_vs_oracle(dtype):
    """The paper's sampler (UniPC, supplementary p.1; /root/reference/app.py:28) through the HIP pipeline for 10 steps against
    the reference loop semantics (oracle/pipeline.py: two B = 1 UNet calls per step, custom CFG) driven by oracle/unipc.py,
    the float64 restatement of the library's ``multistep_uni_p/c_bh_update``.  Same relative bars as the DDIM trajectories."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import UniPCMultistepScheduler
    from oracle.pipeline import denoise
    from oracle.unipc import UniPCOracle
    p = build_pair(SMALL, seed=0, dtype=dtype)
    steps, gs = 10, 7.0
    lat = torch.stack([g(42 + i, 4, 16, 16) for i in range(2)])
    pe, ne = g(10, 1, 77, 64, scale=0.5), g(11, 1, 77, 64, scale=0.5)
    cloth = g(12, 2, 16, 64, scale=0.5); refl = g(13, 1, 4, 16, 16)
    ref = torch.cat([denoise(p["o_unet"], p["o_ref"], UniPCOracle(), lat[i:i + 1], pe, ne, cloth, refl, steps, gs) for i in range(2)])
    sch = UniPCMultistepScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=lambda h: h, scheduler=sch, safety_checker=None, feature_extractor=None)
    out = pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128,
               num_inference_steps=steps, guidance_scale=gs, num_images_per_prompt=2, prompt_embeds=pe.cuda(),
               negative_prompt_embeds=ne.cuda(), ref_clip_hidden_states=cloth[1:2].cuda(), ref_image_latents=refl.cuda(),
               latents=lat.cuda(), output_type="latent").images
    st = err_stats(out, ref); record(f"pipeline_unipc_10_steps[{dtype}]", st)
    bar = _traj_bar(dtype)
    assert torch.isfinite(out).all()
    assert st["max_abs"] < bar["max_abs"] * max(st["ref_std"], 1.0) and st["rel_rms"] < bar["rel_rms"], st


@torch.no_grad()
def test_pipeline_unipc_sampler_small():
    """UniPC through the pipeline == the same coefficient lists applied by hand in fp64 to the same UNet
    outputs (the scheduler's host math is tested separately on the CPU); 10 steps beat nothing here -- this checks plumbing:
    CFG folded into the x0 prediction, history handling, the emitted 16-bit UNet input."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import UniPCMultistepScheduler
    dt = torch.float16
    p = build_pair(SMALL, seed=0, dtype=dt)
    mk = lambda: UniPCMultistepScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")   # noqa: E731

    class Proj:
        def __call__(self, h):
            return h
    sch = mk()
    pipe = IMAGDressing_v1(vae=None, reference_unet=p["e_ref"], unet=p["e_unet"], tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=Proj(), scheduler=sch, safety_checker=None, feature_extractor=None)
    lat = g(20, 2, 4, 16, 16)
    steps, gs = 6, 7.5
    trace = []
    kw = dict(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=128, height=128, num_inference_steps=steps,
              guidance_scale=gs, num_images_per_prompt=2, prompt_embeds=g(10, 1, 77, 64, scale=0.5).cuda(),
              negative_prompt_embeds=g(11, 1, 77, 64, scale=0.5).cuda(), ref_clip_hidden_states=g(12, 1, 16, 64, scale=0.5).cuda(),
              ref_image_latents=g(13, 1, 4, 16, 16).cuda(), latents=lat.cuda(), output_type="latent")
    out = pipe(trace=trace, **kw).images
    assert torch.isfinite(out).all() and len(trace) == steps
    assert [int(t) for t in sch.timesteps] == [999, 832, 666, 500, 333, 166]
    # deterministic and restartable (set_timesteps resets the history)
    assert torch.equal(pipe(**kw).images, out)
    # the first step has no history: it is the DDIM update from sigma(999) to sigma(832) with the guided epsilon
    ref = mk(); ref.set_timesteps(steps)
    a0, s0 = ref._alpha_sigma(0); a1, s1 = ref._alpha_sigma(1)
    z0 = lat.cuda().permute(0, 2, 3, 1).reshape(2, 256, 4)
    from imagdressing_amd.unet import nchw_to_nhwc8
    x_in = torch.cat([nchw_to_nhwc8(lat.cuda(), dt)] * 2)
    p["e_ref"].forward_nhwc(nchw_to_nhwc8(kw["ref_image_latents"], dt), 0, kw["ref_clip_hidden_states"].to(dt).contiguous())
    sa = {n: pr.cache["hidden_states"] for n, pr in p["e_ref"].attn_processors.items()}
    ehs = torch.cat([kw["prompt_embeds"], kw["negative_prompt_embeds"]]).to(dt).contiguous()
    mask = torch.cat([torch.ones(2), torch.zeros(2)]).cuda()
    eps = p["e_unet"].forward_nhwc(x_in, 999, ehs, {"sa_hidden_states": sa, "sa_batch_mask": mask}).view(4, 256, 4).double()
    e = gs * eps[:2] + (1 - gs) * eps[2:]
    want = a1 * (z0.double() - s0 * e) / a0 + s1 * e
    assert torch.allclose(trace[0].double(), want, rtol=1e-4, atol=1e-4)
