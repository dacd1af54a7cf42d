"""Host-side logic that needs no GPU: schedule, parameter inventory, plugin surface, sharding."""
import os

import pytest
from types import SimpleNamespace
import torch

from imagdressing_amd import dist as D
from imagdressing_amd import unet as E
from imagdressing_amd.scheduler import DDIMScheduler
from oracle import sd15
from oracle.ddim import DDIMOracle


def make_sched():
    return DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                         clip_sample=False, set_alpha_to_one=False, steps_offset=1)


@pytest.mark.parametrize("n", [20, 50])
def test_ddim_schedule_matches_oracle_and_known_values(n):
    s, o = make_sched(), DDIMOracle()
    s.set_timesteps(n); ts = o.set_timesteps(n)
    assert s.timesteps.tolist() == ts.tolist()
    # leading spacing with steps_offset=1: t_i = (1000/n)(n-1-i) + 1
    assert ts.tolist() == [(1000 // n) * (n - 1 - i) + 1 for i in range(n)]
    assert ts[0] == 1000 - 1000 // n + 1 and ts[-1] == 1
    for t in ts.tolist():
        assert s.alpha(t) == pytest.approx(float(o.alphas_cumprod[t]), rel=0, abs=0)
        prev = t - 1000 // n
        exp = float(o.alphas_cumprod[prev]) if prev >= 0 else float(o.alphas_cumprod[0])   # set_alpha_to_one=False
        assert s.alpha_prev(t) == exp
    # published SD schedule end points: alphas_cumprod[0] = 1 - 0.00085, [999] ~ 0.00466
    assert float(o.alphas_cumprod[0]) == pytest.approx(1 - 0.00085, abs=1e-7)
    assert float(o.alphas_cumprod[999]) == pytest.approx(0.004660, abs=2e-5)


def test_stochastic_ddim_step_of_the_oracle_and_the_host_sigma():
    """eta > 0: the oracle's step is formula (12)/(16) of the DDIM paper as diffusers 0.24 writes it; the host-side sigma handed
    to the fused kernel is the same number; eta = 1 at the 1000-step schedule is the DDPM posterior variance"""
    s, o = make_sched(), DDIMOracle()
    s.set_timesteps(50); ts = o.set_timesteps(50)
    x = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(0)); e = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    n = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(2))
    for t in (int(ts[0]), int(ts[20]), int(ts[-1])):
        a_t, a_p = s.alpha(t), s.alpha_prev(t)
        for eta in (0.25, 1.0):
            std = eta * ((1 - a_p) / (1 - a_t) * (1 - a_t / a_p)) ** 0.5
            assert s.sigma(t, eta) == pytest.approx(std, rel=1e-12)
            det, sto = o.step(e, t, x), o.step(e, t, x, eta=eta, variance_noise=n)
            # stochastic = deterministic with the direction coefficient shrunk, plus std * noise
            exp = det - ((1 - a_p) ** 0.5 - (1 - a_p - std ** 2) ** 0.5) * e + std * n
            assert torch.allclose(sto, exp, atol=1e-6)
        assert torch.equal(o.step(e, t, x, eta=0.0), o.step(e, t, x))
    # eta = 1 on consecutive timesteps: sigma^2 = beta_tilde_t = (1 - a_{t-1}) / (1 - a_t) * beta_t (DDPM posterior variance)
    s.set_timesteps(1000)
    t = 500
    beta_t = 1 - s.alpha(t) / s.alpha(t - 1)
    assert s.alpha_prev(t) == s.alpha(t - 1)
    assert s.sigma(t, 1.0) ** 2 == pytest.approx((1 - s.alpha(t - 1)) / (1 - s.alpha(t)) * beta_t, rel=1e-9)


def test_unet_and_controlnet_inventories_match_oracle_modules():
    u = sd15.UNet2DConditionModel().state_dict()
    s = E.unet_param_shapes()
    assert set(u) == set(s) and all(tuple(u[k].shape) == s[k] for k in s)
    assert sum(v.numel() for v in u.values()) == 859_520_964          # SD1.5 UNet parameter count
    c = sd15.ControlNetModel().state_dict()
    cs = E.controlnet_param_shapes()
    assert set(c) == set(cs) and all(tuple(c[k].shape) == cs[k] for k in cs)
    assert sum(v.numel() for v in c.values()) == 361_279_120          # SD1.5 ControlNet parameter count


def test_processor_names_and_order():
    names = list(sd15.UNet2DConditionModel().attn_processors.keys())
    assert len(names) == 32
    assert names[0] == "down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor"
    assert names[12].startswith("up_blocks.1.") and names[-1] == "mid_block.attentions.0.transformer_blocks.0.attn2.processor"
    assert [i for i, n in enumerate(names) if n.endswith("attn1.processor")] == list(range(0, 32, 2))   # adapter_modules.{even}


REF = "/root/reference/adapter/attention_processor.py"


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference checkout not present")
def test_plugin_surface_matches_reference():
    """same class names, ctor signatures, state_dict keys and isinstance relations as the reference module"""
    import inspect
    from imagdressing_amd.adapter import attention_processor as A
    from imagdressing_amd.adapter import resampler as R
    from oracle.ref_loader import load_reference_adapter
    ap, rs = load_reference_adapter()
    ctor = {
        "RefSAttnProcessor2_0": ("n", 64), "LoraRefSAttnProcessor2_0": ("n", 64), "RefLoraSAttnProcessor2_0": ("n", 64),
        "CAttnProcessor2_0": ("n", 64, 96), "LoRAIPAttnProcessor2_0": (64, 96), "IPAttnProcessor2_0": (64, 96),
        "SAttnProcessor2_0": ("n", 64), "BaseSAttnProcessor2_0": ("n", 64), "RefCAttnProcessor2_0": ("n", 64, 96),
    }
    for cls, args in ctor.items():
        ours, theirs = getattr(A, cls), getattr(ap, cls)
        assert list(inspect.signature(ours.__init__).parameters) == list(inspect.signature(theirs.__init__).parameters), cls
        so, st = ours(*args).state_dict(), theirs(*args).state_dict()
        assert {k: tuple(v.shape) for k, v in so.items()} == {k: tuple(v.shape) for k, v in st.items()}, cls
    assert list(inspect.signature(A.CacheAttnProcessor2_0.__init__).parameters) == ["self"]
    # sibling relations the reference pipelines' isinstance checks rely on
    assert not isinstance(A.RefLoraSAttnProcessor2_0("n", 64), A.LoraRefSAttnProcessor2_0)
    assert not isinstance(A.LoraRefSAttnProcessor2_0("n", 64), A.RefSAttnProcessor2_0)
    assert not isinstance(A.IPAttnProcessor2_0(64, 96), A.LoRAIPAttnProcessor2_0)
    cfg = dict(dim=64, depth=2, dim_head=16, heads=4, num_queries=4, embedding_dim=48, output_dim=32, ff_mult=4)
    assert list(inspect.signature(R.Resampler.__init__).parameters) == list(inspect.signature(rs.Resampler.__init__).parameters)
    a, b = R.Resampler(**cfg).state_dict(), rs.Resampler(**cfg).state_dict()
    assert {k: tuple(v.shape) for k, v in a.items()} == {k: tuple(v.shape) for k, v in b.items()}
    a, b = R.ProjPlusModel().state_dict(), rs.ProjPlusModel().state_dict()
    assert {k: tuple(v.shape) for k, v in a.items()} == {k: tuple(v.shape) for k, v in b.items()}
    assert list(inspect.signature(R.ProjPlusModel.forward).parameters) == list(inspect.signature(rs.ProjPlusModel.forward).parameters)


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference checkout not present")
def test_pipeline_call_signatures_cover_reference():
    """every keyword the reference's __call__ / __init__ accept is accepted here (extensions are appended)"""
    import ast
    import inspect
    import importlib
    for mod in ("IMAGDressing_v1_pipeline", "IMAGDressing_v1_pipeline_controlnet", "IMAGDressing_v1_pipeline_ipa_controlnet",
                "IMAGDressing_v1_pipeline_controlnet_inpainting"):
        tree = ast.parse(open(f"/root/reference/dressing_sd/pipelines/{mod}.py").read())
        cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "IMAGDressing_v1"][0]
        ref = {f.name: [a.arg for a in f.args.args] for f in cls.body if isinstance(f, ast.FunctionDef)}
        ours = importlib.import_module(f"dressing_sd.pipelines.{mod}").IMAGDressing_v1
        for fn in ("__init__", "__call__"):
            mine = list(inspect.signature(getattr(ours, fn)).parameters)
            assert mine[:len(ref[fn])] == ref[fn], (mod, fn, mine, ref[fn])
        for meth in ("set_scale",) + (("set_ipa_scale", "get_image_embeds", "init_proj", "load_ip_adapter") if "ipa" in mod else ()):
            assert hasattr(ours, meth), (mod, meth)


def test_set_scale_semantics():
    from imagdressing_amd.adapter import attention_processor as A
    from imagdressing_amd.dressing_sd.pipelines._base import set_scale_by_type

    class U:
        attn_processors = {"a": A.RefSAttnProcessor2_0("a", 64), "b": A.LoraRefSAttnProcessor2_0("b", 64),
                           "c": A.RefLoraSAttnProcessor2_0("c", 64), "d": A.CAttnProcessor2_0("d", 64, 96)}
    set_scale_by_type(U, A.RefSAttnProcessor2_0, scale=0.3)
    assert U.attn_processors["a"].scale == 0.3 and U.attn_processors["b"].scale == 1.0
    set_scale_by_type(U, A.LoraRefSAttnProcessor2_0, scale=0.7, lora_scale=0.2)
    assert (U.attn_processors["b"].scale, U.attn_processors["b"].lora_scale) == (0.7, 0.2)
    assert U.attn_processors["c"].scale == 1.0            # app.py's class is NOT matched, as in the reference


def test_tensor_cache_tracks_identity_and_version():
    from imagdressing_amd.adapter.attention_processor import _TensorCache
    c = _TensorCache(2)
    a = torch.zeros(4)
    assert c.get((a,)) is None
    c.put((a,), "v1")
    assert c.get((a,)) == "v1"
    a.add_(1)                                   # in-place change bumps the version -> stale entry is not returned
    assert c.get((a,)) is None
    b = torch.zeros(4)
    c.put((b,), "vb"); c.put((a,), "va")
    assert c.get((b,)) == "vb" and c.get((a,)) == "va"
    c.put((torch.ones(1),), "x")               # capacity 2: the least recently used entry is dropped
    assert sum(c.get((t,)) is not None for t in (a, b)) == 1


def test_shard_bounds_partition():
    for n in (1, 4, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_feature_layout_matches_oracle_garment_features():
    from oracle import processors as OP
    from oracle.pipeline import garment_features
    from tests.harness import SMALL, oracle_cfg
    o = sd15.UNet2DConditionModel(oracle_cfg(SMALL))
    o.set_attn_processor({n: OP.CacheAttn() for n in o.attn_processors})
    with torch.no_grad():
        feats = garment_features(o, torch.randn(1, 4, 16, 24), torch.randn(2, 16, 64))

    class U:
        cfg = dict(E.SD15_CONFIG, **SMALL)
        attn_processors = feats
    layout = D.feature_layout(U, (16, 24))
    assert [(n, tuple(feats[n].shape)) for n in feats] == layout
    names = [n for n, _ in layout if n.endswith("attn1.processor")]
    flat, lay = D.pack_features(feats, names)
    back = D.unpack_features(flat, lay)
    assert all(torch.equal(back[n], feats[n]) for n in names)


def test_controlnet_keep():
    from imagdressing_amd.dressing_sd.pipelines._base import controlnet_keep
    assert controlnet_keep(4, 0.0, 1.0) == [1.0] * 4
    assert controlnet_keep(4, 0.5, 1.0) == [0.0, 0.0, 1.0, 1.0]
    assert controlnet_keep(4, 0.0, 0.5) == [1.0, 1.0, 0.0, 0.0]


def test_vae_state_dict_layout_matches_diffusers_shape_count():
    """SD1.5 AutoencoderKL: 248 tensors, 83,653,863 parameters; the engine's shape table == the oracle module's keys"""
    import math

    from imagdressing_amd.vae import vae_param_shapes
    from oracle import vae as OV
    sh = vae_param_shapes()
    assert len(sh) == 248 and sum(math.prod(v) for v in sh.values()) == 83_653_863
    o = OV.AutoencoderKL().state_dict()
    assert set(o) == set(sh) and all(tuple(o[k].shape) == sh[k] for k in sh)
    small = dict(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8)
    m = OV.AutoencoderKL(small)
    assert set(m.state_dict()) == set(vae_param_shapes(small))
    mean, logvar = m.encode_moments(torch.zeros(1, 3, 16, 16))
    assert mean.shape == logvar.shape == (1, 4, 8, 8) and m.decode(mean).shape == (1, 3, 16, 16)


def test_imagdressing_checkpoint_split_and_file_roundtrip(tmp_path):
    """key routing of inference_IMAGdressing.py:99-112 and the DeepSpeed {"module": ...} unwrapping (:97)"""
    from imagdressing_amd import checkpoint as CK
    sd = {"ref_unet.conv_in.weight": torch.ones(2), "unet.conv_in.weight": torch.zeros(2), "proj.latents": torch.ones(1, 2, 3),
          "adapter_modules.0.to_k_ref.weight": torch.ones(3, 3), "adapter_modules.31.to_v_ref.weight": torch.ones(3, 3),
          "projection_extra": torch.ones(1), "optimizer_state": torch.ones(1)}
    parts = CK.split_imagdressing_state_dict(sd)
    assert set(parts["ref_unet"]) == {"conv_in.weight"} and set(parts["unet"]) == {"conv_in.weight"}
    assert set(parts["proj"]) == {"latents", "projection_extra"}      # the reference's startswith("proj") also swallows this key
    assert set(parts["adapter_modules"]) == {"0.to_k_ref.weight", "31.to_v_ref.weight"} and set(parts["other"]) == {"optimizer_state"}
    f = tmp_path / "ck.pt"
    torch.save({"module": sd, "global_steps": 3}, f)
    back = CK.load_state_dict_file(str(f))
    assert set(back) == set(sd) and torch.equal(back["proj.latents"], sd["proj.latents"])
    from safetensors.torch import save_file
    g = tmp_path / "w.safetensors"
    save_file({k: v.contiguous() for k, v in sd.items()}, str(g))
    assert set(CK.load_state_dict_file(str(g))) == set(sd)
    assert CK.hidden_size_of("up_blocks.1.attentions.0.transformer_blocks.0.attn1.processor", (320, 640, 1280, 1280)) == 1280
    assert CK.hidden_size_of("down_blocks.1.attentions.0.transformer_blocks.0.attn2.processor", (320, 640, 1280, 1280)) == 640


# ---------------------------------------------------------------------------------------------------------------------
# UniPC multistep sampler: host-side coefficient math, applied here with numpy
# ---------------------------------------------------------------------------------------------------------------------
def _unipc_run(order, N, corrector, stop, model, x, lower_order_final=False):
    """drive UniPCMultistepScheduler's coefficient lists exactly as `_advance` does, on numpy vectors"""
    import numpy as np
    from imagdressing_amd.scheduler import UniPCMultistepScheduler
    sch = UniPCMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", solver_order=order,
                                  disable_corrector=() if corrector else tuple(range(N)), lower_order_final=lower_order_final)
    sch.set_timesteps(N)
    for pos in range(stop):
        a, s = sch._alpha_sigma(pos)
        named = {"x": x, "eps": model(x, a, s)}
        mt = sum(c * named[n] for c, n in sch.x0_terms(pos))
        sch.step_index = pos
        hist = {f"m{k}": m for k, m in enumerate(reversed(sch.model_outputs))}
        if pos > 0 and (pos - 1) not in sch.disable_corrector and sch.last_sample is not None:
            d = dict(hist, x=sch.last_sample, mt=mt)
            x = sum(c * d[n] for c, n in sch.corrector_terms(pos, sch.this_order))
        sch.model_outputs = (sch.model_outputs + [mt])[-order:]
        sch.ts_hist = (sch.ts_hist + [pos])[-order:]
        sch.this_order = sch._order_now()
        sch.last_sample = x
        hist = {f"m{k}": m for k, m in enumerate(reversed(sch.model_outputs))}
        d = dict(hist, x=x)
        x = sum(c * d[n] for c, n in sch.predictor_terms(pos, sch.this_order))
        if sch.lower_order_nums < order:
            sch.lower_order_nums += 1
    return sch, x


def test_unipc_convergence_orders_on_the_gaussian_case():
    """Data ~ N(0, s^2): the optimal denoiser is linear and the probability-flow ODE has the closed form
    x_t = x_T sqrt(a_t^2 s^2 + sigma_t^2) / sqrt(a_T^2 s^2 + sigma_T^2).  On the smooth first half of the schedule the global
    error must fall like N^-1 (UniP-1 = DDIM), N^-2 (UniP-1 + UniC, UniP-2), N^-3 (UniPC-2), N^-4 (UniPC-3)."""
    import numpy as np
    sd = 0.7

    def model(x, a, s):
        return (x - a * (a * sd * sd / (a * a * sd * sd + s * s)) * x) / s

    def err(order, N, corr):
        x0 = np.array([1.3, -0.4, 2.0])
        sch, x = _unipc_run(order, N, corr, N // 2, model, x0.copy())
        a0, s0 = sch._alpha_sigma(0); a1, s1 = sch._alpha_sigma(N // 2)
        exact = x0 * np.sqrt(a1 * a1 * sd * sd + s1 * s1) / np.sqrt(a0 * a0 * sd * sd + s0 * s0)
        return np.abs(x - exact).max() / np.abs(exact).max()
    for order, corr, rate in ((1, False, 1), (1, True, 2), (2, False, 2), (2, True, 3), (3, True, 4)):
        e40, e80 = err(order, 40, corr), err(order, 80, corr)
        assert 2 ** rate * 0.6 < e40 / e80 < 2 ** rate * 1.6, (order, corr, e40, e80)
    assert err(2, 40, True) < 1e-4 < err(1, 40, False)


def test_unipc_order1_is_ddim_and_constant_prediction_is_exact():
    import numpy as np
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(5)
    # (1) order 1, no corrector: every step is the deterministic DDIM update x' = a' x0_pred + s' eps
    eps_fix = rng.standard_normal(5)
    sch, x1 = _unipc_run(1, 10, False, 1, lambda x, a, s: eps_fix, x0.copy())
    a, s = sch._alpha_sigma(0); a2, s2 = sch._alpha_sigma(1)
    ddim = a2 * (x0 - s * eps_fix) / a + s2 * eps_fix
    assert np.allclose(x1, ddim, rtol=1e-12, atol=1e-12)
    # (2) a model whose data prediction is the constant c is integrated exactly by every order (all differences vanish)
    c = rng.standard_normal(5)
    for order in (1, 2, 3):
        sch, x = _unipc_run(order, 12, True, 12, lambda x, a, s: (x - a * c) / s, x0.copy(), lower_order_final=True)
        a0, s0 = sch._alpha_sigma(0); aT, sT = sch._alpha_sigma(12)
        assert np.allclose(x, sT / s0 * (x0 - a0 * c) + aT * c, rtol=1e-9, atol=1e-9)
    # timesteps: "linspace" spacing of diffusers (50 steps: 999, 979, ..., 20), final sigma = that of training timestep 0
    from imagdressing_amd.scheduler import UniPCMultistepScheduler
    s50 = UniPCMultistepScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    s50.set_timesteps(50)
    ts = [int(t) for t in s50.timesteps]
    assert ts[0] == 999 and ts[-1] == 20 and len(ts) == 50 and all(a > b for a, b in zip(ts, ts[1:]))
    assert abs(s50._sig[-1] - ((1 - float(s50._ac[0])) / float(s50._ac[0])) ** 0.5) < 1e-12


def test_unipc_coefficient_lists_equal_the_library_form_oracle():
    """imagdressing_amd.scheduler.UniPCMultistepScheduler (flat coefficient lists, applied by one fused launch each) against
    oracle/unipc.py (the library's own tensor-form update functions, float64) over whole 10- and 25-step trajectories with a
    nonlinear, state-dependent epsilon model -- orders 1 to 3, corrector on, lower_order_final as the library defaults."""
    import numpy as np
    import torch
    from oracle.unipc import UniPCOracle
    rng = np.random.default_rng(3)
    W = rng.standard_normal((6, 6)) * 0.4

    def eps_model(x, pos):           # smooth, nonlinear in x, different at every step
        return np.tanh(x @ W + 0.1 * pos) + 0.05 * x
    for order in (1, 2, 3):
        for N in (10, 25):
            x0 = rng.standard_normal((2, 6))
            orc = UniPCOracle(solver_order=order)
            ts = orc.set_timesteps(N)
            xo = torch.from_numpy(x0.copy())
            x = x0.copy()
            sch, _ = _unipc_run(order, N, True, 0, None, x, lower_order_final=True)
            assert [int(t) for t in sch.timesteps] == [int(t) for t in ts]
            for pos in range(N):
                xo = orc.step(torch.from_numpy(eps_model(xo.numpy(), pos)), ts[pos], xo)
                # the same update sequence `_advance` performs, on numpy vectors (see _unipc_run)
                named = {"x": x, "eps": eps_model(x, pos)}
                mt = sum(c * named[n] for c, n in sch.x0_terms(pos))
                sch.step_index = pos
                hist = {f"m{k}": m for k, m in enumerate(reversed(sch.model_outputs))}
                if pos > 0 and sch.last_sample is not None:
                    d = dict(hist, x=sch.last_sample, mt=mt)
                    x = sum(c * d[n] for c, n in sch.corrector_terms(pos, sch.this_order))
                sch.model_outputs = (sch.model_outputs + [mt])[-order:]
                sch.ts_hist = (sch.ts_hist + [pos])[-order:]
                sch.this_order = sch._order_now()
                sch.last_sample = x
                hist = {f"m{k}": m for k, m in enumerate(reversed(sch.model_outputs))}
                x = sum(c * dict(hist, x=x)[n] for c, n in sch.predictor_terms(pos, sch.this_order))
                if sch.lower_order_nums < order:
                    sch.lower_order_nums += 1
                assert np.allclose(x, xo.numpy(), rtol=1e-9, atol=1e-10), (order, N, pos, np.abs(x - xo.numpy()).max())


def test_pack_ff_fused_is_an_exact_reparametrisation():
    """ops.pack_ff_fused (operands of the one-launch feed-forward, csrc/ff_fused.hip): emulate on the CPU what the kernel does with
    the packed tensors -- first GEMM in packed row order, value / gate pairing by accumulator register (rows i and i + 8), the
    lane's GEGLU outputs taken in register order as the k-slots of the packed second GEMM -- and compare with
    LayerNorm -> Linear -> GEGLU -> Linear + residual on the original parameters."""
    import torch.nn.functional as F
    from imagdressing_amd import ops
    g = torch.Generator().manual_seed(0)
    Cc, inner, M = 320, 1280, 6
    x = torch.randn(M, Cc, generator=g, dtype=torch.float64)
    w1 = torch.randn(2 * inner, Cc, generator=g, dtype=torch.float64) * Cc ** -0.5; b1 = torch.randn(2 * inner, generator=g, dtype=torch.float64)
    w2 = torch.randn(Cc, inner, generator=g, dtype=torch.float64) * inner ** -0.5; b2 = torch.randn(Cc, generator=g, dtype=torch.float64)
    gam = 1 + 0.3 * torch.randn(Cc, generator=g, dtype=torch.float64); bet = 0.2 * torch.randn(Cc, generator=g, dtype=torch.float64)
    pk = ops.pack_ff_fused(w1, b1, w2, b2, gam, bet, dtype=torch.float64)
    assert pk["w1"].shape == (2 * inner, Cc) and pk["b1"].shape == (2 * inner,) and pk["w2"].shape == (inner // 32, Cc, 32) and pk["ln"]
    n = F.layer_norm(x, (Cc,), None, None, 1e-5)                      # the kernel normalises WITHOUT affine
    out = x + b2
    for blk in range(inner // 16):                                    # one 32-row MFMA block of the first GEMM = 16 inner channels
        acc = n @ pk["w1"][32 * blk:32 * blk + 32].t() + pk["b1"][32 * blk:32 * blk + 32]          # [M, 32 packed rows]
        h = torch.empty(M, 16, dtype=torch.float64)                   # k-slot order of the second GEMM
        for hi in range(2):                                           # lane half: accumulator register r <-> packed row (r & 3) + 8 (r >> 2) + 4 hi
            row = lambda r: (r & 3) + 8 * (r >> 2) + 4 * hi
            for e in range(4):
                h[:, 8 * hi + e] = acc[:, row(e)] * F.gelu(acc[:, row(4 + e)])
                h[:, 8 * hi + 4 + e] = acc[:, row(8 + e)] * F.gelu(acc[:, row(12 + e)])
        c, t = blk // 2, blk % 2
        out = out + h @ pk["w2"][c][:, 16 * t:16 * t + 16].t()
    hid, gate = F.linear(F.layer_norm(x, (Cc,), gam, bet, 1e-5), w1, b1).chunk(2, dim=-1)
    ref = x + F.linear(hid * F.gelu(gate), w2, b2)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5), float((out - ref).abs().max())      # (the packer folds in fp32)


def test_tuning_table_only_names_known_tile_configs():
    import json
    import os
    from imagdressing_amd import ops
    with open(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "gemm_tuning.json")) as f:
        shapes = json.load(f)["shapes"]
    assert len(shapes) > 250
    for key, ent in shapes.items():
        base, _, geom = key.partition("|")          # 3x3 convs may carry their output map: "M,N,K,taps,stride,ups|HxW" (round 3)
        M, N, K, taps, stride, ups = map(int, base.split(","))
        if geom:
            Ho, Wo = map(int, geom.split("x"))
            assert taps == 9 and M % (Ho * Wo) == 0, key
        for c in (ent["cfg"], ent["cfg_nosplit"]):
            assert c in range(0, 33), (key, ent)
            if c in (17, 19, 25, 27, 30, 31, 32): assert K % 64 == 0 and taps == 1, key
            if c in (18, 20): assert taps == 9 and (K // 9) % 32 == 0, key
            if c in (26, 28): assert taps == 9 and (K // 9) % 64 == 0, key
            if c == 12: assert K == 320 and N <= 320 and N % 64 == 0 and taps == 1, key
            if c == 13: assert K == 640 and N % 160 == 0 and taps == 1, key
            if c == 14: assert K == 1280 and N % 160 == 0 and taps == 1, key
            if c == 15: assert K == 320 and N == 960 and taps == 1, key
            if c == 16: assert K % 64 == 0 and taps == 1, key
            if c == 29: assert taps == 9 and stride == 1 and (K // 9) % 64 == 0, key
            if c in (5, 21, 22, 23, 29):
                assert taps == 9 and stride == 1, key
                if geom: assert Wo >= 16 and Ho >= 8, key          # the halo-patch kernel's 8 x 16 pixel tiles
            if c == 24:
                assert taps == 9 and stride == 1 and geom and Wo == 8 and Ho <= 12 and ent["split"] >= 2 and N % 64 == 0 and (K // 9) % 32 == 0, key
        assert ent["split"] >= 1 and (ent["split"] == 1 or ent["cfg"] not in (12, 13, 14, 15, 16)), (key, ent)


def test_fold_layernorm_affine_is_exact_algebra():
    """LN_affine(x) W^T + b == LN_plain(x) W'^T + b' with (W', b') = ops.fold_layernorm_affine (the host side of the fused
    LayerNorm -> linear launches)"""
    import torch.nn.functional as F
    from imagdressing_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 320, generator=g, dtype=torch.float64) * 2 + 0.3
    w = torch.randn(96, 320, generator=g, dtype=torch.float64); b = torch.randn(96, generator=g, dtype=torch.float64)
    gam = 1 + 0.4 * torch.randn(320, generator=g, dtype=torch.float64); bet = 0.3 * torch.randn(320, generator=g, dtype=torch.float64)
    w2, b2 = ops.fold_layernorm_affine(w, b, gam, bet)
    ref = F.linear(F.layer_norm(x, (320,), gam, bet, 1e-5), w, b)
    got = F.linear(F.layer_norm(x, (320,), None, None, 1e-5), w2.double(), b2.double())
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-5), float((got - ref).abs().max())      # (the fold itself runs in fp32)
    w3, b3 = ops.fold_layernorm_affine(w, None, gam, bet)
    assert torch.allclose(b3.double(), w @ bet, atol=1e-5)


def test_engine_attention_hands_unnormalised_states_only_to_processors_that_opted_in(monkeypatch):
    """unet.Attention.__call__(layernorm=...): processors of this package that declare ``fused_layernorm`` (and ``fused_residual``)
    receive the block's raw hidden state plus ``imd_layernorm``; every other processor -- the diffusers protocol -- receives
    normalised states and never sees the extra keyword."""
    from imagdressing_amd import ops, unet
    calls = []
    monkeypatch.setattr(ops, "layer_norm", lambda x, w, b, eps=1e-5, out=None: (calls.append("ln"), x + 1000.0)[1])
    monkeypatch.setattr(ops, "add", lambda a, b, *args, **kw: a + b)
    norm = SimpleNamespace(weight=torch.ones(320), bias=torch.zeros(320))

    class Foreign:                       # a diffusers-style processor: positional protocol, no opt-in attributes
        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            assert "imd_layernorm" not in kw and "imd_residual" not in kw
            self.seen = hidden_states
            return hidden_states

    class Engine:
        fused_residual = True
        fused_layernorm = True

        def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, imd_residual=None, imd_layernorm=None, **kw):
            self.seen, self.ln = hidden_states, imd_layernorm
            return hidden_states

    a = unet.Attention.__new__(unet.Attention)
    a.dtype = torch.float32
    h = torch.zeros(2, 128, 320); ehs = torch.zeros(2, 77, 768)
    a.processor = Foreign()
    a(h, encoder_hidden_states=ehs, residual=h, layernorm=(norm, 1e-5))
    assert calls == ["ln"] and float(a.processor.seen.min()) == 1000.0           # normalised by the engine, in front of the processor
    calls.clear()
    a.processor = Engine()
    a(h, encoder_hidden_states=ehs, residual=h, layernorm=(norm, 1e-5))
    assert calls == [] and float(a.processor.seen.max()) == 0.0 and a.processor.ln[2] == 1e-5       # raw state + the norm's parameters
    a(torch.zeros(2, 128, 96), encoder_hidden_states=ehs, residual=None, layernorm=(norm, 1e-5))     # a width no fused kernel exists for
    assert calls == ["ln"] and a.processor.ln is None
    calls.clear()
    monkeypatch.setattr(ops, "FUSED_LN", False)                                 # the A/B switch restores the two-launch path
    a(h, encoder_hidden_states=ehs, residual=h, layernorm=(norm, 1e-5))
    assert calls == ["ln"] and a.processor.ln is None


def test_tuning_scope_builds_per_call_flags():
    """ops.tuning_scope: the per-call tuning word that rides in the params blocks' `flags` (IMD_TUNING_PER_CALL, include/imagdressing_hip.h);
    scopes nest, None inherits, leaving a scope restores the outer one; no library knob is touched."""
    import re
    from imagdressing_amd import ops
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "imagdressing_hip.h")).read()
    assert int(re.search(r"#define IMD_TUNING_PER_CALL (0x[0-9a-fA-F]+)", hdr).group(1), 16) == ops.TUNING_PER_CALL
    assert ops._gemm_call_flags() == 0 and ops._attn_call_flags() == 0
    with ops.tuning_scope(gemm_flags=3):
        assert ops._gemm_call_flags() == ops.TUNING_PER_CALL | 3 and ops._attn_call_flags() == 0
        with ops.tuning_scope(attn_variant=12, attn_xcd=False):
            assert ops._gemm_call_flags() == ops.TUNING_PER_CALL | 3
            assert ops._attn_call_flags() == ops.TUNING_PER_CALL | 12 | 256
            with ops.tuning_scope(attn_xcd=True, gemm_flags=23):
                assert ops._attn_call_flags() == ops.TUNING_PER_CALL | 12
                assert ops._gemm_call_flags() == ops.TUNING_PER_CALL | 23
        assert ops._attn_call_flags() == 0
    assert ops._TUNING.get() is None
    with pytest.raises(ops.L.ImdError):
        with ops.tuning_scope(attn_variant=99):
            ops._attn_call_flags()
    with pytest.raises(ops.L.ImdError):          # 14..54 exist in -DIMD_ABLATIONS builds only: the same range imd_set_tuning(0, .) accepts
        with ops.tuning_scope(attn_variant=14):
            ops._attn_call_flags()
    assert ops._TUNING.get() is None
    # the tag is 8 bits wide (ABI v9) and the header's mask covers it
    assert ops.TUNING_PER_CALL & 0x00FFFFFF == 0 and int(re.search(r"#define IMD_TUNING_TAG_MASK (0x[0-9a-fA-F]+)u", hdr).group(1), 16) == 0xFF000000


def test_tuning_scope_is_per_thread():
    """The scope lives in a ContextVar: a second thread (a second pipeline on its own stream) neither sees the first thread's scope nor
    clobbers it when its own scope exits (round-5 advisor finding: a module global raced)."""
    import threading
    from imagdressing_amd import ops
    seen, go, done = {}, threading.Event(), threading.Event()

    def other():
        seen["inherits"] = ops._gemm_call_flags()
        with ops.tuning_scope(gemm_flags=1):
            seen["own"] = ops._gemm_call_flags()
            go.set(); done.wait(5)
        seen["after"] = ops._gemm_call_flags()
    with ops.tuning_scope(gemm_flags=3):
        t = threading.Thread(target=other); t.start(); go.wait(5)
        assert ops._gemm_call_flags() == ops.TUNING_PER_CALL | 3
        done.set(); t.join()
        assert ops._gemm_call_flags() == ops.TUNING_PER_CALL | 3          # the other thread's __exit__ did not restore over this scope
    assert seen == {"inherits": 0, "own": ops.TUNING_PER_CALL | 1, "after": 0}
