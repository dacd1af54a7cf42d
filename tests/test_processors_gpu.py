"""The HIP processors / resamplers (through the plugin surface and the C ABI) against the committed golden
vectors that the REFERENCE's own source produced (tests/golden/*.pt, oracle/make_golden.py).
fp16 is held to the north-star atol 1e-2; bf16 to 4e-2 (8 mantissa bits; see DESIGN.md 'Numerics')."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.cases import cross_inputs, hybrid_inputs, proj_plus_inputs, resampler_inputs  # noqa: E402

DT = pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
ATOL = {torch.float16: 1e-2, torch.bfloat16: 4e-2}
RTOL = {torch.float16: 0.0, torch.bfloat16: 1e-2}     # bf16: + 1 % of |ref| (outputs reach |4.4|; one bf16 ulp there is 3.1e-2)


def make_attn(i, heads, dt):
    from imagdressing_amd.unet import Attention
    sd = {"a.to_q.weight": i["wq"], "a.to_k.weight": i["wk"], "a.to_v.weight": i["wv"], "a.to_out.0.weight": i["wo"],
          "a.to_out.0.bias": i["bo"]}
    return Attention(sd, "a", heads, "cuda", dt)


def set_lora(proc, lw):
    with torch.no_grad():
        for nm in ("q", "k", "v", "out"):
            layer = getattr(proc, f"to_{nm}_lora")
            layer.down.weight.copy_(lw[nm][0]); layer.up.weight.copy_(lw[nm][1])


def check(got, ref, dt, what):
    e = (got.float().cpu() - ref).abs()
    bad = e > ATOL[dt] + RTOL[dt] * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, max abs err {e.max().item():.4g} (atol {ATOL[dt]}, rtol {RTOL[dt]}, ref max {ref.abs().max().item():.3g})"


@DT
@pytest.mark.parametrize("name", ["hybrid_small", "hybrid_small_lora", "hybrid_d40", "hybrid_d80", "hybrid_d160", "hybrid_d40_lora"])
@torch.no_grad()
def test_hybrid_processor_vs_reference_golden(golden_processors, name, dt):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.adapter import attention_processor as A
    c = golden_processors[name]
    if c["C"] // c["heads"] not in (40, 64, 80, 160):
        pytest.skip("head dim of this fixture is not an SD1.5 head dim (kernel supports 40/64/80/160)")
    i = hybrid_inputs(c)
    attn = make_attn(i, c["heads"], dt)
    pname = "blk.attn1.processor"
    if c["rank"]:
        proc = A.LoraRefSAttnProcessor2_0(pname, c["C"], rank=c["rank"], lora_scale=c["lora_scale"], scale=c["scale"])
        set_lora(proc, i["lora"])
    else:
        proc = A.RefSAttnProcessor2_0(pname, c["C"], scale=c["scale"])
    proc.to_k_ref.weight.copy_(i["wk_ref"]); proc.to_v_ref.weight.copy_(i["wv_ref"])
    attn.set_processor(proc)
    x = i["x"].cuda().to(dt)
    ref = i["ref"].cuda()
    cond = attn(x, **{"sa_hidden_states": {pname: ref}})
    check(cond, c["out_cond"], dt, f"{name} cond")
    unc = attn(x)
    check(unc, c["out_uncond"], dt, f"{name} uncond")
    if c["B"] == 2:     # per-row garment switch: row 0 cond, row 1 uncond in ONE call
        mixed = attn(x, sa_hidden_states={pname: ref}, sa_batch_mask=torch.tensor([1.0, 0.0], device="cuda"))
        check(mixed[0:1], c["out_cond"][0:1], dt, f"{name} mixed row 0")
        check(mixed[1:2], c["out_uncond"][1:2], dt, f"{name} mixed row 1")
    # fused residual == separate add
    res = torch.randn_like(x)
    fused = attn(x, residual=res, sa_hidden_states={pname: ref})
    check(fused, c["out_cond"] + res.float().cpu(), dt, f"{name} fused residual")


@DT
@pytest.mark.parametrize("name", ["cross_small", "cross_small_ip", "cross_d40", "cross_d160_ip"])
@torch.no_grad()
def test_cross_processor_vs_reference_golden(golden_processors, name, dt):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.adapter import attention_processor as A
    c = golden_processors[name]
    if c["C"] // c["heads"] not in (40, 64, 80, 160):
        pytest.skip("head dim of this fixture is not an SD1.5 head dim")
    i = cross_inputs(c)
    attn = make_attn(i, c["heads"], dt)
    if c["ip_tokens"]:
        proc = A.LoRAIPAttnProcessor2_0(c["C"], c["KD"], rank=c["rank"], lora_scale=c["lora_scale"], scale=c["ip_scale"],
                                        num_tokens=c["ip_tokens"])
        set_lora(proc, i["lora"])
        proc.to_k_ip.weight.copy_(i["wk_ip"]); proc.to_v_ip.weight.copy_(i["wv_ip"])
    else:
        proc = A.CAttnProcessor2_0("blk.attn2.processor", c["C"], c["KD"])
    attn.set_processor(proc)
    ehs = i["ehs"].cuda()
    out = attn(i["x"].cuda().to(dt), encoder_hidden_states=ehs, sa_hidden_states={"unused": None})
    check(out, c["out"], dt, name)
    out2 = attn(i["x"].cuda().to(dt), encoder_hidden_states=ehs)          # second call hits the cached K/V
    assert torch.equal(out, out2)


@DT
@torch.no_grad()
def test_cache_processor_stores_input(golden_processors, dt):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.adapter import attention_processor as A
    c = golden_processors["cache_small"]
    if c["C"] // c["heads"] not in (40, 64, 80, 160):
        pytest.skip("fixture head dim 8 is not an SD1.5 head dim")
    attn = make_attn(c, c["heads"], dt)
    p = A.CacheAttnProcessor2_0(); attn.set_processor(p)
    x = c["x"].cuda().to(dt)
    out = attn(x)
    assert p.cache["hidden_states"] is x
    check(out, c["out"], dt, "cache")


@DT
@pytest.mark.parametrize("name", ["resampler_real"])
@torch.no_grad()
def test_resampler_vs_reference_golden(golden_resampler, name, dt):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.adapter.resampler import Resampler
    c = golden_resampler[name]
    sd, x = resampler_inputs(c)
    m = Resampler(**c["cfg"])
    m.load_state_dict(sd, strict=True)
    out = m(x.cuda().to(dt))
    assert out.dtype == dt and tuple(out.shape) == tuple(c["out"].shape)
    check(out, c["out"], dt, name)


@DT
@torch.no_grad()
def test_proj_plus_vs_reference_golden(golden_resampler, dt):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd.adapter.resampler import ProjPlusModel
    c = golden_resampler["proj_plus_real"]
    sd, idv, clip = proj_plus_inputs(c)
    m = ProjPlusModel()
    m.load_state_dict(sd, strict=True)
    check(m(idv.cuda().to(dt), clip.cuda().to(dt)), c["out"], dt, "proj_plus")
    check(m(idv.cuda().to(dt), clip.cuda().to(dt), shortcut=True, scale=0.7), c["out_shortcut"], dt, "proj_plus shortcut")
