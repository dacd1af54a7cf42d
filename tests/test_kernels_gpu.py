"""Per-kernel parity of the HIP path (through the C ABI) against plain fp32 torch references of the
same op, on seeded inputs.  Tolerance: the north-star bar, atol 1e-2 on O(1) outputs computed in
bf16 with fp32 accumulation (plus a relative term for large-magnitude GEMM outputs)."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16
DTS = pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
TOL = {torch.bfloat16: 1e-2, torch.float16: 2e-3}      # fp16 carries 3 more mantissa bits


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd import ops as o
    return o


def dev(t):
    return t.to("cuda")


def rnd(seed, *shape, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def assert_close(got, ref, atol=None, rtol=None, what=""):
    base = TOL.get(got.dtype, 1e-2)
    atol = base if atol is None else atol
    rtol = base if rtol is None else rtol
    got = got.float().cpu()
    ref = ref.float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max abs err {err.max().item():.4g} "
            f"(ref max {ref.abs().max().item():.4g}); first bad index {idx}: got {got[tuple(idx)].item():.5g} "
            f"ref {ref[tuple(idx)].item():.5g}")


# ------------------------------------------------------------------------------------------
# linear / GEMM
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, -1])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 320, 320), (77, 64, 768), (8, 1280, 320), (130, 4, 72)])
@DTS
def test_linear(ops, cfg, M, N, K, dt):
    x = rnd(1, M, K).to(dt)
    w = rnd(2, N, K, scale=K ** -0.5).to(dt)      # asymmetric: catches transposes
    b = rnd(3, N)
    ref = x.float() @ w.float().t() + b
    out = ops.linear(dev(x), dev(w), dev(b), cfg=cfg)
    assert out.dtype == dt
    assert_close(out, ref, what=f"linear cfg={cfg}")


@DTS
def test_linear_epilogues(ops, dt):
    M, N, K = 200, 128, 64
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N)
    res = rnd(4, M, N).to(dt)
    base = x.float() @ w.float().t() + b
    assert_close(ops.linear(dev(x), dev(w), dev(b), res=dev(res)), base + res.float(), what="residual")
    assert_close(ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_SILU), F.silu(base), what="silu")
    out32 = ops.linear(dev(x), dev(w), dev(b), out_f32=True)
    assert out32.dtype == torch.float32
    assert_close(out32, base, atol=2e-3, rtol=2e-3, what="fp32 out")
    # GEGLU with interleaved (value, gate) rows
    g = ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_GEGLU)
    assert g.shape == (M, N // 2)
    assert_close(g, base[:, 0::2] * F.gelu(base[:, 1::2]), what="geglu")


@DTS
@pytest.mark.parametrize("M,N,K,split,cfg", [(512, 1280, 11520, 6, 0), (200, 320, 2304, 3, 2), (2048, 132, 4096, 4, 1), (64, 64, 8192, 8, -1), (300, 640, 2560, 5, 6), (130, 64, 4096, 7, 7), (600, 384, 2048, 3, 9), (520, 520, 1024, 2, 10),
                                                  (512, 1280, 5120, 3, 17), (300, 132, 2048, 4, 17), (512, 1280, 5120, 3, 25), (300, 132, 2048, 4, 27),
                                                  (512, 1280, 5120, 3, 30), (300, 132, 2048, 4, 31), (1000, 640, 2560, 2, 31), (4096, 640, 2560, 5, 30), (1000, 640, 2560, 2, 32), (700, 132, 2048, 4, 32)])
def test_linear_split_k(ops, M, N, K, split, cfg, dt):
    """K slices into fp32 slabs + fixed-order finish kernel == unsplit result (bias + residual + SiLU epilogue)."""
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N); res = rnd(4, M, N).to(dt)
    ref = F.silu(x.float() @ w.float().t() + b + res.float())
    out = ops.linear(dev(x), dev(w), dev(b), res=dev(res), act=ops.ACT_SILU, cfg=cfg, split_k=split)
    assert_close(out, ref, what=f"split-K {split}")
    one = ops.linear(dev(x), dev(w), dev(b), res=dev(res), act=ops.ACT_SILU, cfg=cfg, split_k=1)
    assert_close(out, one.float(), atol=2e-2 if dt == bf16 else 4e-3, what="split vs unsplit")
    assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), act=ops.ACT_SILU, cfg=cfg, split_k=split)), "not deterministic"
    # the in-kernel reduction (last-arriving workgroup of every tile) and the separate finish launch sum in the same order
    ops.SPLITK_IN_KERNEL = True
    try:
        fused = ops.linear(dev(x), dev(w), dev(b), res=dev(res), act=ops.ACT_SILU, cfg=cfg, split_k=split)
        fused2 = ops.linear(dev(x), dev(w), dev(b), res=dev(res), act=ops.ACT_SILU, cfg=cfg, split_k=split)
    finally:
        ops.SPLITK_IN_KERNEL = False
    assert torch.equal(out, fused) and torch.equal(out, fused2), "in-kernel split-K reduction differs from the finish kernel"
    assert int(ops.splitk_counters(out.device).abs().max()) == 0, "arrival counters not left at zero"


@pytest.mark.parametrize("M,N", [(128, 320), (300, 320), (4096 + 37, 320), (200, 64), (1000, 192), (129, 256)])
@DTS
def test_row_linear(ops, M, N, dt):
    """row-resident kernel (tile config 12: K = 320, rows in registers, weights through the LDS-DMA ring) == x W^T + b (+ res)"""
    K = 320
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N); res = rnd(4, M, N).to(dt)
    ref = x.float() @ w.float().t() + b
    assert_close(ops.linear(dev(x), dev(w), dev(b), cfg=12), ref, what="row linear")
    out = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=12)
    assert_close(out, ref + res.float(), what="row linear + residual")
    tiled = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=4)
    assert_close(out, tiled.float(), atol=2e-2 if dt == bf16 else 4e-3, what="row-resident vs tiled kernel")
    assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=12)), "not deterministic"


@pytest.mark.parametrize("M,N", [(128, 320), (333, 320), (2048 + 5, 320), (260, 128)])
@DTS
def test_row_linear_layernorm(ops, M, N, dt):
    """LayerNorm -> linear as ONE launch (affine folded into the weights) == F.layer_norm + F.linear in fp32"""
    K = 320
    x = (rnd(1, M, K) * 1.7 + 0.6 * rnd(5, 1, K)).to(dt)          # per-channel offsets: the mean matters
    w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N)
    g = 1.0 + 0.3 * rnd(6, K); be = 0.2 * rnd(7, K)
    ref = F.linear(F.layer_norm(x.float(), (K,), g, be, 1e-5), w.float(), b)
    w2, b2 = ops.fold_layernorm_affine(dev(w), dev(b), dev(g), dev(be))
    out = ops.linear(dev(x), w2, b2, ln_eps=1e-5)
    assert_close(out, ref, atol=3e-2 if dt == bf16 else None, what="LN + linear (fused)")
    two = ops.linear(ops.layer_norm(dev(x), dev(g), dev(be), 1e-5), dev(w), dev(b))
    assert_close(out, two.float(), atol=3e-2 if dt == bf16 else 5e-3, what="fused vs two launches")


@DTS
def test_row_linear_head_split(ops, dt):
    """head-split Q epilogue of the row-resident kernel (norm2 -> attn2.to_q of the 64x64 level: 8 heads x 40)"""
    B, HW, Cc, H, D = 2, 200, 320, 8, 40
    DPK, _ = ops.attn_padded_dims(D)
    x = rnd(1, B * HW, Cc).to(dt); w = rnd(2, Cc, Cc, scale=Cc ** -0.5).to(dt)
    g = 1.0 + 0.3 * rnd(6, Cc); be = 0.2 * rnd(7, Cc)
    q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda")
    w2, b2 = ops.fold_layernorm_affine(dev(w), None, dev(g), dev(be))
    heads = dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.5)])
    ops.conv_gemm(dev(x), w2, M=B * HW, N=Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, bias=b2, heads=heads, ln_eps=1e-5)
    ref = 0.5 * F.linear(F.layer_norm(x.float(), (Cc,), g, be, 1e-5), w.float())
    ref = ref.view(B, HW, H, D).permute(0, 2, 1, 3)
    assert_close(q[..., :D], ref, what="LN + to_q head split")
    assert float(q[..., D:].abs().max()) == 0.0
    with pytest.raises(ops.L.ImdError):
        ops.linear(dev(rnd(1, 64, 640).to(dt)), dev(rnd(2, 320, 640).to(dt)), cfg=12)       # K != 320: refused, no fallback


@pytest.mark.parametrize("M,ln", [(128, True), (300, True), (2048 + 77, True), (256, False)])
@DTS
def test_ff_geglu_fused(ops, M, ln, dt):
    """norm3 -> GEGLU feed-forward -> + residual as one launch == the fp32 restatement of diffusers' FeedForward(geglu)"""
    Cc, inner = 320, 1280
    x = (rnd(1, M, Cc) * 1.3 + 0.5 * rnd(5, 1, Cc)).to(dt)
    w1 = rnd(2, 2 * inner, Cc, scale=Cc ** -0.5).to(dt); b1 = 0.3 * rnd(3, 2 * inner)
    w2 = rnd(4, Cc, inner, scale=inner ** -0.5).to(dt); b2 = 0.3 * rnd(8, Cc)
    g = 1.0 + 0.3 * rnd(6, Cc); be = 0.2 * rnd(7, Cc)
    xf = x.float()
    n = F.layer_norm(xf, (Cc,), g, be, 1e-5) if ln else xf
    hid, gate = F.linear(n, w1.float(), b1).chunk(2, dim=-1)
    ref = xf + F.linear(hid * F.gelu(gate), w2.float(), b2)
    packed = ops.pack_ff_fused(dev(w1), dev(b1), dev(w2), dev(b2), dev(g) if ln else None, dev(be) if ln else None)
    out = ops.ff_geglu_fused(dev(x), packed, 1e-5)
    assert out.dtype == dt
    assert_close(out, ref, atol=4e-2 if dt == bf16 else 6e-3, rtol=2e-2 if dt == bf16 else 3e-3, what="fused feed-forward")
    # against the three-launch path of the engine (interleaved GEGLU rows)
    wi = torch.stack([w1[:inner], w1[inner:]], dim=1).reshape(2 * inner, Cc); bi = torch.stack([b1[:inner], b1[inner:]], dim=1).reshape(-1)
    nn_ = ops.layer_norm(dev(x), dev(g), dev(be), 1e-5) if ln else dev(x)
    gg = ops.linear(nn_, dev(wi), dev(bi), act=ops.ACT_GEGLU)
    three = ops.linear(gg, dev(w2), dev(b2), res=dev(x))
    assert_close(out, three.float(), atol=5e-2 if dt == bf16 else 8e-3, rtol=2e-2 if dt == bf16 else 3e-3, what="fused vs three launches")
    assert torch.equal(out, ops.ff_geglu_fused(dev(x), packed, 1e-5)), "not deterministic"


@pytest.mark.parametrize("M,N", [(128, 640), (300, 640), (2048 + 37, 640), (200, 160), (1024, 320), (129, 1280)])
@DTS
def test_row_linear_k640(ops, M, N, dt):
    """split-K row-resident kernel (tile config 13: K = 640, two waves share a token block, partial sums swapped through LDS)"""
    K = 640
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N); res = rnd(4, M, N).to(dt)
    ref = x.float() @ w.float().t() + b
    assert_close(ops.linear(dev(x), dev(w), dev(b), cfg=13), ref, what="row linear k640")
    out = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=13)
    assert_close(out, ref + res.float(), what="row linear k640 + residual")
    tiled = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=2)
    assert_close(out, tiled.float(), atol=2e-2 if dt == bf16 else 4e-3, what="row-resident vs tiled kernel")
    assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=13)), "not deterministic"


@DTS
def test_row_linear_k640_layernorm_head_split(ops, dt):
    """norm2 -> attn2.to_q of the 32x32 level (8 heads x 80) as one launch, head-split Q epilogue; and the row-major LN + linear"""
    B, HW, Cc, H, D = 2, 300, 640, 8, 80
    DPK, _ = ops.attn_padded_dims(D)
    x = (rnd(1, B * HW, Cc) * 1.5 + 0.6 * rnd(5, 1, Cc)).to(dt); w = rnd(2, Cc, Cc, scale=Cc ** -0.5).to(dt); b = rnd(3, Cc)
    g = 1.0 + 0.3 * rnd(6, Cc); be = 0.2 * rnd(7, Cc)
    nref = F.layer_norm(x.float(), (Cc,), g, be, 1e-5)
    w2, b2 = ops.fold_layernorm_affine(dev(w), dev(b), dev(g), dev(be))
    out = ops.linear(dev(x), w2, b2, ln_eps=1e-5)
    assert_close(out, F.linear(nref, w.float(), b), atol=3e-2 if dt == bf16 else None, what="LN + linear k640")
    q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda")
    w3, b3 = ops.fold_layernorm_affine(dev(w), None, dev(g), dev(be))
    ops.conv_gemm(dev(x), w3, M=B * HW, N=Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, bias=b3, ln_eps=1e-5,
                  heads=dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.25)]))
    ref = (0.25 * F.linear(nref, w.float())).view(B, HW, H, D).permute(0, 2, 1, 3)
    assert_close(q[..., :D], ref, atol=2e-2 if dt == bf16 else None, what="LN + to_q head split k640")


@pytest.mark.parametrize("M,N", [(64, 1280), (300, 1280), (2048 + 37, 1280), (200, 160), (512, 640)])
@DTS
def test_row_linear_k1280(ops, M, N, dt):
    """4-way split-K row-resident kernel on v_mfma_f32_16x16x32 (tile config 14: K = 1280)"""
    K = 1280
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N); res = rnd(4, M, N).to(dt)
    ref = x.float() @ w.float().t() + b
    assert_close(ops.linear(dev(x), dev(w), dev(b), cfg=14), ref, what="row linear k1280")
    out = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=14)
    assert_close(out, ref + res.float(), what="row linear k1280 + residual")
    tiled = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=2)
    assert_close(out, tiled.float(), atol=2e-2 if dt == bf16 else 4e-3, what="row-resident vs tiled kernel")
    assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=14)), "not deterministic"


@DTS
def test_row_linear_k1280_layernorm_head_split(ops, dt):
    """norm2 -> attn2.to_q of the 16x16 level (8 heads x 160) as one launch, head-split Q epilogue; and the row-major LN + linear"""
    B, HW, Cc, H, D = 2, 200, 1280, 8, 160
    DPK, _ = ops.attn_padded_dims(D)
    x = (rnd(1, B * HW, Cc) * 1.5 + 0.6 * rnd(5, 1, Cc)).to(dt); w = rnd(2, Cc, Cc, scale=Cc ** -0.5).to(dt); b = rnd(3, Cc)
    g = 1.0 + 0.3 * rnd(6, Cc); be = 0.2 * rnd(7, Cc)
    nref = F.layer_norm(x.float(), (Cc,), g, be, 1e-5)
    w2, b2 = ops.fold_layernorm_affine(dev(w), dev(b), dev(g), dev(be))
    out = ops.linear(dev(x), w2, b2, ln_eps=1e-5)
    assert_close(out, F.linear(nref, w.float(), b), atol=3e-2 if dt == bf16 else None, what="LN + linear k1280")
    q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda")
    w3, b3 = ops.fold_layernorm_affine(dev(w), None, dev(g), dev(be))
    ops.conv_gemm(dev(x), w3, M=B * HW, N=Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, bias=b3, ln_eps=1e-5,
                  heads=dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.25)]))
    ref = (0.25 * F.linear(nref, w.float())).view(B, HW, H, D).permute(0, 2, 1, 3)
    assert_close(q[..., :D], ref, atol=2e-2 if dt == bf16 else None, what="LN + to_q head split k1280")


@pytest.mark.parametrize("B,H,W,Cc,G,producer,cfg", [
    (8, 64, 64, 320, 32, "patch", 12),     # 64x64 level of the bench batch: statistics from the halo-patch conv's epilogue (96 partials per image), row_linear.hip
    (3, 32, 32, 640, 32, "patch", 13),     # 32x32 level, row_linear_k640.hip
    (8, 16, 16, 1280, 32, "finish", 14),   # 16x16 level: statistics from the split-K finish launch, row_linear_k1280.hip
    (2, 16, 16, 320, 32, "pass", 12),      # statistics from an ordinary statistics pass (gn_stats_kernel's chunking), 256 rows per image
    (3, 8, 8, 1280, 32, "pass", 14),       # 8x8 maps: 64 rows per image = exactly one row block of the K = 1280 kernel
    (2, 8, 8, 640, 32, "pass", -1),        # 64 rows per image: not a multiple of the K = 640 kernel's 128-row block -> the wrapper runs the two launches
])
@pytest.mark.parametrize("silu", [False, True])
@DTS
def test_row_linear_groupnorm_of_the_input(ops, B, H, W, Cc, G, producer, cfg, silu, dt):
    """imd_conv_gemm_params.gn_in_* (ABI v9): Transformer2DModel.norm -> proj_in as ONE launch of the row-resident projection kernels -- every workgroup
    folds the statistic partials of its image in imd_groupnorm's own order and normalises its rows in registers.  BIT-IDENTICAL to imd_groupnorm followed
    by the same projection (same coefficients, same rounding of the normalised value to the element type), whoever produced the statistics: a halo-patch
    conv's epilogue, a split-K finish launch, or a statistics pass.  Shapes the kernels do not take fall back to the two launches inside the wrapper."""
    from imagdressing_amd import ops as ops_mod
    assert ops_mod.FUSED_GN_PROJ
    gamma = 1.0 + 0.2 * rnd(6, Cc); beta = 0.2 * rnd(7, Cc)
    w = rnd(8, Cc, Cc, scale=Cc ** -0.5).to(dt); b = rnd(9, Cc)
    if producer == "pass":
        x = dev((rnd(1, B, H, W, Cc) * 1.5 + 0.3).to(dt))
        # a statistics pass writes its partials into the shared workspace: hand them over the way a producer would
        a_, b_ = ops.group_norm_coeffs(x, dev(gamma), dev(beta), groups=G, eps=1e-6)       # (runs gn_stats_kernel: partials now sit in the workspace)
        lib = ops.L.load()
        nws = lib.imd_groupnorm_workspace_floats(B, H * W, Cc, G)
        part = ops.workspace("gn_partial", (max(nws, 1),), torch.float32, x.device)
        # chunks per image as norm.hip computes them: the workspace holds [B][nchunks][G][2] partials followed by 2 B C coefficient floats
        nchunks = (nws - 2 * B * Cc) // (B * G * 2)
        assert nchunks >= 1 and nchunks * B * G * 2 + 2 * B * Cc == nws
        x._imd_gn_stats = (part[: B * nchunks * G * 2].view(B, nchunks, G, 2).clone(), nchunks, G)
    else:
        xin = dev(rnd(1, B, H, W, Cc).to(dt))
        wc = rnd(2, Cc, Cc, 3, 3, scale=(9 * Cc) ** -0.5).to(dt)
        kw = dict(cfg=5, split_k=1) if producer == "patch" else dict(cfg=5, split_k=4)     # (cfg of the PRODUCING conv)
        x = ops.conv2d_nhwc(xin, dev(pack_conv(wc)), dev(rnd(3, Cc)), gn_stats_groups=G, **kw)
        assert getattr(x, "_imd_gn_stats", None) is not None
    two = ops.conv2d_nhwc(ops.group_norm(x, dev(gamma), dev(beta), groups=G, eps=1e-6, silu=silu), dev(w), dev(b), taps=1)
    seen = []
    hook = ops_mod.GEMM_EVENT_HOOK
    ops_mod.GEMM_EVENT_HOOK = {}
    try:
        one = ops.conv2d_nhwc(x, dev(w), dev(b), taps=1, cfg=cfg, gn_in=(dev(gamma), dev(beta), 1e-6, silu, G))
        seen = list(ops_mod.GEMM_EVENT_HOOK)
    finally:
        ops_mod.GEMM_EVENT_HOOK = hook
    if cfg >= 0:
        assert len(seen) == 1 and seen[0][1] == cfg, f"expected one row-resident launch, got {seen}"
        two = ops.conv2d_nhwc(ops.group_norm(x, dev(gamma), dev(beta), groups=G, eps=1e-6, silu=silu), dev(w), dev(b), taps=1, cfg=cfg)
    assert torch.equal(one, two), (one.float() - two.float()).abs().max().item()
    ref = F.group_norm(x.float().cpu().permute(0, 3, 1, 2), G, gamma, beta, eps=1e-6)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1) @ w.float().t() + b
    assert_close(one, ref, atol=4 * TOL[dt], what="GroupNorm inside the projection vs fp32")
    ops_mod.FUSED_GN_PROJ = False               # the A/B switch restores the two launches (same values)
    assert torch.equal(ops.conv2d_nhwc(x, dev(w), dev(b), taps=1, cfg=cfg, gn_in=(dev(gamma), dev(beta), 1e-6, silu, G)), two)


@pytest.mark.parametrize("ln", [False, True])
@DTS
def test_row_qkv(ops, ln, dt):
    """norm1 -> to_q / to_k / to_v of a 320-channel block as ONE launch (row_qkv.hip) == LayerNorm + linear in fp32, in the
    attention kernel's layouts (Q / K [B, H, N, 48] with Q pre-scaled, V^T [B, H, 64, pad64(N)]); also against the tiled kernel"""
    B, HW, Cc, H, D = 2, 384, 320, 8, 40
    DPK, DPV = ops.attn_padded_dims(D)
    LP = ops.pad64(HW)
    x = (rnd(1, B * HW, Cc) * 1.4 + 0.5 * rnd(5, 1, Cc)).to(dt); w = rnd(2, 3 * Cc, Cc, scale=Cc ** -0.5).to(dt)
    g = 1.0 + 0.3 * rnd(6, Cc); be = 0.2 * rnd(7, Cc)
    def bufs():
        return (torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda"), torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda"),
                torch.zeros(B, H, DPV, LP, dtype=dt, device="cuda"))
    q, k, vt = bufs()
    heads = dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.3), (k, 0, DPK, HW, 1.0), (vt, 1, DPV, LP, 1.0)])
    if ln:
        w2, b2 = ops.fold_layernorm_affine(dev(w), None, dev(g), dev(be))
        ops.conv_gemm(dev(x), w2, M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, bias=b2, heads=heads, ln_eps=1e-5)
        n = F.layer_norm(x.float(), (Cc,), g, be, 1e-5)
    else:
        ops.conv_gemm(dev(x), dev(w), M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, heads=heads, cfg=15)
        n = x.float()
    ref = F.linear(n, w.float()).view(B, HW, 3, H, D)
    tol = dict(atol=3e-2 if dt == bf16 else 4e-3)
    assert_close(q[..., :D], 0.3 * ref[:, :, 0].permute(0, 2, 1, 3), what="Q", **tol)
    assert_close(k[..., :D], ref[:, :, 1].permute(0, 2, 1, 3), what="K", **tol)
    assert_close(vt[:, :, :D, :HW], ref[:, :, 2].permute(0, 2, 3, 1), what="V^T", **tol)
    assert float(q[..., D:].abs().max()) == 0.0 and float(vt[:, :, D:].abs().max()) == 0.0      # padding untouched
    if not ln:
        q2, k2, vt2 = bufs()
        heads2 = dict(C=Cc, H=H, D=D, dests=[(q2, 0, DPK, HW, 0.3), (k2, 0, DPK, HW, 1.0), (vt2, 1, DPV, LP, 1.0)])
        ops.conv_gemm(dev(x), dev(w), M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, heads=heads2, cfg=4)
        assert_close(q, q2.float(), atol=1e-2 if dt == bf16 else 2e-3, what="Q vs tiled")
        assert_close(vt, vt2.float(), atol=2e-2 if dt == bf16 else 4e-3, what="V^T vs tiled")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 320, 320), (1000, 2560, 640), (77, 64, 768), (520, 1284, 1280), (2048 + 19, 520, 128)])
@pytest.mark.parametrize("cfg", [16, 17, 19, 25, 27, 30, 31, 32])
@DTS
def test_linear_gemm_dma(ops, M, N, K, dt, cfg):
    """tile configs 16 (256 x 256 x 64, two stages), 17 and 19 (128 x 128 x 32, three- / four-stage ring, round 3), 25 and 27 (128 x 128 x 64:
    128-byte rows, two / three stages, round 4), 30 / 31 (gemm_dma256.hip, round 5: 256 x 128 tiles, producer / consumer waves, persistent item loop,
    epilogue from registers), 32 (round 6: the same kernel with 192-row tiles, six consumer + four producer waves): both operands by LDS-DMA == x W^T + b, with the shared epilogues"""
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N); res = rnd(4, M, N).to(dt)
    base = x.float() @ w.float().t() + b
    assert_close(ops.linear(dev(x), dev(w), dev(b), cfg=cfg), base, what="gemm_dma")
    assert_close(ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=cfg), base + res.float(), what="gemm_dma + residual")
    if N % 8 == 0:
        gg = ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_GEGLU, cfg=cfg)
        assert_close(gg, base[:, 0::2] * F.gelu(base[:, 1::2]), what="gemm_dma geglu")
    one = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=cfg)
    assert_close(one, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=0).float(), atol=2e-2 if dt == bf16 else 4e-3, what="gemm_dma vs tiled")
    assert torch.equal(one, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=cfg)), "not deterministic"
    with pytest.raises(ops.L.ImdError):
        ops.linear(dev(rnd(1, 64, 72).to(dt)), dev(rnd(2, 64, 72).to(dt)), cfg=cfg)           # K % 64 != 0: refused


@pytest.mark.parametrize("cfg", [16, 17, 19, 25, 27, 30, 31, 32])
@DTS
def test_gemm_dma_head_split(ops, dt, cfg):
    """head-split q / k / v epilogue through tile config 16 (the 32x32-level projection: 8 heads x 80)"""
    B, HW, Cc, H, D = 2, 300, 640, 8, 80
    DPK, DPV = ops.attn_padded_dims(D); LP = ops.pad64(HW)
    x = rnd(1, B * HW, Cc).to(dt); w = rnd(2, 3 * Cc, Cc, scale=Cc ** -0.5).to(dt)
    q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda"); k = torch.zeros_like(q); vt = torch.zeros(B, H, DPV, LP, dtype=dt, device="cuda")
    ops.conv_gemm(dev(x), dev(w), M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, cfg=cfg,
                  heads=dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.3), (k, 0, DPK, HW, 1.0), (vt, 1, DPV, LP, 1.0)]))
    ref = F.linear(x.float(), w.float()).view(B, HW, 3, H, D)
    assert_close(q[..., :D], 0.3 * ref[:, :, 0].permute(0, 2, 1, 3), what="Q")
    assert_close(k[..., :D], ref[:, :, 1].permute(0, 2, 1, 3), what="K")
    assert_close(vt[:, :, :D, :HW], ref[:, :, 2].permute(0, 2, 3, 1), what="V^T")


@pytest.mark.parametrize("M,N,K,act,split", [(8192, 5120, 640, "geglu", 1), (2048, 10240, 1280, "geglu", 1), (8192, 640, 2560, "res", 1),
                                             (8192, 640, 2560, "res", 2), (8192 + 70, 1920, 640, "heads", 1)])
@pytest.mark.parametrize("cfg", [30, 31, 32])
@DTS
def test_gemm_dma256_persistent_item_loop(ops, M, N, K, act, split, cfg, dt):
    """gemm_dma256.hip at the feed-forward shapes it was written for (BASELINE configs[1], batch 4: GEGLU 8192 x 5120 x 640 and
    2048 x 10240 x 1280, FF-out 8192 x 640 x 2560 with and without K slices, the 32x32-level q/k/v projection with a ragged row count):
    640 ... 1280 (tile, slice) items walked by 256 persistent workgroups -- several items per workgroup, the operand ring running across item
    boundaries, results leaving from registers.  Against fp32 torch, and BIT-identical to the 128 x 128 LDS-DMA tiles (tile config 25: same
    MFMA instruction, same ascending K order, same epilogue arithmetic) wherever no K slices are involved."""
    x = rnd(1, M, K).to(dt); w = rnd(2, N, K, scale=K ** -0.5).to(dt); b = rnd(3, N)
    base = x.float() @ w.float().t() + b
    if act == "geglu":
        out = ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_GEGLU, cfg=cfg)
        assert_close(out, base[:, 0::2] * F.gelu(base[:, 1::2]), what="geglu")
        assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), act=ops.ACT_GEGLU, cfg=25)), "differs from the 128 x 128 tiles"
    elif act == "res":
        res = rnd(4, M, N).to(dt)
        out = ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=cfg, split_k=split)
        assert_close(out, base + res.float(), what="residual")
        if split == 1:
            assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=25, split_k=1)), "differs from the 128 x 128 tiles"
        else:
            assert torch.equal(out, ops.linear(dev(x), dev(w), dev(b), res=dev(res), cfg=25, split_k=split)), "K slices differ from the 128 x 128 tiles'"
    else:
        B, H, D = 2, 8, 80
        HW, Cc = M // B, N // 3
        DPK, DPV = ops.attn_padded_dims(D); LP = ops.pad64(HW)
        outs = []
        for c in (cfg, 25):
            q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda"); k = torch.zeros_like(q); vt = torch.zeros(B, H, DPV, LP, dtype=dt, device="cuda")
            ops.conv_gemm(dev(x), dev(w), M=M, N=N, Cin=K, Hin=HW, Win=1, Hout=HW, Wout=1, cfg=c,
                          heads=dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.3), (k, 0, DPK, HW, 1.0), (vt, 1, DPV, LP, 1.0)]))
            outs.append((q, k, vt))
        ref = F.linear(x.float(), w.float()).view(B, HW, 3, H, D)
        assert_close(outs[0][0][..., :D], 0.3 * ref[:, :, 0].permute(0, 2, 1, 3), what="Q")
        assert_close(outs[0][2][:, :, :D, :HW], ref[:, :, 2].permute(0, 2, 3, 1), what="V^T")
        for a, b2 in zip(outs[0], outs[1]):
            assert torch.equal(a, b2), "head-split output differs from the 128 x 128 tiles"
    assert torch.isfinite(out.float()).all() if act != "heads" else True


@DTS
def test_conv_auto_split_small_m(ops, dt):
    """the 8x8 ResNet conv shape (M = 512, K = 11520) takes the automatic split-K path"""
    B, H, W, Cin, Cout = 8, 8, 8, 1280, 1280
    assert ops.L.load().imd_conv_gemm_auto_split(B * H * W, Cout, 9 * Cin, -1) > 1
    x = rnd(1, B, Cin, H, W).to(dt); w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt); b = rnd(3, Cout)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1)
    out = ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b))
    assert_close(out, ref, what="auto split conv")


@DTS
def test_linear_no_bias_large_k(ops, dt):
    M, N, K = 512, 1280, 11520
    x = rnd(5, M, K).to(dt); w = rnd(6, N, K, scale=K ** -0.5).to(dt)
    ref = x.float() @ w.float().t()
    assert_close(ops.linear(dev(x), dev(w)), ref, what="large K")


# ------------------------------------------------------------------------------------------
# convolution (implicit GEMM) vs F.conv2d
# ------------------------------------------------------------------------------------------
def pack_conv(w):  # [Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 18])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [
    (2, 16, 16, 64, 64, 1, False), (1, 12, 20, 32, 320, 1, False), (2, 16, 16, 64, 128, 2, False),
    (1, 8, 8, 64, 64, 1, True), (1, 9, 7, 8, 320, 1, False), (3, 6, 6, 320, 4, 1, False)])
@DTS
def test_conv3x3(ops, cfg, B, H, W, Cin, Cout, stride, ups, dt):
    if cfg == 18 and Cin % 32:          # the gathering LDS-DMA form moves 32-channel pieces
        pytest.skip("tile config 18 needs Cin % 32 == 0")
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout)
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1).permute(0, 2, 3, 1)
    out = ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b), taps=9, stride=stride,
                          ups=ups, cfg=cfg)
    assert_close(out, ref, what=f"conv3x3 cfg={cfg}")


@pytest.mark.parametrize("B,H,W,Cin,Cout,split", [
    (2, 16, 16, 64, 64, 1), (1, 8, 32, 32, 320, 1), (3, 24, 16, 96, 132, 1), (1, 32, 32, 320, 320, 1), (2, 16, 16, 640, 256, 4),
    (1, 8, 16, 1280, 128, 8), (2, 16, 16, 640, 320, 2), (1, 16, 32, 64, 192, 1),      # (N = 320 / 192: a narrow last channel tile, 4 x 1 waves)
    # ragged maps: tiles hang over the right / bottom edge (the 96 x 72 latent of BASELINE configs[4] and its 48 x 36 level)
    (1, 96, 72, 320, 320, 1), (2, 48, 36, 640, 640, 2), (1, 12, 18, 64, 64, 1), (2, 9, 17, 32, 40, 1)])
@pytest.mark.parametrize("reg_staged", [False, True], ids=["dma", "regs"])
@DTS
def test_conv3x3_halo_patch(ops, B, H, W, Cin, Cout, split, dt, reg_staged):
    """tile config 5 (LDS-resident halo patch) == F.conv2d, with the full epilogue and with K slices; both staging paths: every
    operand by LDS-DMA with source-side swizzles (round 3, default) and through registers (tuning knob 2, bit 9)"""
    lib = ops.L.load()
    ops.L.check(lib.imd_set_tuning(2, 23 | (512 if reg_staged else 0)))
    try:
        _halo_patch_case(ops, B, H, W, Cin, Cout, split, dt)
    finally:
        ops.L.check(lib.imd_set_tuning(2, 23))


def _halo_patch_case(ops, B, H, W, Cin, Cout, split, dt):
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, H, W, Cout).to(dt)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1) + temb[:, None, None, :] + res.float()
    out = ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b), rowvec=dev(temb), rowvec_stride=Cout,
                          res=dev(res), cfg=5, split_k=split)
    assert_close(out, ref, what=f"halo-patch conv split={split}")
    if split > 1:        # in-kernel reduction by the last-arriving workgroup == separate finish launch, bit for bit; counters left at zero
        ops.SPLITK_IN_KERNEL = True
        try:
            two = ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b), rowvec=dev(temb), rowvec_stride=Cout,
                                  res=dev(res), cfg=5, split_k=split)
        finally:
            ops.SPLITK_IN_KERNEL = False
        assert torch.equal(out, two)
        assert int(ops.splitk_counters(out.device).abs().max()) == 0
    with pytest.raises(ops.L.ImdError):
        ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b), stride=2, cfg=5)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,split", [(8, 8, 8, 1280, 1280, 1, 6), (8, 16, 16, 640, 128, 2, 3), (2, 8, 8, 2560, 132, 1, 8),
                                                       (1, 17, 9, 64, 64, 2, 1)])
@pytest.mark.parametrize("cfg", [18, 20, 26, 28])
@DTS
def test_conv3x3_dma_gather_split_k(ops, B, H, W, Cin, Cout, stride, split, dt, cfg):
    """tile configs 18 / 20 (3x3 conv gathered tile by tile into the three- / four-stage LDS-DMA ring of gemm_dma.hip; 26 / 28: the 128-byte-row
    form, one tap x 64 channels per tile) on the maps the halo-patch kernel cannot tile -- 8 x 8, stride 2, odd sizes -- with K slices and the
    full epilogue"""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout)
    ref = F.conv2d(x.float(), w.float(), b, padding=1, stride=stride).permute(0, 2, 3, 1)
    res = rnd(5, *ref.shape).to(dt)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), stride=stride, res=dev(res), cfg=cfg, split_k=split)
    assert_close(out, ref + res.float(), what=f"conv_dma split={split}")
    assert torch.equal(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), stride=stride, res=dev(res), cfg=cfg, split_k=split)), "not deterministic"


@pytest.mark.parametrize("B,H,W,Cin,Cout,split,ups", [
    (1, 32, 32, 320, 320, 1, False),      # five 64-channel chunks (odd: the two-slot weight ring ends on the other slot), N = 320: half-empty last channel tile
    (2, 16, 16, 640, 256, 2, False), (1, 8, 16, 1280, 128, 4, False), (1, 96, 72, 320, 320, 1, False), (2, 9, 17, 64, 40, 1, False),
    (1, 16, 16, 128, 136, 1, True), (1, 24, 16, 192, 132, 3, False)])
@DTS
def test_conv3x3_halo_patch_128_byte_rows(ops, B, H, W, Cin, Cout, split, ups, dt):
    """tile config 29 (conv_patch.hip with 64-channel chunks: 128-byte rows, two-slot weight ring, two workgroups per CU) == F.conv2d with the full
    epilogue, K slices, ragged maps and the fused upsample; close to the 32-channel-chunk kernel (the K order inside a tap differs), the GroupNorm
    statistics of the un-split epilogue included"""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, Ho, Wo, Cout).to(dt)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b, padding=1).permute(0, 2, 3, 1) + temb[:, None, None, :] + res.float()
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    kw = dict(rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), ups=ups, split_k=split)
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=29, **kw)
    assert tuple(out.shape) == (B, Ho, Wo, Cout)
    assert_close(out, ref, what=f"halo-patch conv, 128-byte rows, split={split} ups={ups}")
    assert_close(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=5, **kw).float(), atol=2e-2 if dt == bf16 else 4e-3, what="vs 64-byte rows")
    assert torch.equal(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=29, **kw)), "not deterministic"
    if Cout % 32 == 0 and Cout // 32 >= 8:
        st_out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=29, gn_stats_groups=32, **kw)
        assert torch.equal(st_out, out)
        st = getattr(st_out, "_imd_gn_stats", None)
        assert st is not None
        folded = st[0].double().sum(1).cpu()
        o = out.double().cpu().permute(0, 3, 1, 2).reshape(B, 32, -1)
        assert torch.allclose(folded[..., 0] / o.shape[-1], o.mean(-1), atol=1e-4)
    if Cin % 64:
        pytest.fail("unreachable: every case has Cin % 64 == 0")


def test_conv3x3_halo_patch_128_byte_rows_refuses_other_chunks(ops):
    x = rnd(1, 1, 96, 8, 16).to(bf16); w = rnd(2, 64, 96, 3, 3).to(bf16)
    with pytest.raises(ops.L.ImdError):
        ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), None, cfg=29)        # Cin % 64 != 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,split", [(1, 8, 8, 64, 64, 1), (2, 16, 16, 640, 640, 2), (1, 12, 18, 32, 320, 1), (2, 32, 32, 320, 132, 1)])
@DTS
def test_conv3x3_halo_patch_fused_upsample(ops, B, H, W, Cin, Cout, split, dt):
    """Upsample2D (nearest 2x -> conv3x3) on the halo-patch kernel: the upsample is a source-pixel map of the LDS-DMA staging
    (logical pixel >> 1), the zero halo applies to the upsampled map; ragged tiles, K slices; equals the gather kernel's fused form"""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b, padding=1).permute(0, 2, 3, 1)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), ups=True, cfg=5, split_k=split)
    assert tuple(out.shape) == (B, 2 * H, 2 * W, Cout)
    assert_close(out, ref, what="halo-patch conv with fused nearest-2x upsample")
    gather = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), ups=True, cfg=0, split_k=1)
    assert_close(out, gather.float(), what="halo-patch vs gather kernel (fused upsample)")
    lib = ops.L.load()
    ops.L.check(lib.imd_set_tuning(2, 23 | 512))          # the register-staged form has no upsample path: refused, not wrong
    try:
        with pytest.raises(ops.L.ImdError):
            ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), ups=True, cfg=5, split_k=1)
    finally:
        ops.L.check(lib.imd_set_tuning(2, 23))


@pytest.mark.parametrize("B,H,W,Cin,Cout,split,ups", [
    (2, 16, 16, 64, 64, 1, False), (1, 32, 32, 320, 320, 1, False), (1, 64, 64, 320, 320, 1, False), (3, 24, 16, 96, 132, 1, False),
    (2, 16, 16, 640, 256, 4, False), (1, 16, 32, 1280, 128, 8, False), (2, 16, 16, 640, 320, 2, False),
    # ragged maps (tiles hang over the right / bottom edge) and the fused nearest-2x upsample (H, W = the STORED map)
    (1, 96, 72, 320, 320, 1, False), (2, 48, 36, 640, 640, 2, False), (1, 17, 19, 32, 40, 1, False),
    (1, 8, 8, 64, 64, 1, True), (2, 16, 16, 640, 640, 2, True), (1, 12, 18, 32, 320, 1, True)])
@DTS
def test_conv3x3_halo_patch_256_pixel_tiles(ops, B, H, W, Cin, Cout, split, ups, dt):
    """tile config 21 (conv_patch2.hip: 16 x 16 pixel tiles, wave tiles 128 pixels x 64 channels) == F.conv2d with the full epilogue,
    K slices, ragged maps and the fused upsample; bit-identical to the 8 x 16 tile kernel (same K order per output element)"""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, Ho, Wo, Cout).to(dt)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b, padding=1).permute(0, 2, 3, 1) + temb[:, None, None, :] + res.float()
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    kw = dict(rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), ups=ups, split_k=split)
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=21, **kw)
    assert tuple(out.shape) == (B, Ho, Wo, Cout)
    assert_close(out, ref, what=f"256-pixel halo-patch conv split={split} ups={ups}")
    assert torch.equal(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=5, **kw)), "differs from the 8 x 16 tile kernel"
    with pytest.raises(ops.L.ImdError):
        ops.conv2d_nhwc(dev(rnd(1, 1, 8, 16, 32).to(dt)), dev(rnd(2, 32, 288).to(dt)), None, cfg=21)        # H < 16


@pytest.mark.parametrize("B,H,W,Cin,Cout,split,ups", [
    (1, 64, 64, 320, 320, 1, False), (2, 32, 32, 640, 320, 1, False), (2, 16, 16, 64, 160, 1, False), (3, 8, 16, 96, 132, 1, False),
    (2, 16, 16, 640, 640, 4, False), (1, 16, 32, 1280, 1280, 8, False), (2, 16, 16, 960, 320, 3, False), (1, 8, 16, 32, 36, 1, False),
    # ragged maps (tiles hang over the right / bottom edge) and the fused nearest-2x upsample (H, W = the STORED map)
    (1, 96, 72, 320, 320, 1, False), (2, 48, 36, 640, 640, 2, False), (1, 17, 19, 32, 40, 1, False),
    (1, 8, 8, 64, 320, 1, True), (2, 16, 16, 640, 640, 2, True), (1, 12, 18, 32, 200, 1, True)])
@pytest.mark.parametrize("cfg", [22, 23])
@DTS
def test_conv3x3_halo_patch_160_channel_tiles(ops, B, H, W, Cin, Cout, split, ups, cfg, dt):
    """tile configs 22 / 23 (conv_patch3.hip: 8 x 16 | 16 x 16 pixels x 160 channels per workgroup, wave tiles 32 x 160) == F.conv2d with the full
    epilogue, K slices, ragged maps, channel counts that are not multiples of 160 and the fused upsample; bit-identical to the
    128-channel-tile kernel (same K order and MFMA sequence per output element)"""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, Ho, Wo, Cout).to(dt)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), b, padding=1).permute(0, 2, 3, 1) + temb[:, None, None, :] + res.float()
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    kw = dict(rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), ups=ups, split_k=split)
    if cfg == 23 and Ho < 16:
        with pytest.raises(ops.L.ImdError):
            ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=23, **kw)
        return
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=cfg, **kw)
    assert tuple(out.shape) == (B, Ho, Wo, Cout)
    assert_close(out, ref, what=f"160-channel halo-patch conv cfg={cfg} split={split} ups={ups}")
    assert torch.equal(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=5, **kw)), "differs from the 128-channel tile kernel"
    if split > 1:         # K slices: the finish launch can emit the GroupNorm statistics for this tile config as well
        G = 4 if Cout % 32 else 32
        if Cout % G == 0 and Cout // G >= 8:
            st_out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=cfg, gn_stats_groups=G, **kw)
            assert torch.equal(st_out, out) and getattr(st_out, "_imd_gn_stats", None) is not None
    with pytest.raises(ops.L.ImdError):
        ops.conv2d_nhwc(dev(rnd(1, 1, 8, 8, 32).to(dt)), dev(rnd(2, 32, 288).to(dt)), None, cfg=cfg)        # W < 16


@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (2, 32, 32, 320, 320), (1, 16, 32, 960, 640)])
@DTS
def test_conv3x3_fused_groupnorm(ops, B, H, W, Cin, Cout, silu, dt):
    """GroupNorm(+SiLU) fused into the conv's patch staging == GroupNorm -> SiLU -> conv (zero padding AFTER the activation)"""
    x = (rnd(1, B, Cin, H, W) * 1.5 + 0.3).to(dt)
    gamma = 1.0 + 0.2 * rnd(6, Cin); beta = 0.2 * rnd(7, Cin)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt); b = rnd(3, Cout)
    G = 32 if Cin >= 256 else 8
    h = F.group_norm(x.float(), G, gamma, beta, eps=1e-5)
    if silu:
        h = F.silu(h)
    ref = F.conv2d(h.to(dt).float(), w.float(), b, padding=1).permute(0, 2, 3, 1)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    ca, cb = ops.group_norm_coeffs(xd, dev(gamma), dev(beta), groups=G, eps=1e-5)
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), gn=(ca, cb, silu))
    assert_close(out, ref, atol=2 * TOL[dt], what="fused GN conv")
    # and against the unfused HIP path (same rounding points)
    two = ops.conv2d_nhwc(ops.group_norm(xd, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=silu), dev(pack_conv(w)), dev(b), cfg=0)
    assert_close(out, two.float(), atol=2 * TOL[dt], what="fused vs unfused")


@pytest.mark.parametrize("B,H,W,Cin,Cout,G", [(2, 32, 32, 320, 320, 32), (1, 16, 32, 64, 640, 32), (1, 24, 40, 64, 320, 32), (2, 16, 16, 96, 128, 8),
                                               (1, 64, 64, 32, 1280, 32)])
@pytest.mark.parametrize("cfg", [5, 22, 23])
@DTS
def test_conv3x3_epilogue_groupnorm_statistics(ops, B, H, W, Cin, Cout, G, cfg, dt):
    """The halo-patch conv's epilogue emits the GroupNorm statistics of its OUTPUT (bias, time-embedding vector and residual
    applied; groups that straddle 128-channel tiles, ragged pixel tiles): folded, they equal the statistics of the output tensor,
    and the next group_norm of that tensor -- which then skips its own statistics pass -- equals F.group_norm."""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, H, W, Cout).to(dt)
    gamma = 1.0 + 0.2 * rnd(6, Cout); beta = 0.2 * rnd(7, Cout)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), cfg=cfg, split_k=1,
                          gn_stats_groups=G)
    assert torch.equal(out, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), cfg=5, split_k=1))
    st = getattr(out, "_imd_gn_stats", None)
    assert st is not None and st[2] == G, "the halo-patch kernel must hand its statistics on"
    part, nparts, _ = st
    folded = part.float().sum(1).cpu()                                   # [B, G, 2]
    o = out.float().cpu().permute(0, 3, 1, 2).reshape(B, G, -1)          # [B, G, cpg * H * W] (values after 16-bit rounding)
    n = o.shape[-1]
    assert torch.allclose(folded[..., 0] / n, o.mean(-1), atol=2e-3), "group means"
    assert torch.allclose(folded[..., 1] / n, (o * o).mean(-1), rtol=5e-3, atol=2e-3), "group second moments"
    fused = ops.group_norm(out, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    plain_in = out.clone()                                               # (a clone carries no statistics: two-launch path)
    assert getattr(plain_in, "_imd_gn_stats", None) is None
    plain = ops.group_norm(plain_in, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    ref = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), G, gamma, beta, eps=1e-5)).permute(0, 2, 3, 1)
    assert_close(fused, ref, atol=2 * TOL[dt], what="group_norm on producer statistics")
    assert_close(fused, plain.float(), atol=2 * TOL[dt], what="producer statistics vs own pass")
    # K slices: the statistics then come from the finish launch (round 4; test_splitk_finish_groupnorm_statistics)
    out2 = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), cfg=5, split_k=2 if Cin >= 64 else 1, gn_stats_groups=G)
    assert getattr(out2, "_imd_gn_stats", None) is not None


@pytest.mark.parametrize("B,H,Cin,Cout,split", [
    (8, 8, 1280, 1280, 12),       # the 8x8-level conv of the bench batch: 8 images = 8 waves, 40 chunks over 12 slices (4 / 3 per slice)
    (8, 8, 2560, 128, 12),        # the up block's wide input
    (2, 8, 320, 64, 3),           # batch 1 (2 CFG rows): two waves
    (3, 8, 64, 192, 2),           # a ragged image group (3 of 4 waves own an image), 2 chunks per slice
    (11, 8, 96, 64, 4),           # two image groups, the second ragged; 3 chunks over 4 slices (one slice is empty)
    (4, 10, 160, 128, 5),         # the 10 x 8 level of the 512 x 640 geometry: three pixel blocks per image, the last half empty
    (5, 12, 32, 64, 2),           # 12 rows: the tallest map the kernel takes
    (1, 5, 32, 64, 2),            # one chunk (the second slice is empty), a short map
])
@DTS
def test_conv3x3_whole_small_maps(ops, B, H, Cin, Cout, split, dt):
    """tile config 24 (conv_img.hip: all pixels of up to 8 images x 64 channels x one K slice per workgroup, both operands by LDS-DMA
    into plane-layout ring slots) == F.conv2d on maps 8 pixels wide, finished by the shared split-K launch with the full epilogue."""
    W = 8
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, H, W, Cout).to(dt)
    ref = (F.conv2d(x.float(), w.float(), b, padding=1) + temb[:, :, None, None]).permute(0, 2, 3, 1) + res.float()
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), taps=9, rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), cfg=24, split_k=split)
    assert_close(out, ref, what=f"conv3x3 cfg=24 B={B} H={H}")
    # a pixel-strided input (channels 16 .. 16 + Cin of a wider NHWC buffer): same result
    wide = torch.full((B * H * W, Cin + 24), 7.0, dtype=dt, device="cuda")
    wide[:, :Cin] = xd.view(-1, Cin)
    out2 = ops.conv_gemm(wide, dev(pack_conv(w)), M=B * H * W, N=Cout, Cin=Cin, taps=9, Hin=H, Win=W, Hout=H, Wout=W, bias=dev(b),
                         rowvec=dev(temb), rowvec_stride=Cout, res=dev(res).view(-1, Cout), cfg=24, split_k=split, x_pix_stride=Cin + 24)
    assert torch.equal(out.view(-1, Cout), out2)
    with pytest.raises(ops.L.ImdError):           # un-split: refused, the kernel has no epilogue of its own
        ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), taps=9, cfg=24, split_k=1)


@pytest.mark.parametrize("cfg,B,H,W,Cin,Cout,G,split,stride", [
    (24, 8, 8, 8, 640, 1280, 32, 6, 1),        # the whole-map kernel of the 8-wide levels (always K-sliced)
    (2, 8, 8, 8, 1280, 1280, 32, 6, 1),        # the 8x8-level weight-streaming conv of the bench batch (64^2 register-staged tiles)
    (18, 2, 8, 8, 320, 1280, 32, 3, 1),        # gathering LDS-DMA tiles
    (5, 2, 16, 16, 640, 1280, 32, 4, 1),       # halo-patch kernel with K slices
    (21, 1, 32, 32, 640, 640, 32, 3, 1),       # 16 x 16-pixel-tile halo-patch kernel
    (0, 2, 32, 32, 320, 320, 32, 2, 2),        # stride-2 downsample conv (feeds the next level's norm1)
    (2, 3, 8, 8, 128, 96, 8, 2, 1),            # 12 channels per group: 8-channel chunks straddle groups
    (0, 2, 16, 16, 64, 2560, 32, 2, 1),        # 320 columns: one full pass of the block
])
@DTS
def test_splitk_finish_groupnorm_statistics(ops, cfg, B, H, W, Cin, Cout, G, split, stride, dt):
    """A K-sliced convolution whose output feeds a GroupNorm: the finish launch (slice sum + bias / time-embedding vector / residual) also
    emits the statistics of the tensor it stores; the tensor is bit-identical to the plain finish, the folded partials are the tensor's
    moments, and the group_norm that consumes them equals F.group_norm of the stored tensor."""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    Ho, Wo = H // stride, W // stride
    b = rnd(3, Cout); temb = rnd(4, B, Cout); res = rnd(5, B, Ho, Wo, Cout).to(dt)
    gamma = 1.0 + 0.2 * rnd(6, Cout); beta = 0.2 * rnd(7, Cout)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    kw = dict(rowvec=dev(temb), rowvec_stride=Cout, res=dev(res), cfg=cfg, split_k=split, stride=stride)
    out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), gn_stats_groups=G, **kw)
    plain_out = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), **kw)
    assert getattr(plain_out, "_imd_gn_stats", None) is None
    assert torch.equal(out, plain_out), "the statistics form of the finish launch must store the same tensor"
    st = getattr(out, "_imd_gn_stats", None)
    assert st is not None and st[2] == G
    part, nparts, _ = st
    assert tuple(part.shape) == (B, nparts, G, 2)
    folded = part.double().sum(1).cpu()
    o = out.double().cpu().permute(0, 3, 1, 2).reshape(B, G, -1)
    n = o.shape[-1]
    assert torch.allclose(folded[..., 0] / n, o.mean(-1), atol=1e-4), "group means"
    assert torch.allclose(folded[..., 1] / n, (o * o).mean(-1), rtol=1e-4, atol=1e-4), "group second moments"
    fused = ops.group_norm(out, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    ref = F.silu(F.group_norm(out.float().cpu().permute(0, 3, 1, 2), G, gamma, beta, eps=1e-5)).permute(0, 2, 3, 1)
    assert_close(fused, ref, atol=2 * TOL[dt], what="group_norm on the finish launch's statistics")
    own = ops.group_norm(plain_out, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    assert_close(fused, own.float(), atol=2 * TOL[dt], what="finish statistics vs own pass")
    with pytest.raises(ops.L.ImdError):           # a request the launch cannot honour is an error at the C ABI, not a silent skip
        p = dict(kw, cfg=6, split_k=1)
        t = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), **p)
        q = ops.L.ConvGemmParams()
        q.x, q.w, q.out = xd.data_ptr(), dev(pack_conv(w)).data_ptr(), t.data_ptr()
        q.M, q.N, q.K, q.Cin, q.taps = B * Ho * Wo, Cout, 9 * Cin, Cin, 9
        q.Hin, q.Win, q.Hout, q.Wout, q.stride, q.x_pix_stride, q.out_ld, q.res_ld = H, W, Ho, Wo, stride, Cin, Cout, Cout
        q.out_scale, q.split_k, q.dtype = 1.0, 1, (1 if dt == torch.float16 else 0)
        q.gn_stats_out, q.gn_stats_groups = part.data_ptr(), G
        ops.L.check(ops.L.load().imd_conv_gemm(ctypes.byref(q), 6, 0))          # (the 64 x 320 tiles have no statistics epilogue)


@pytest.mark.parametrize("cfg,B,H,W,Cin,Cout,G,stride,taps", [
    (0, 2, 64, 64, 8, 320, 32, 1, 9),          # conv_in: 128 x 128 tiles, 32 tiles per image x 3 column tiles (the last one half empty)
    (-1, 2, 64, 64, 8, 320, 32, 1, 9),         # ... with the tile config left to the library
    (2, 2, 32, 32, 64, 320, 32, 2, 9),         # the stride-2 downsampler of the 64x64 level: 64 x 64 tiles, 10 channels per group (chunks straddle groups)
    (2, 4, 8, 8, 128, 1280, 32, 1, 1),         # Transformer2DModel.proj_out of the 8x8 level (+ residual): one 64-row tile per image
    (1, 2, 16, 16, 64, 192, 8, 1, 9),          # 128 x 64 tiles, 24 channels per group
    (4, 1, 32, 32, 32, 96, 8, 1, 9),           # 128 x 128 x 32 tiles, 12 channels per group, one ragged column tile
    (3, 2, 16, 16, 32, 128, 16, 1, 9), (7, 2, 16, 16, 32, 128, 16, 1, 1),
])
@DTS
def test_register_staged_tiles_groupnorm_statistics(ops, monkeypatch, cfg, B, H, W, Cin, Cout, G, stride, taps, dt):
    """The register-staged tile kernel (tile configs 0..4, 7) writes the GroupNorm statistics of the tensor it stores where a tile's rows lie in one
    image (round 6: conv_in, the 64x64-level downsampler, proj_out of the 8x8 level): same tensor as without, partials = the stored tensor's moments,
    group_norm on them == group_norm with its own statistics pass within the 16-bit bar; silently absent where a tile would span images."""
    monkeypatch.setattr(ops, "GENERIC_GN_STATS", True)          # (opt-in: measured neutral / slower end to end, see ops.py)
    x = rnd(1, B, H, W, Cin).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt) if taps == 9 else rnd(2, Cout, Cin, scale=Cin ** -0.5).to(dt)
    wd = dev(pack_conv(w)) if taps == 9 else dev(w)
    Ho, Wo = H // stride, W // stride
    b = rnd(3, Cout); res = rnd(5, B, Ho, Wo, Cout).to(dt)
    gamma = 1.0 + 0.2 * rnd(6, Cout); beta = 0.2 * rnd(7, Cout)
    kw = dict(res=dev(res), cfg=cfg, split_k=1, stride=stride, taps=taps)
    out = ops.conv2d_nhwc(dev(x), wd, dev(b), gn_stats_groups=G, **kw)
    plain = ops.conv2d_nhwc(dev(x), wd, dev(b), **kw)
    assert torch.equal(out, plain)
    st = getattr(out, "_imd_gn_stats", None)
    assert st is not None and st[2] == G and tuple(st[0].shape) == (B, st[1], G, 2)
    folded = st[0].double().sum(1).cpu()
    o = out.double().cpu().permute(0, 3, 1, 2).reshape(B, G, -1)
    n = o.shape[-1]
    assert torch.allclose(folded[..., 0] / n, o.mean(-1), atol=1e-4), "group means"
    assert torch.allclose(folded[..., 1] / n, (o * o).mean(-1), rtol=1e-4, atol=1e-4), "group second moments"
    fused = ops.group_norm(out, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    own = ops.group_norm(plain, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=True)
    assert_close(fused, own.float(), atol=2 * TOL[dt], what="epilogue statistics vs own pass")
    monkeypatch.setattr(ops, "GENERIC_GN_STATS", False)
    assert getattr(ops.conv2d_nhwc(dev(x), wd, dev(b), gn_stats_groups=G, **kw), "_imd_gn_stats", None) is None


@DTS
def test_register_staged_tiles_statistics_absent_when_a_tile_spans_images(ops, monkeypatch, dt):
    monkeypatch.setattr(ops, "GENERIC_GN_STATS", True)
    x = dev(rnd(1, 4, 8, 8, 64).to(dt)); w = dev(rnd(2, 128, 64, scale=0.125).to(dt))
    out = ops.conv2d_nhwc(x, w, None, taps=1, cfg=0, split_k=1, gn_stats_groups=16)          # 64 pixels per image, 128-row tiles
    assert getattr(out, "_imd_gn_stats", None) is None
    assert_close(out, x.float() @ w.float().t(), what="conv")


@pytest.mark.parametrize("cfg,B,H,W,Cin,Cout,G,split", [
    (24, 8, 8, 8, 1280, 1280, 32, 6),          # the whole-map kernel of the 8-wide level (bench batch): 64 pixels x 40 channels per (image, group)
    (2, 2, 8, 8, 1280, 1280, 32, 6),           # batch 1 (CFG pair) on the register-staged 64^2 tiles
    (5, 8, 16, 16, 640, 1280, 32, 4),          # 16x16 level, halo-patch kernel with K slices: 256 pixels x 40 channels = 2560 units
    (18, 2, 16, 16, 320, 640, 32, 3),          # gathering LDS-DMA tiles; 20 channels per group (five 4-channel units per pixel)
    (0, 3, 8, 16, 64, 96, 8, 2),               # 12 channels per group, ragged unit count per thread
])
@pytest.mark.parametrize("silu", [True, False])
@DTS
def test_splitk_finish_with_groupnorm_of_the_output(ops, cfg, B, H, W, Cin, Cout, G, split, silu, dt):
    """imd_conv_gemm_params.gn_out_* (ABI v9): the finish launch of a K-sliced convolution owns whole (image, group) slabs and applies GroupNorm (+ SiLU)
    to its output itself -- ResnetBlock2D conv1 -> (+ temb) -> norm2 -> SiLU as TWO launches instead of three, the raw conv1 output never stored.
    == the plain finish followed by imd_groupnorm up to the statistics' fp32 summation order (both normalise the values ROUNDED to the
    element type): compared against F.group_norm of the stored plain tensor at the engine's GroupNorm tolerance, and elementwise against the
    two-launch result (at most a few last-place differences where a statistic's last bit moved a rounding)."""
    x = rnd(1, B, Cin, H, W).to(dt)
    w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout); temb = rnd(4, B, Cout)
    gamma = 1.0 + 0.2 * rnd(6, Cout); beta = 0.2 * rnd(7, Cout)
    xd = dev(x.permute(0, 2, 3, 1).contiguous())
    kw = dict(rowvec=dev(temb), rowvec_stride=Cout, cfg=cfg, split_k=split)
    from imagdressing_amd import ops as ops_mod
    ops_mod.FUSED_GN_FINISH = True               # (opt-in switch: measured no faster than the two launches, see ops.py; restored by the autouse knob fixture)
    plain = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), gn_stats_groups=G, **kw)
    two = ops.group_norm(plain, dev(gamma), dev(beta), groups=G, eps=1e-5, silu=silu)
    fused = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), gn_stats_groups=G, gn_out=(dev(gamma), dev(beta), 1e-5, silu, G), **kw)
    assert getattr(fused, "_imd_gn_applied", False), "the finish launch should have taken the normalisation"
    assert getattr(fused, "_imd_gn_stats", None) is None
    ref = F.group_norm(plain.float().cpu().permute(0, 3, 1, 2), G, gamma, beta, eps=1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1)
    assert_close(fused, ref, atol=2 * TOL[dt], what="finish launch with GroupNorm vs F.group_norm of the stored tensor")
    d = (fused.float() - two.float()).abs()
    ulp = 2.0 ** (-7 if dt == bf16 else -10)
    assert (d <= ulp * two.float().abs().clamp_min(1.0)).all(), d.max().item()
    assert (d > 0).float().mean().item() < 0.02, "more than last-place noise between the one- and two-launch forms"
    assert torch.equal(fused, ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), gn_out=(dev(gamma), dev(beta), 1e-5, silu, G), **kw)), "not deterministic"
    # a residual (conv2) or an un-sliced launch cannot take it: the wrapper falls back to the raw output, the C ABI refuses
    raw = ops.conv2d_nhwc(xd, dev(pack_conv(w)), dev(b), res=dev(rnd(5, B, H, W, Cout).to(dt)), gn_out=(dev(gamma), dev(beta), 1e-5, silu, G), **kw)
    assert not getattr(raw, "_imd_gn_applied", False)
    import ctypes
    q = ops.L.ConvGemmParams()
    q.x, q.w, q.out = xd.data_ptr(), dev(pack_conv(w)).data_ptr(), raw.data_ptr()
    q.M, q.N, q.K, q.Cin, q.taps = B * H * W, Cout, 9 * Cin, Cin, 9
    q.Hin, q.Win, q.Hout, q.Wout, q.stride, q.x_pix_stride, q.out_ld, q.res_ld = H, W, H, W, 1, Cin, Cout, Cout
    q.out_scale, q.split_k, q.dtype = 1.0, 1, (1 if dt == torch.float16 else 0)
    q.gn_out_gamma, q.gn_out_beta, q.gn_out_eps, q.gn_out_silu, q.gn_out_groups = dev(gamma).data_ptr(), dev(beta).data_ptr(), 1e-5, int(silu), G
    assert ops.L.load().imd_conv_gemm_gn_out_supported(ctypes.byref(q)) == 0
    with pytest.raises(ops.L.ImdError):
        ops.L.check(ops.L.load().imd_conv_gemm(ctypes.byref(q), 0, 0))


@DTS
def test_conv_epilogue_rowvec_residual_scale(ops, dt):
    B, H, W, Cin, Cout = 2, 8, 8, 64, 128
    x = rnd(1, B, Cin, H, W).to(dt); w = rnd(2, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5).to(dt)
    b = rnd(3, Cout); temb = rnd(4, B, 256); res = rnd(5, B, H, W, Cout).to(dt)
    conv = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1)
    ref = (conv + temb[:, None, None, 64:64 + Cout]) * 0.5 + res.float()
    out = ops.conv2d_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), dev(pack_conv(w)), dev(b),
                          rowvec=dev(temb[:, 64:].contiguous()), rowvec_stride=192, res=dev(res), out_scale=0.5)
    assert_close(out, ref, what="conv rowvec+res+scale")


@DTS
def test_conv1x1(ops, dt):
    B, H, W, Cin, Cout = 2, 8, 8, 320, 320
    x = rnd(1, B, H, W, Cin).to(dt); w = rnd(2, Cout, Cin, scale=Cin ** -0.5).to(dt); b = rnd(3, Cout)
    ref = x.float() @ w.float().t() + b
    out = ops.conv2d_nhwc(dev(x), dev(w), dev(b), taps=1)
    assert_close(out, ref, what="conv1x1")


# ------------------------------------------------------------------------------------------
# head-split epilogue + attention
# ------------------------------------------------------------------------------------------
def to_heads(x, H, DP, scale=1.0, dt=bf16):
    B, Lt, Cc = x.shape
    d = Cc // H
    out = torch.zeros(B, H, Lt, DP, dtype=dt)
    out[..., :d] = (x.float() * scale).to(dt).view(B, Lt, H, d).transpose(1, 2)
    return out


def to_heads_t(x, H, DPV, LP, dt=bf16):
    B, Lt, Cc = x.shape
    d = Cc // H
    out = torch.zeros(B, H, DPV, LP, dtype=dt)
    out[:, :, :d, :Lt] = x.view(B, Lt, H, d).permute(0, 2, 3, 1)
    return out


@DTS
def test_qkv_head_split(ops, dt):
    B, Lt, Cc, H = 2, 100, 320, 8
    d = Cc // H
    dpk, dpv = ops.attn_padded_dims(d)
    LP = ops.pad64(Lt)
    x = rnd(1, B * Lt, Cc).to(dt); w = rnd(2, 3 * Cc, Cc, scale=Cc ** -0.5).to(dt)
    q = torch.zeros(B, H, Lt, dpk, dtype=dt, device="cuda"); k = torch.zeros_like(q)
    vt = torch.zeros(B, H, dpv, LP, dtype=dt, device="cuda")
    ops.conv_gemm(dev(x), dev(w), M=B * Lt, N=3 * Cc, Cin=Cc, Hout=Lt, Wout=1, Hin=Lt, Win=1,
                  heads=dict(C=Cc, H=H, D=d, dests=[(q, 0, dpk, Lt, 0.25), (k, 0, dpk, Lt, 1.0), (vt, 1, dpv, LP, 1.0)]))
    y = (x.float() @ w.float().t()).view(B, Lt, 3 * Cc)
    assert_close(q, to_heads(y[..., :Cc] * 0.25, H, dpk, dt=dt), what="q heads")
    assert_close(k, to_heads(y[..., Cc:2 * Cc], H, dpk, dt=dt), what="k heads")
    assert_close(vt, to_heads_t(y[..., 2 * Cc:].to(dt), H, dpv, LP, dt=dt), what="v^T heads")


def ref_attn(q, k, v, H):
    from oracle.processors import sdpa
    return sdpa(q.float(), k.float(), v.float(), H)


@pytest.mark.parametrize("D,B,N,L1,L2", [
    (40, 2, 200, 200, 330), (40, 1, 1100, 1100, 0), (80, 2, 144, 144, 100), (160, 1, 64, 64, 80),
    (64, 2, 16, 273, 0), (40, 2, 130, 77, 4), (160, 1, 70, 77, 0)])
@DTS
def test_attention(ops, D, B, N, L1, L2, dt):
    H = 8
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt)
    k1 = rnd(2, B, L1, Cc).to(dt); v1 = rnd(3, B, L1, Cc).to(dt)
    scale = D ** -0.5 * math.log2(math.e)
    qh = dev(to_heads(q, H, dpk, scale, dt=dt))
    out = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ref = ref_attn(q, k1, v1, H)
    kw = {}
    if L2:
        k2 = rnd(4, 1, L2, Cc).to(dt); v2 = rnd(5, 1, L2, Cc).to(dt)
        s2 = torch.tensor([0.9, 0.0][:B] if B == 2 else [0.9])
        r2 = ref_attn(q, k2.expand(B, -1, -1), v2.expand(B, -1, -1), H)
        # phase 1 is rounded to bf16 before the add (the reference adds two half tensors, :612)
        ref = ref.to(dt).float() + s2[:, None, None] * r2
        kw = dict(k2=dev(to_heads(k2, H, dpk, dt=dt)), v2t=dev(to_heads_t(v2, H, dpv, ops.pad64(L2), dt=dt)), scale2=dev(s2),
                  L2=L2, L2P=ops.pad64(L2), kv2_bdiv=B)
    ops.attention(qh, dev(to_heads(k1, H, dpk, dt=dt)), dev(to_heads_t(v1, H, dpv, ops.pad64(L1), dt=dt)), out,
                  B=B, H=H, N=N, D=D, L1=L1, L1P=ops.pad64(L1), **kw)
    assert_close(out, ref, atol=1e-2, rtol=1e-2, what=f"attention D={D}")


@pytest.mark.parametrize("variant", [13, 12, 10, 11, 9, 7])
@pytest.mark.parametrize("pad_one", [True, False])
@pytest.mark.parametrize("N,L1,L2", [(640, 640, 330), (530, 700, 0), (512, 1000, 520), (768, 1408, 1216), (512, 1344, 64)])
@DTS
def test_attention_d40_kernel_variants(ops, N, L1, L2, pad_one, variant, dt):
    """The software-pipelined head-dim-40 kernel (attention_d40.hip; N >= 512) in every shipped variant -- 13: the default
    (interior steps unchecked; fp16 with the biased reference maximum), 12: compile-time ring slots, 10: head-dim rows 32..40 of P.V on v_mfma_f32_16x16x32 (round-3 default),
    9: the round-2 kernel, 11 / 7: the same two with register staging -- with and without the caller's K pad-column guarantee
    (LDS-DMA vs register staging), ragged key counts, a second key set on one of two batch rows, and a late spike that takes the
    exact (redo) path and raises the deferred maximum."""
    out, ref = _run_d40_variant(ops, N, L1, L2, pad_one, variant, 3.0, dt)
    assert_close(out, ref, atol=1e-2 if dt == torch.float16 else 2e-2, rtol=2e-2, what=f"attention d40 variant {variant}")


def _run_d40_variant(ops, N, L1, L2, pad_one, variant, spike, dt, spike_at=None):
    D, H, B = 40, 8, 2
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt)
    k1 = rnd(2, B, L1, Cc).to(dt); v1 = rnd(3, B, L1, Cc).to(dt)
    k1[:, L1 - 90 if spike_at is None else spike_at] = q[:, 7] * spike      # a key far above the rest late in the sequence (or where the caller says)
    scale = D ** -0.5 * math.log2(math.e)

    def kbuf(x):
        h = to_heads(x, H, dpk, dt=dt)
        if pad_one:
            h[..., D] = 1.0
        return dev(h)
    out = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ref = ref_attn(q, k1, v1, H)
    kw = {}
    if L2:
        k2 = rnd(4, 1, L2, Cc).to(dt); v2 = rnd(5, 1, L2, Cc).to(dt)
        k2[:, L2 - 40] = q[1, 300] * spike
        s2 = torch.tensor([0.9, 0.0])
        r2 = ref_attn(q, k2.expand(B, -1, -1), v2.expand(B, -1, -1), H)
        ref = ref.to(dt).float() + s2[:, None, None] * r2
        kw = dict(k2=kbuf(k2), v2t=dev(to_heads_t(v2, H, dpv, ops.pad64(L2), dt=dt)), scale2=dev(s2), L2=L2, L2P=ops.pad64(L2),
                  kv2_bdiv=B)
    lib = ops.L.load()
    prev = lib.imd_get_tuning(0)
    ops.L.check(lib.imd_set_tuning(0, variant))
    try:
        ops.attention(dev(to_heads(q, H, dpk, scale, dt=dt)), kbuf(k1), dev(to_heads_t(v1, H, dpv, ops.pad64(L1), dt=dt)), out,
                      B=B, H=H, N=N, D=D, L1=L1, L1P=ops.pad64(L1), k_pad_one=pad_one, **kw)
        torch.cuda.synchronize()
    finally:
        ops.L.check(lib.imd_set_tuning(0, prev))
    assert torch.isfinite(out).all()
    return out, ref


@pytest.mark.parametrize("N,L1,L2", [(640, 640, 330), (512, 1000, 520), (768, 1408, 0)])
@pytest.mark.parametrize("where", ["late", "interior"])
@DTS
def test_attention_d40_unchecked_steps_rerun_on_overflow(ops, N, L1, L2, where, dt):
    """Variant 13 (the default of both element types) tests the deferred maximum only on the first and last steps of a phase.  A key whose
    score sits far above everything the first block held -- 170..400 base-2 units for bf16 (P = 2^(s - m_ref) leaves fp32), 35..80 for fp16
    (its P = 2^(s - m_first - 4) passes 65504 at + 20) -- planted EITHER among the last keys (checked steps: the exact path raises the
    reference maximum, nothing overflows) OR in the middle of the sequence (unchecked interior step: the workgroup of query rows 0..255
    finds an infinite softmax denominator when the phase is done and runs again as variant 12).  Either way the rows of that workgroup
    must be finite and -- where the re-run happened -- variant 12's bit for bit; workgroups that stay in range keep the unchecked result,
    which differs from 12's by rounding only (12 rescales where 13 lets P grow)."""
    D, H = 40, 8
    spike = 30.0 if dt == torch.bfloat16 else 6.0
    q = rnd(1, 2, N, H * D).to(dt).float().view(2, N, H, D)
    k0 = rnd(2, 2, L1, H * D).to(dt).float().view(2, L1, H, D)
    sc = D ** -0.5 * math.log2(math.e)
    first = torch.einsum("bhd,bkhd->bhk", q[:, 7], k0[:, :32]).amax(-1) * sc          # query 7's maximum over the first 32-key block
    late = (q[:, 7] * q[:, 7]).sum(-1) * spike * sc                                    # ... and its score on the planted key
    assert ((late - first) > (130 if dt == torch.bfloat16 else 26)).all()              # P is past the format's range in every head
    at = None if where == "late" else (L1 // 2) // 64 * 64 + 5                         # an interior 64-key unit of the first key set
    out13, _ = _run_d40_variant(ops, N, L1, L2, True, 13, spike, dt, spike_at=at)
    out12, _ = _run_d40_variant(ops, N, L1, L2, True, 12, spike, dt, spike_at=at)
    if where == "interior":
        assert torch.equal(out13[:, :256], out12[:, :256])
    assert_close(out13, out12, atol=2e-2, rtol=2e-2, what="variant 13 vs 12 under large scores")


@DTS
def test_per_call_tuning_matches_the_process_wide_knobs(ops, dt):
    """IMD_TUNING_PER_CALL (ops.tuning_scope): the head-dim-40 variant / work order and the GEMM tuning bits chosen per call give exactly what
    the process-wide knobs of imd_set_tuning give, and leave those knobs alone."""
    lib = ops.L.load()
    before = [lib.imd_get_tuning(k) for k in range(3)]
    for variant in (7, 12):
        ref, _ = _run_d40_variant(ops, 640, 1000, 520, True, variant, 3.0, dt)
        D, H, B, N, L1 = 40, 8, 2, 640, 1000
        with ops.tuning_scope(attn_variant=variant):
            got, _ = _run_d40_variant(ops, N, L1, 520, True, before[0], 3.0, dt)      # (the helper sets knob 0 to the shipped value: the scope must win)
        assert torch.equal(got, ref), f"per-call variant {variant}"
    x = rnd(1, 2, 32, 32, 64).to(dt); w = rnd(2, 128, 64, 3, 3, scale=(9 * 64) ** -0.5).to(dt); b = rnd(3, 128)
    wp = dev(pack_conv(w))
    ops.L.check(lib.imd_set_tuning(2, 0))
    try:
        plain = ops.conv2d_nhwc(dev(x), wp, dev(b), cfg=0)
    finally:
        ops.L.check(lib.imd_set_tuning(2, before[2]))
    with ops.tuning_scope(gemm_flags=0):
        scoped = ops.conv2d_nhwc(dev(x), wp, dev(b), cfg=0)
    assert torch.equal(plain, scoped)
    assert_close(scoped, F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1).permute(0, 2, 3, 1), what="conv under a tuning scope")
    assert [lib.imd_get_tuning(k) for k in range(3)] == before


@pytest.mark.parametrize("pad_one", [True, False])
@pytest.mark.parametrize("B,N,L1,L2", [(2, 640, 640, 330), (3, 1000, 1000, 1000), (2, 512, 700, 0), (4, 4096, 4096, 4096)])
@DTS
def test_attention_d40_fused_out_projection(ops, B, N, L1, L2, pad_one, dt):
    """ABI v7: the block's out-projection (to_out[0] + bias + residual, attention_processor.py:614-622) inside the attention launch --
    heads of a 256-row block joined through write-through O tiles and an arrival counter, the last head projects.  == the same
    attention followed by imd_conv_gemm: the O buffer bit for bit, the projection up to fp32 summation order; ragged row blocks
    (N = 1000, 640), garment rows and plain rows in one launch, repeated launches (counters left at zero)."""
    D, H = 40, 8
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt)
    k1 = rnd(2, B, L1, Cc).to(dt); v1 = rnd(3, B, L1, Cc).to(dt)
    wo = rnd(6, Cc, Cc, scale=Cc ** -0.5).to(dt); bo = rnd(7, Cc); res = rnd(8, B, N, Cc).to(dt)
    scale = D ** -0.5 * math.log2(math.e)

    def kbuf(x):
        h = to_heads(x, H, dpk, dt=dt)
        if pad_one:
            h[..., D] = 1.0
        return dev(h)
    kw = {}
    if L2:
        k2 = rnd(4, 1, L2, Cc).to(dt); v2 = rnd(5, 1, L2, Cc).to(dt)
        s2 = torch.tensor([0.9] + [0.0] * (B - 1)) if B < 4 else torch.tensor([1.0, 1.0, 0.0, 0.0])
        kw = dict(k2=kbuf(k2), v2t=dev(to_heads_t(v2, H, dpv, ops.pad64(L2), dt=dt)), scale2=dev(s2), L2=L2, L2P=ops.pad64(L2), kv2_bdiv=B)
    args = (dev(to_heads(q, H, dpk, scale, dt=dt)), kbuf(k1), dev(to_heads_t(v1, H, dpv, ops.pad64(L1), dt=dt)))
    common = dict(B=B, H=H, N=N, D=D, L1=L1, L1P=ops.pad64(L1), k_pad_one=pad_one, **kw)
    assert ops.attention_proj_supported(H, N, D)
    o_plain = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    # the fused form is built on variant 12 (per-step overflow test): the plain launch it must reproduce BIT FOR BIT is variant 12's.  (The default
    # plain launch -- variant 13 -- agrees with it bit for bit in bf16; in fp16 its P carries the 2^-4 bias, which re-rounds the subnormal tail.)
    with ops.tuning_scope(attn_variant=12):
        ops.attention(*args, o_plain, **common)
    o_default = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ops.attention(*args, o_default, **common)
    assert_close(o_default, o_plain.float(), atol=2e-2 if dt == bf16 else 2e-3, rtol=1e-2, what="default variant vs variant 12")
    two = ops.linear(o_plain.view(B * N, Cc), dev(wo), dev(bo), res=dev(res).view(B * N, Cc)).view(B, N, Cc)
    exact = o_plain.float() @ dev(wo).float().t() + dev(bo) + dev(res).float()
    for rep in range(3):
        o_f = torch.full((B, N, Cc), float("nan"), dtype=dt, device="cuda")
        fused = torch.full((B, N, Cc), float("nan"), dtype=dt, device="cuda")
        got = ops.attention(*args, o_f, proj=(dev(wo), dev(bo), dev(res), fused), **common)
        assert got is fused
        torch.cuda.synchronize()
        assert torch.equal(o_f, o_plain), "the O hand-off buffer differs from the plain launch"
        assert_close(fused, exact, atol=3e-2 if dt == bf16 else 6e-3, rtol=1e-2, what=f"fused out-projection (launch {rep})")
        assert_close(fused, two.float(), atol=3e-2 if dt == bf16 else 6e-3, rtol=1e-2, what="fused vs attention + imd_conv_gemm")
        assert int(ops.proj_counters(1, fused.device).abs().max()) == 0
    # no bias / no residual
    f2 = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ops.attention(*args, torch.empty_like(o_plain), proj=(dev(wo), None, None, f2), **common)
    assert_close(f2, o_plain.float() @ dev(wo).float().t(), atol=3e-2 if dt == bf16 else 6e-3, rtol=1e-2, what="fused out-projection, bare")
    with pytest.raises(ops.L.ImdError):          # other head dims keep the separate launch
        ops.attention(dev(rnd(1, 1, 8, 96, 80).to(dt)), dev(rnd(2, 1, 8, 96, 80).to(dt)), dev(rnd(3, 1, 8, 80, 128).to(dt)), torch.empty(1, 96, 640, dtype=dt, device="cuda"),
                      B=1, H=8, N=96, D=80, L1=96, L1P=128, proj=(dev(rnd(6, 640, 640).to(dt)), None, None, torch.empty(1, 96, 640, dtype=dt, device="cuda")))


@DTS
def test_attention_shared_kv_batch_div(ops, dt):
    """text K/V computed once per prompt and shared by groups of batch rows (kv batch = b // bdiv)."""
    D, H, B, N, L1 = 40, 8, 4, 96, 77
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt); k = rnd(2, 2, L1, Cc).to(dt); v = rnd(3, 2, L1, Cc).to(dt)
    out = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ops.attention(dev(to_heads(q, H, dpk, D ** -0.5 * math.log2(math.e), dt=dt)), dev(to_heads(k, H, dpk, dt=dt)),
                  dev(to_heads_t(v, H, dpv, 128, dt=dt)), out, B=B, H=H, N=N, D=D, L1=L1, L1P=128, kv1_bdiv=2)
    ref = ref_attn(q, k.repeat_interleave(2, 0), v.repeat_interleave(2, 0), H)
    assert_close(out, ref, what="kv batch div")


@DTS
def test_attention_softmax_spike(ops, dt):
    """A key far above the rest late in the sequence forces the online-softmax rescale path."""
    D, H, B, N, L1 = 40, 8, 1, 64, 256
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt); k = rnd(2, B, L1, Cc).to(dt); v = rnd(3, B, L1, Cc).to(dt)
    k[:, 200] = q[:, 5] * 4.0          # row 5 (and friends) suddenly meet a huge logit in tile 3
    out = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ops.attention(dev(to_heads(q, H, dpk, D ** -0.5 * math.log2(math.e), dt=dt)), dev(to_heads(k, H, dpk, dt=dt)),
                  dev(to_heads_t(v, H, dpv, 256, dt=dt)), out, B=B, H=H, N=N, D=D, L1=L1, L1P=256)
    assert_close(out, ref_attn(q, k, v, H), atol=2e-2, what="spike")


@DTS
def test_attention_extreme_logits(ops, dt):
    """Rows whose logits are all hugely negative (the deferred max must be LOWERED on the first tile, or P underflows
    to 0 and 0/0 appears) and rows whose logits grow tile after tile (repeated raises of the running reference)."""
    D, H, B, N, L1 = 40, 8, 1, 64, 320
    Cc = H * D
    dpk, dpv = ops.attn_padded_dims(D)
    q = rnd(1, B, N, Cc).to(dt); k = rnd(2, B, L1, Cc).to(dt); v = rnd(3, B, L1, Cc).to(dt)
    base = rnd(4, 1, 1, Cc)
    q[:, :8] = base * 3.0                       # rows 0..7 share a direction ...
    k[:, :, :] = k * 0.3 - base * 4.0           # ... that every key opposes: all logits << 0 for those rows
    ramp = torch.linspace(0.0, 6.0, L1).view(1, L1, 1)
    k[:, :, :] = k + ramp * q[:, 20:21] * 0.2   # row 20 sees logits that keep increasing along the key axis
    k = k.to(dt)
    out = torch.empty(B, N, Cc, dtype=dt, device="cuda")
    ops.attention(dev(to_heads(q, H, dpk, D ** -0.5 * math.log2(math.e), dt=dt)), dev(to_heads(k, H, dpk, dt=dt)),
                  dev(to_heads_t(v, H, dpv, 320, dt=dt)), out, B=B, H=H, N=N, D=D, L1=L1, L1P=320)
    assert torch.isfinite(out.float()).all()
    assert_close(out, ref_attn(q, k, v, H), atol=2e-2, rtol=2e-2, what="extreme logits")


# ------------------------------------------------------------------------------------------
# norms / elementwise
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,HW,Cc", [(2, 256, 320), (1, 100, 640), (2, 64, 960), (1, 70, 1280), (1, 16, 1920), (1, 9, 2560),
                                      (2, 300, 128), (1, 65536, 128), (1, 20000, 256)])      # VAE: 4 channels / group; two-level fold
@pytest.mark.parametrize("silu", [False, True])
@DTS
def test_groupnorm(ops, B, HW, Cc, silu, dt):
    x = (rnd(1, B, HW, Cc) * 1.5 + 0.3).to(dt)
    g = 1.0 + rnd(2, Cc, scale=0.2); b = rnd(3, Cc, scale=0.2)
    ref = F.group_norm(x.float().transpose(1, 2), 32, g, b, eps=1e-5).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    out = ops.group_norm(dev(x), dev(g), dev(b), eps=1e-5, silu=silu)
    assert_close(out, ref, atol=2e-2, rtol=1e-2, what="groupnorm")


@pytest.mark.parametrize("rows,Cc", [(300, 320), (77, 768), (10, 1280), (5, 640)])
@DTS
def test_layernorm(ops, rows, Cc, dt):
    x = (rnd(1, rows, Cc) * 2 + 0.5).to(dt)
    g = 1.0 + rnd(2, Cc, scale=0.2); b = rnd(3, Cc, scale=0.2)
    ref = F.layer_norm(x.float(), (Cc,), g, b, 1e-5)
    assert_close(ops.layer_norm(dev(x), dev(g), dev(b)), ref, atol=2e-2, what="layernorm")


def test_timestep_embedding(ops):
    from oracle.sd15 import timestep_embedding
    t = torch.tensor([981.0, 1.0, 500.0])
    assert_close(ops.timestep_embedding(dev(t), 320), timestep_embedding(t, 320), atol=2e-3, rtol=0, what="timestep emb")


@pytest.mark.parametrize("inpaint", [False, True])
@DTS
def test_ddim_cfg_step(ops, inpaint, dt):
    from oracle.ddim import DDIMOracle
    B, HW = 3, 500
    sch = DDIMOracle(); sch.set_timesteps(50)
    t = int(sch.timesteps[7]); t_next = int(sch.timesteps[8])
    z = rnd(1, B, HW, 4); eps = rnd(2, 2 * B, HW, 4); g = 7.5
    e = eps[B:] + g * (eps[:B] - eps[B:])
    ref = sch.step(e, t, z)
    kw = {}
    if inpaint:
        mask = (rnd(3, B, HW) > 0).float(); zi = rnd(4, B, HW, 4); nz = rnd(5, B, HW, 4)
        ref = (1 - mask[..., None]) * sch.add_noise(zi, nz, t_next) + mask[..., None] * ref
        kw = dict(mask=dev(mask), z_img=dev(zi), noise=dev(nz), a_next=float(sch.alphas_cumprod[t_next]))
    zd = dev(z.clone()); xn = torch.empty(2 * B, HW, 8, dtype=dt, device="cuda")
    prev_t = t - 1000 // 50
    ops.ddim_cfg_step(zd, dev(eps), xn, guidance=g, a_t=float(sch.alphas_cumprod[t]),
                      a_prev=float(sch.alphas_cumprod[prev_t]), **kw)
    assert_close(zd, ref, atol=1e-4, rtol=1e-4, what="ddim z")
    assert_close(xn[:B, :, :4], ref, atol=2e-2, what="next input cond")
    assert_close(xn[B:, :, :4], ref, atol=2e-2, what="next input uncond")
    assert float(xn[..., 4:].abs().max()) == 0.0


@pytest.mark.parametrize("inpaint", [False, True])
@pytest.mark.parametrize("eta", [0.3, 1.0])
def test_ddim_cfg_step_stochastic(ops, inpaint, eta):
    """eta > 0 (DDIMScheduler.step's variance noise, reached through prepare_extra_step_kwargs, IMAGDressing_v1_pipeline.py:102-119):
    the fused step == oracle step with the same noise; the noise is added BEFORE the inpaint blend"""
    from oracle.ddim import DDIMOracle
    from imagdressing_amd.scheduler import DDIMScheduler
    B, HW = 2, 700
    sch = DDIMOracle(); sch.set_timesteps(50)
    mine = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                         set_alpha_to_one=False, steps_offset=1); mine.set_timesteps(50)
    for k in (0, 7, 49):                          # first, middle, last (prev_t < 0 -> final_alpha_cumprod) step
        t = int(sch.timesteps[k])
        z = rnd(1, B, HW, 4); eps = rnd(2, 2 * B, HW, 4); vn = rnd(6, B, HW, 4); g = 7.5
        e = eps[B:] + g * (eps[:B] - eps[B:])
        ref = sch.step(e, t, z, eta=eta, variance_noise=vn)
        assert not torch.allclose(ref, sch.step(e, t, z), atol=1e-3)
        kw = {}
        if inpaint:
            mask = (rnd(3, B, HW) > 0).float(); zi = rnd(4, B, HW, 4); nz = rnd(5, B, HW, 4)
            proper = zi if k == 49 else sch.add_noise(zi, nz, int(sch.timesteps[k + 1]))
            ref = (1 - mask[..., None]) * proper + mask[..., None] * ref
            kw = dict(mask=dev(mask), z_img=dev(zi), noise=dev(nz), a_next=None if k == 49 else float(sch.alphas_cumprod[int(sch.timesteps[k + 1])]))
        zd = dev(z.clone()); xn = torch.empty(2 * B, HW, 8, dtype=bf16, device="cuda")
        ops.ddim_cfg_step(zd, dev(eps), xn, guidance=g, a_t=mine.alpha(t), a_prev=mine.alpha_prev(t), var_noise=dev(vn), sigma=mine.sigma(t, eta), **kw)
        assert_close(zd, ref, atol=2e-4, rtol=1e-4, what=f"stochastic ddim z (step {k})")
        assert_close(xn[:B, :, :4], ref, atol=3e-2, what="next input")
    # the tensor API of the scheduler (NCHW), noise passed in and drawn from a generator
    x = rnd(7, 2, 4, 16, 24); mo = rnd(8, 2, 4, 16, 24); vn = rnd(9, 2, 4, 16, 24)
    t = int(sch.timesteps[20])
    got = mine.step(dev(mo), t, dev(x), eta=eta, variance_noise=dev(vn))[0]
    assert_close(got, sch.step(mo, t, x, eta=eta, variance_noise=vn), atol=2e-4, rtol=1e-4, what="scheduler.step(eta)")
    g1 = mine.step(dev(mo), t, dev(x), eta=eta, generator=torch.Generator("cuda").manual_seed(3))[0]
    g2 = mine.step(dev(mo), t, dev(x), eta=eta, generator=torch.Generator("cuda").manual_seed(3))[0]
    g3 = mine.step(dev(mo), t, dev(x), eta=eta, generator=torch.Generator("cuda").manual_seed(4))[0]
    assert torch.equal(g1, g2) and not torch.equal(g1, g3)
    with pytest.raises(ops.L.ImdError):
        ops.ddim_cfg_step(dev(rnd(1, B, HW, 4)), dev(eps), None, guidance=g, var_noise=dev(rnd(6, B, HW + 1, 4)), sigma=0.1)


@DTS
@pytest.mark.parametrize("B,H,W,Ca,Cb,half,ctrl", [(4, 16, 16, 1280, 640, True, False), (2, 32, 32, 640, 320, False, True), (8, 64, 64, 320, 320, True, False),
                                                 (2, 8, 8, 1280, 1280, False, False), (1, 20, 16, 640, 640, False, True), (2, 16, 16, 64, 64, False, False)])
def test_concat_that_also_writes_the_groupnorm_statistics(ops, dt, monkeypatch, B, H, W, Ca, Cb, half, ctrl):
    """The skip concatenation of an up block with the statistics of the resnet's norm1 behind it (imd_concat2_gn_stats): the concatenated
    tensor is what imd_concat2 writes, and GroupNorm + SiLU of it with the statistics that ride on it is BIT-IDENTICAL to the two-launch
    GroupNorm of the same tensor (same partials: same chunking, same order).  ``half``: the skip tensor holds one copy for both CFG halves;
    ``ctrl``: a ControlNet residual is added to the skip."""
    a = dev(rnd(1, B, H, W, Ca).to(dt))
    b = dev((rnd(2, B // 2 if half else B, H, W, Cb) * 1.5 + 0.25).to(dt))
    c = dev(rnd(3, B, H, W, Cb).to(dt)) if ctrl else None
    gamma, beta = dev(rnd(4, Ca + Cb) * 0.2 + 1.0), dev(rnd(5, Ca + Cb) * 0.1)
    plain = ops.concat_channels(a, b, c)
    assert getattr(plain, "_imd_gn_stats", None) is None
    fused = ops.concat_channels(a, b, c, gn_stats_groups=32)
    st = getattr(fused, "_imd_gn_stats", None)
    assert st is not None and st[2] == 32 and tuple(st[0].shape) == (B, st[1], 32, 2)
    assert torch.equal(plain, fused)
    want = ops.group_norm(plain, gamma, beta, groups=32, eps=1e-5, silu=True)          # statistics launch + normalise
    got = ops.group_norm(fused, gamma, beta, groups=32, eps=1e-5, silu=True)           # normalise only
    assert torch.equal(want, got)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(plain.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert_close(got, ref, what="groupnorm of the concatenation")
    monkeypatch.setattr(ops, "FUSED_CONCAT_STATS", False)
    off = ops.concat_channels(a, b, c, gn_stats_groups=32)
    assert getattr(off, "_imd_gn_stats", None) is None and torch.equal(off, plain)


@DTS
@pytest.mark.parametrize("M,N,K,rows", [(2048, 320, 320, 1024), (1024, 320, 320, 256), (512, 640, 640, 256), (1024, 320, 320, 64), (768, 192, 320, 256)])
def test_residual_that_repeats_over_the_batch(ops, dt, monkeypatch, M, N, K, rows):
    """A residual with fewer rows than the output is periodic (row m adds res[m % rows]: one copy of the block input for both halves of a CFG batch,
    imd_conv_gemm_params.res_rows).  The K = 320 row-resident projection reads it in place; every other launch gets it repeated first -- bit-identical
    to handing in the repeated tensor either way, and to the switch being off."""
    x, w, b = dev(rnd(1, M, K).to(dt)), dev((rnd(2, N, K) * K ** -0.5).to(dt)), dev(rnd(3, N) * 0.1)
    r = dev(rnd(4, rows, N).to(dt))
    full = r.repeat(M // rows, 1).contiguous()
    want = ops.linear(x, w, b, res=full)
    got = ops.linear(x, w, b, res=r)
    assert torch.equal(want, got)
    ref = x.float() @ w.float().t() + b + full.float()
    assert_close(got, ref, what="linear + periodic residual")
    monkeypatch.setattr(ops, "PERIODIC_RES", False)
    assert torch.equal(ops.linear(x, w, b, res=r), want)
    with pytest.raises(ops.L.ImdError):
        ops.linear(x, w, b, res=dev(rnd(5, rows - 8, N).to(dt)))          # not a divisor of M


@DTS
@pytest.mark.parametrize("M,N,K,rows", [(1024, 320, 320, 256), (512, 320, 320, 128), (512, 640, 640, 256)])
def test_periodic_residual_behind_the_layernorm_prologue(ops, dt, M, N, K, rows):
    """LayerNorm prologue + residual is the STAGED epilogue of the row-resident kernel: its periodic form (K = 320: the residual base shifted per
    workgroup; other K: repeated first by ops.conv_gemm) equals the repeated residual bit for bit, and the fp32 reference within the 16-bit bar."""
    x, w, b = dev(rnd(1, M, K).to(dt)), dev((rnd(2, N, K) * K ** -0.5).to(dt)), dev(rnd(3, N) * 0.1)
    r = dev(rnd(4, rows, N).to(dt))
    full = r.repeat(M // rows, 1).contiguous()
    want = ops.linear(x, w, b, res=full, ln_eps=1e-5)
    got = ops.linear(x, w, b, res=r, ln_eps=1e-5)
    assert torch.equal(want, got)
    xn = torch.nn.functional.layer_norm(x.float(), (K,), eps=1e-5)
    assert_close(got, xn @ w.float().t() + b + full.float(), what="LayerNorm + linear + periodic residual")


def test_periodic_residual_is_refused_where_no_kernel_reads_it(ops):
    """res_rows on a launch that is not the K = 320 row-resident projection is an error, not a silently ignored field."""
    L = ops.L
    M, N, K = 512, 640, 640
    x, w = dev(rnd(1, M, K).to(torch.bfloat16)), dev(rnd(2, N, K).to(torch.bfloat16))
    r, out = dev(rnd(3, 256, N).to(torch.bfloat16)), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    p = L.ConvGemmParams()
    p.dtype = 0
    p.x, p.w, p.out, p.res = x.data_ptr(), w.data_ptr(), out.data_ptr(), r.data_ptr()
    p.M, p.N, p.K, p.Cin, p.taps, p.Hin, p.Win, p.Hout, p.Wout, p.stride = M, N, K, K, 1, 1, M, 1, M, 1
    p.x_pix_stride, p.res_ld, p.out_ld, p.out_scale, p.split_k = K, N, N, 1.0, 1
    p.res_rows = 256
    for cfg in (2, 13):
        with pytest.raises(L.ImdError, match="res_rows|periodic"):
            L.check(L.load().imd_conv_gemm(ctypes.byref(p), cfg, None))
    p.K = p.Cin = p.x_pix_stride = 320
    p.N = p.res_ld = p.out_ld = 320
    p.res_rows = 192                                                          # not a multiple of the 128-row block
    with pytest.raises(L.ImdError, match="res_rows"):
        L.check(L.load().imd_conv_gemm(ctypes.byref(p), 12, None))


@DTS
@pytest.mark.parametrize("D,B,R,N,L2", [(80, 8, 4, 1024, 1024), (80, 4, 2, 320, 257), (160, 8, 4, 256, 256), (160, 2, 1, 64, 64), (160, 6, 3, 80, 200), (64, 4, 4, 128, 77), (80, 3, 1, 96, 96)])
def test_hybrid_attention_phase_split_is_bit_identical(ops, dt, monkeypatch, D, B, R, N, L2):
    """imd_attn_params.phase2_rows: the rows [0, R) carry the second key set; their two softmaxes run as separate workgroups of one launch and a follow-up
    launch adds the fp32 second result to the stored first one -- the arithmetic of the one-workgroup form, so the output is BIT-IDENTICAL; also against the
    fp32 reference of the hybrid attention (attention_processor.py:589-612)."""
    H = 4
    dpk, dpv = ops.attn_padded_dims(D)
    def r(seed, *s): return rnd(seed, *s)
    q = torch.zeros(B, H, N, dpk); q[..., :D] = r(1, B, H, N, D) * (D ** -0.5 * math.log2(math.e))
    k = torch.zeros(B, H, N, dpk); k[..., :D] = r(2, B, H, N, D)
    v = r(3, B, H, N, D)
    k2 = torch.zeros(1, H, L2, dpk); k2[..., :D] = r(4, 1, H, L2, D)
    v2 = r(5, 1, H, L2, D)
    def vt_of(vv, L):
        t = torch.zeros(vv.shape[0], H, dpv, ops.pad64(L)); t[:, :, :D, :L] = vv.transpose(-1, -2); return t
    qd, kd, vtd = dev(q.to(dt)), dev(k.to(dt)), dev(vt_of(v, N).to(dt))
    if dpk > D:
        kd[..., D] = 1.0
    k2d, v2td = dev(k2.to(dt)), dev(vt_of(v2, L2).to(dt))
    if dpk > D:
        k2d[..., D] = 1.0
    s2 = dev(torch.cat([torch.full((R,), 0.75), torch.zeros(B - R)]))
    kw = dict(B=B, H=H, N=N, D=D, L1=N, L1P=ops.pad64(N), k2=k2d, v2t=v2td, scale2=s2, L2=L2, L2P=ops.pad64(L2), kv2_bdiv=B, k_pad_one=True)
    one = torch.empty(B, N, H * D, dtype=dt, device="cuda")
    two = torch.empty_like(one)
    ops.attention(qd, kd, vtd, one, **kw)
    ops.attention(qd, kd, vtd, two, phase2_rows=R, **kw)
    assert torch.equal(one, two)
    monkeypatch.setattr(ops, "ATTN_PHASE_SPLIT", False)
    off = torch.empty_like(one)
    ops.attention(qd, kd, vtd, off, phase2_rows=R, **kw)
    assert torch.equal(one, off)
    qf, kf, vf = q.to(dt).float()[..., :D] / math.log2(math.e), k.to(dt).float()[..., :D], v.to(dt).float()
    a1 = torch.softmax(qf @ kf.transpose(-1, -2), -1) @ vf
    a2 = torch.softmax(qf @ k2.to(dt).float()[..., :D].transpose(-1, -2), -1) @ v2.to(dt).float()
    ref = a1 + s2.cpu().view(B, 1, 1, 1) * a2
    assert_close(two, ref.permute(0, 2, 1, 3).reshape(B, N, H * D), what="hybrid attention")


def test_phase_split_is_refused_where_no_kernel_runs_it(ops):
    L = ops.L
    p = L.AttnParams()
    t = torch.zeros(4096, dtype=torch.bfloat16, device="cuda")
    f = torch.zeros(4096, dtype=torch.float32, device="cuda")
    p.q = p.k1 = p.v1t = p.k2 = p.v2t = p.out = t.data_ptr()
    p.scale2 = f.data_ptr()
    p.B, p.H, p.N, p.D, p.L1, p.L1P, p.kv1_bdiv, p.L2, p.L2P, p.kv2_bdiv, p.out_ld = 2, 1, 64, 40, 64, 64, 1, 64, 64, 2, 40
    p.phase2_rows, p.phase2_out = 1, f.data_ptr()
    with pytest.raises(L.ImdError, match="phase2_rows"):
        L.check(L.load().imd_attention(ctypes.byref(p), None))           # head dim 40: the d = 40 kernel has no phase-split form
    p.D, p.out_ld, p.phase2_out = 80, 80, None
    with pytest.raises(L.ImdError, match="go together"):
        L.check(L.load().imd_attention(ctypes.byref(p), None))


@DTS
def test_add_concat_cast(ops, dt):
    a = rnd(1, 2, 50, 320).to(dt); b = rnd(2, 2, 50, 640).to(dt); c = rnd(3, 2, 50, 640).to(dt)
    assert_close(ops.add(dev(b), dev(c), 0.5), b.float() + 0.5 * c.float(), what="add")
    assert_close(ops.concat_channels(dev(a), dev(b)), torch.cat([a, b], -1), atol=0, rtol=0, what="concat")
    assert_close(ops.concat_channels(dev(a), dev(b), dev(c)), torch.cat([a.float(), b.float() + c.float()], -1), what="concat+add")
    f = rnd(4, 1000)
    assert_close(ops.f32_to_16(dev(f), dt), f.to(dt), atol=0, rtol=0, what="cast")


def test_no_cpu_fallback(ops):
    from imagdressing_amd._lib import ImdError
    with pytest.raises(ImdError):
        ops.linear(torch.zeros(8, 8, dtype=bf16), torch.zeros(8, 8, dtype=bf16))


def test_lincomb(ops):
    xs = [dev(rnd(i, 3, 1000, 4)) for i in range(5)]
    cs = [0.5, -1.25, 2.0, 0.0, 1e-3]
    out = ops.lincomb(list(zip(cs, xs)))
    ref = sum(c * x for c, x in zip(cs, xs))
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    ops.lincomb([(2.0, xs[0]), (1.0, xs[1])], out=xs[0])            # in place
    assert torch.allclose(xs[0], 2.0 * dev(rnd(0, 3, 1000, 4)) + xs[1], rtol=1e-6, atol=1e-6)
    odd = [dev(rnd(9, 1001)), dev(rnd(10, 1001))]                    # numel % 4 != 0: scalar tail
    assert torch.allclose(ops.lincomb([(1.0, odd[0]), (-3.0, odd[1])]), odd[0] - 3.0 * odd[1], rtol=1e-6, atol=1e-6)
    with pytest.raises(ops.L.ImdError):
        ops.lincomb([(1.0, xs[0])] * 9)
