"""MX-FP8 hybrid attention (BASELINE.json configs[4]: 768x576 ControlNet-inpainting "with fp8 MFMA attention"): the quantisers
bit for bit against torch's OCP e4m3 conversion, and the kernel at the configuration's level-0 shape (N = M = 6912, head dim 40)
against fp32 oracles -- one on the DEQUANTISED operands (isolates the kernel: only P's e4m3 rounding remains) and one on the
original 16-bit operands (the whole error budget of running this layer in fp8)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from imagdressing_amd import ops as o
    return o


def rnd(seed, *shape, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def e4m3_bytes(x):
    return x.to(torch.float8_e4m3fn).view(torch.uint8)


def key_perm(LP):
    """position of key k inside V8^T rows: 64-groups, 32 ((k >> 3) & 1) + 8 ((k >> 4) & 3) + (k & 7)"""
    k = torch.arange(LP)
    return (k & ~63) + 32 * ((k >> 3) & 1) + 8 * ((k >> 4) & 3) + (k & 7)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_fp8_quantisers_bit_exact(ops, dt):
    rows = rnd(1, 3, 8, 100, 48).to(dt)
    got = ops.quantize_fp8_rows(rows.cuda(), 4, pad_val=64.0).cpu()
    want = torch.zeros(3, 8, 100, 64, dtype=torch.uint8)
    want[..., :40] = e4m3_bytes(rows[..., :40].float() * 16.0)
    want[..., 40:42] = e4m3_bytes(torch.tensor(64.0))
    assert torch.equal(got, want)
    vt = rnd(2, 2, 8, 64, 192).to(dt)
    got = ops.quantize_fp8_vt(vt.cuda(), 3).cpu()
    want = torch.zeros(2, 8, 64, 192, dtype=torch.uint8)
    want[:, :, :40, key_perm(192)] = e4m3_bytes(vt[:, :, :40].float() * 8.0)
    assert torch.equal(got[:, :, :40], want[:, :, :40])


def dequant(b8, exp):
    return b8.cpu().view(torch.float8_e4m3fn).float() * 2.0 ** -exp


def oracle(q, k, v, kr, vr, s2, D, dt):
    """q [B,H,N,D], k [B,H,L,D], v [B,H,L,D] fp32 (already carrying every scale), base-2 softmax; phase 0 rounded to dt"""
    B, H, N, _ = q.shape
    out = torch.empty(B, N, H * D)
    ln2 = math.log(2.0)
    for b in range(B):
        for h in range(H):
            qq = q[b, h] * ln2
            o = (torch.softmax(qq @ k[b, h].t(), -1) @ v[b, h]).to(dt).float()
            if s2[b] != 0:
                o = o + s2[b] * (torch.softmax(qq @ kr[0, h].t(), -1) @ vr[0, h])
            out[b, :, h * D:(h + 1) * D] = o
    return out


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("N,M", [(6912, 6912), (1000, 700)], ids=["configs4-level0", "ragged"])
def test_fp8_attention_vs_oracles(ops, dt, N, M):
    D, H, B = 40, 8, 2
    e = ops.FP8_EXPS
    sc = D ** -0.5 * math.log2(math.e)
    q16 = torch.zeros(B, H, N, 48); q16[..., :D] = rnd(1, B, H, N, D) * sc * 2.5           # logits with std 2.5: peaked rows
    k16 = torch.zeros(B, H, N, 48); k16[..., :D] = rnd(2, B, H, N, D)
    v16 = rnd(3, B, H, N, D); kr16 = torch.zeros(1, H, M, 48); kr16[..., :D] = rnd(4, 1, H, M, D); vr16 = rnd(5, 1, H, M, D)
    q16, k16, kr16, v16, vr16 = (t.to(dt) for t in (q16, k16, kr16, v16, vr16))
    LPn, LPm = ops.pad64(N), ops.pad64(M)

    def vt_of(v, L, LP):
        t = torch.zeros(v.shape[0], H, 64, LP, dtype=dt)
        t[:, :, :D, :L] = v.transpose(-1, -2)
        return t
    q8 = ops.quantize_fp8_rows(q16.cuda(), e["q"])
    k8 = ops.quantize_fp8_rows(k16.cuda(), e["k"], pad_val=2.0 ** (e["q"] + e["k"]))
    kr8 = ops.quantize_fp8_rows(kr16.cuda(), e["k"], pad_val=2.0 ** (e["q"] + e["k"]))
    v8 = ops.quantize_fp8_vt(vt_of(v16, N, LPn).cuda(), e["v"])
    vr8 = ops.quantize_fp8_vt(vt_of(vr16, M, LPm).cuda(), e["v"])
    s2 = torch.tensor([0.9, 0.0])
    out = torch.empty(B, N, H * D, dtype=dt, device="cuda")
    ops.attention_fp8(q8, k8, v8, out, B=B, H=H, N=N, L1=N, L1P=LPn, k2=kr8, v2t=vr8, scale2=s2.cuda(), L2=M, L2P=LPm, kv2_bdiv=B)
    out = out.float().cpu()
    assert torch.isfinite(out).all()
    # (1) oracle on the dequantised operands: what remains is P's e4m3 rounding (3 mantissa bits) and fp32 summation order
    inv = torch.empty_like(key_perm(LPn)); inv[key_perm(LPn)] = torch.arange(LPn)

    def v_from(v8t, L, LP):
        vt = dequant(v8t, e["v"])[:, :, :D]                      # [.., D, LP] in permuted key order
        return vt[..., key_perm(LP)][..., :L].transpose(-1, -2)   # position p holds key inverse(p): read position of key k
    ref_q = oracle(dequant(q8, e["q"])[..., :D], dequant(k8, e["k"])[..., :D], v_from(v8, N, LPn), dequant(kr8, e["k"])[..., :D],
                   v_from(vr8, M, LPm), s2, D, dt)
    # (2) oracle on the 16-bit operands: the error budget of fp8 for this layer
    ref_16 = oracle(q16.float()[..., :D], k16.float()[..., :D], v16.float(), kr16.float()[..., :D], vr16.float(), s2, D, dt)

    def stats(ref):
        err = (out - ref)
        return dict(rel_rms=(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), max_abs=err.abs().max().item(),
                    ref_rms=ref.pow(2).mean().sqrt().item())
    sq, s16 = stats(ref_q), stats(ref_16)
    print("fp8 attention", dt, N, M, "vs dequantised-operand oracle", sq, "| vs 16-bit-operand oracle", s16)
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fp8_attention_error_budget.jsonl", "a") as f:
        f.write(json.dumps(dict(dtype=str(dt), N=N, M=M, logit_std=2.5, vs_dequantised_operands=sq, vs_16bit_operands=s16)) + "\n")
    assert sq["rel_rms"] < 2.5e-2, sq           # kernel correctness: only P's 3-bit mantissa separates it from this oracle
    # whole budget of e4m3 Q, K, V and P at logits of std 2.5 (measured 8.6 % rms: a 3-bit mantissa carries a logit of
    # magnitude s only to about 0.03 s); DESIGN.md section 3 discusses what that means for configs[4]
    assert s16["rel_rms"] < 0.12, s16
