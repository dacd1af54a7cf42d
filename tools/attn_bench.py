"""Micro-benchmark of the fused hybrid-attention kernel at the UNet level-0 shape of the 512x512 CFG batch
(4 cond rows with the garment branch + 4 uncond rows, N = M = 4096, 8 heads, d = 40) and the other levels.
Also the target of the PMC passes in tools/gpu.sh pmc (HBM traffic, MFMA busy cycles)."""
import argparse, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def run(D, N, M, Bimg, iters, dt, qw=None, xcd=1, zero=False, fp8=False, phase_split=False):
    H = 8
    B = 2 * Bimg
    dpk, dpv = ops.attn_padded_dims(D)
    g = torch.Generator(device="cuda").manual_seed(0)
    def r(*s): return torch.randn(*s, generator=g, device="cuda").to(dt) * (0.0 if zero else 1.0)
    q = torch.zeros(B, H, N, dpk, dtype=dt, device="cuda"); q[..., :D] = r(B, H, N, D) * (D ** -0.5 * math.log2(math.e))
    k = ops.k_buffer((B, H, N, dpk), D, dt, "cuda"); k[..., :D] = r(B, H, N, D)
    vt = torch.zeros(B, H, dpv, ops.pad64(N), dtype=dt, device="cuda"); vt[:, :, :D, :N] = r(B, H, D, N)
    kr = ops.k_buffer((1, H, M, dpk), D, dt, "cuda"); kr[..., :D] = r(1, H, M, D)
    vr = torch.zeros(1, H, dpv, ops.pad64(M), dtype=dt, device="cuda"); vr[:, :, :D, :M] = r(1, H, D, M)
    s2 = torch.cat([torch.ones(Bimg), torch.zeros(Bimg)]).cuda()
    out = torch.empty(B, N, H * D, dtype=dt, device="cuda")
    if qw is not None:
        ops.L.check(ops.L.load().imd_set_tuning(0, qw))
    ops.L.check(ops.L.load().imd_set_tuning(1, xcd))
    if fp8:
        e = ops.FP8_EXPS
        q8 = ops.quantize_fp8_rows(q, e["q"]); k8 = ops.quantize_fp8_rows(k, e["k"], 2.0 ** (e["q"] + e["k"]))
        kr8 = ops.quantize_fp8_rows(kr, e["k"], 2.0 ** (e["q"] + e["k"]))
        v8 = ops.quantize_fp8_vt(vt, e["v"]); vr8 = ops.quantize_fp8_vt(vr, e["v"])

    def go():
        if fp8:
            ops.attention_fp8(q8, k8, v8, out, B=B, H=H, N=N, L1=N, L1P=ops.pad64(N), k2=kr8, v2t=vr8, scale2=s2, L2=M, L2P=ops.pad64(M), kv2_bdiv=B)
            return
        ops.attention(q, k, vt, out, B=B, H=H, N=N, D=D, L1=N, L1P=ops.pad64(N), k2=kr, v2t=vr, scale2=s2, L2=M, L2P=ops.pad64(M), kv2_bdiv=B, k_pad_one=True,
                      phase2_rows=Bimg if phase_split else 0)
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    C = H * D
    fl = Bimg * (4.0 * N * N * C + 4.0 * N * M * C) + Bimg * 4.0 * N * N * C
    # algorithmic HBM bytes: Q,K read + V^T read (D rows) per row, garment K/V once per (cond row, head) from L2/HBM, O written
    alg_bytes = 2 * (B * H * N * dpk * 2) + B * H * D * N * 2 + H * (M * dpk + D * M) * 2 + B * N * C * 2
    extra = {}
    if qw in (31, 32):
        out.zero_(); go(); torch.cuda.synchronize()
        c = out.view(torch.int64).flatten()[:6].tolist()
        waves, steps = max(c[5], 1), max(c[4], 1)
        extra = dict(cycles_per_step=dict(loop_total=round(c[0] / steps, 1), slots=round(c[1] / steps, 1), check=round(c[2] / steps, 1),
                                          sync_per_step=round(c[3] / steps, 1)), steps_per_wave=round(steps / waves, 1), waves=waves)
    return dict(extra, D=D, N=N, M=M, Bimg=Bimg, qw=qw, xcd=xcd, us=round(us, 1), tflops=round(fl / us / 1e6, 1), flops=fl, alg_bytes=alg_bytes)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only-l0", action="store_true")
    ap.add_argument("--variants", default="", help="comma list of knob-0 values to A/B on the level-0 shape")
    ap.add_argument("--cycles", action="store_true", help="variants 31 / 32 (-DIMD_ABLATIONS builds): print the in-kernel s_memtime counters (per wave and 32-key step)")
    ap.add_argument("--fp8", action="store_true", help="also time imd_attention_fp8 (operands quantised outside the timed region)")
    ap.add_argument("--N", type=int, default=4096, help="tokens of the level-0 shape (6912 = the 768x576 configuration)")
    ap.add_argument("--zero", action="store_true", help="all-zero Q/K/V: same instruction stream, far fewer toggling bits (clock / power probe)")
    ap.add_argument("--default-only", action="store_true", help="level-0 shape with the library's default knobs only (PMC passes)")
    ap.add_argument("--phase-split", action="store_true", help="with --level: A/B the phase-split launch (imd_attn_params.phase2_rows) against the one-workgroup form")
    ap.add_argument("--level", default="", help="D,N: only this level's shape with the library's default knobs (PMC passes of the generic kernel: 80,1024 / 160,256)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    cases = [(40, 4096, 4096, 4, 2, 1), (40, 4096, 4096, 4, 1, 1), (40, 4096, 4096, 4, 2, 0), (40, 4096, 4096, 4, 1, 0)]
    if a.variants:
        cases = [(40, a.N, a.N, 4, int(v), 1) for v in a.variants.split(",")]
    elif a.level:
        d_, n_ = (int(v) for v in a.level.split(","))
        cases = [(d_, n_, n_, 4, None, 1)]
    elif a.default_only:
        cases = [(40, 4096, 4096, 4, None, 1)]
    elif not a.only_l0:
        cases += [(80, 1024, 1024, 4, None, 1), (80, 1024, 1024, 4, None, 0), (160, 256, 256, 4, None, 1), (160, 64, 64, 4, None, 1)]
    for rep in range(2):          # interleaved repeats: within-run A/B
        for D, N, M, Bi, qw, xcd in cases:
            print(json.dumps(dict(run(D, N, M, Bi, a.iters, dt, qw, xcd, a.zero), zero=a.zero)), flush=True)
            if a.phase_split and a.level:
                print(json.dumps(dict(run(D, N, M, Bi, a.iters, dt, qw, xcd, a.zero, phase_split=True), phase_split=True)), flush=True)
        if a.fp8:
            print(json.dumps(dict(run(40, a.N, a.N, 4, a.iters, dt, None, 1, a.zero, fp8=True), fp8=True)), flush=True)
    ops.L.load().imd_set_tuning(0, 10); ops.L.load().imd_set_tuning(1, 1)
