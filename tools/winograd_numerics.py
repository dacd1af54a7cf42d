#!/usr/bin/env python
"""Step 0 of the Winograd F(2x2, 3x3) question (round-5 review item 2): NUMERICS on the CPU, no GPU time.

Emulates what an MFMA Winograd kernel would compute for the stride-1 3x3 convolutions of the 64x64 / 32x32 levels -- 16-bit inputs,
input transform V = B^T d B in fp32 rounded to 16 bits (the MFMA operand), weights transformed offline U = G g G^T in fp32 from the fp32
master weights and rounded to 16 bits, fp32 accumulation over channels, output transform A^T M A in fp32 -- against the direct form the
shipped kernels compute (16-bit inputs and weights, fp32 accumulation) and against F.conv2d in fp32 on the same 16-bit inputs.
Prints one JSON line per case (dtype x input distribution x shape).  Reference layers: diffusers-0.24 ResnetBlock2D.conv1 / conv2
(SURVEY 8a A12)."""
import json
import sys

import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv(x16, w32, dt, round_u=True, round_v=True):
    """x16 [B, C, H, W] already rounded to `dt` (held as fp32); w32 [O, C, 3, 3] fp32 master weights.  H, W even."""
    B, C, H, W = x16.shape
    xp = F.pad(x16, (1, 1, 1, 1))
    # 4x4 input tiles with stride 2: [B, C, H/2, W/2, 4, 4]
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = BT @ t @ BT.t()                                   # fp32 from exact 16-bit values: sums of 4 -> exact in fp32
    if round_v:
        V = V.to(dt).float()
    U = G @ w32.unsqueeze(0).reshape(-1, 3, 3).reshape(w32.shape[0], C, 3, 3) @ G.t()      # [O, C, 4, 4]
    if round_u:
        U = U.to(dt).float()
    # M[b, o, ty, tx, i, j] = sum_c U[o, c, i, j] * V[b, c, ty, tx, i, j]   (16 batched GEMMs, fp32 accumulate)
    M = torch.einsum("ocij,bcyxij->boyxij", U, V)
    Y = AT @ M @ AT.t()                                   # [B, O, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w32.shape[0], H, W)


def case(dt, dist, C, O, HW, seed=0, B=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, HW, HW, generator=g)
    if dist == "silu":                                     # what conv1 / conv2 actually see: SiLU of a normalised tensor (mean 0.21, sigma 0.56)
        x = F.silu(x)
    w = torch.randn(O, C, 3, 3, generator=g) * (9 * C * x.pow(2).mean().item()) ** -0.5      # output sigma ~ 1
    x16 = x.to(dt).float()
    ref = F.conv2d(x16.double(), w.double(), padding=1).float()                            # exact conv of the 16-bit inputs with the master weights
    direct = F.conv2d(x16, w.to(dt).float(), padding=1)                                  # the shipped kernels: 16-bit weights, fp32 accumulate
    wino = winograd_conv(x16, w, dt)
    wino_v = winograd_conv(x16, w, dt, round_u=False)                                    # attribution: only V rounded
    wino_u = winograd_conv(x16, w, dt, round_v=False)                                    # only U rounded
    st = lambda y: {"max": round((y - ref).abs().max().item(), 5), "rms": round((y - ref).pow(2).mean().sqrt().item(), 6)}
    return {"dtype": str(dt).replace("torch.", ""), "input": dist, "Cin": C, "Cout": O, "map": HW, "out_sigma": round(ref.std().item(), 3),
            "direct_vs_exact": st(direct), "winograd_vs_exact": st(wino), "winograd_only_V_rounded": st(wino_v), "winograd_only_U_rounded": st(wino_u),
            "winograd_vs_direct_max": round((wino - direct).abs().max().item(), 5)}


if __name__ == "__main__":
    torch.set_num_threads(8)
    for dt in (torch.float16, torch.bfloat16):
        for dist in ("normal", "silu"):
            for (C, O, HW) in ((320, 320, 64), (640, 640, 32), (960, 320, 64)):
                print(json.dumps(case(dt, dist, C, O, HW)), flush=True)
