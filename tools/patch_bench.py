"""A/B on the GPU: GroupNorm+SiLU -> conv3x3 as (two GN kernels + gather GEMM) versus (GN statistics + halo-patch conv
with the normalisation fused into its staging).  One JSON line per ResNet conv shape of the 512x512 UNet at batch 8."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagdressing_amd import ops  # noqa: E402

SHAPES = [  # (name, B, H, W, Cin, Cout)
    ("L0 320->320", 8, 64, 64, 320, 320), ("L0 640->320", 8, 64, 64, 640, 320), ("L0 960->320", 8, 64, 64, 960, 320),
    ("L1 320->640", 8, 32, 32, 320, 640), ("L1 640->640", 8, 32, 32, 640, 640), ("L1 1280->640", 8, 32, 32, 1280, 640),
    ("L1 1920->640", 8, 32, 32, 1920, 640), ("L1 960->640", 8, 32, 32, 960, 640),
    ("L2 640->1280", 8, 16, 16, 640, 1280), ("L2 1280->1280", 8, 16, 16, 1280, 1280), ("L2 2560->1280", 8, 16, 16, 2560, 1280),
]


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    for name, B, H, W, Cin, Cout in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt)
        bias = torch.randn(Cout, device="cuda")
        gamma = torch.ones(Cin, device="cuda"); beta = torch.zeros(Cin, device="cuda")
        flops = 2.0 * B * H * W * Cout * 9 * Cin
        row = dict(shape=name)
        row["gn_us"] = round(timed(lambda: ops.group_norm(x, gamma, beta, silu=True), a.iters), 1)
        row["coef_us"] = round(timed(lambda: ops.group_norm_coeffs(x, gamma, beta), a.iters), 1)
        row["conv_auto_us"] = round(timed(lambda: ops.conv2d_nhwc(x, w, bias), a.iters), 1)
        for sk in (1, 2, 4):
            try:
                row[f"patch_s{sk}_us"] = round(timed(lambda: ops.conv2d_nhwc(x, w, bias, cfg=5, split_k=sk), a.iters), 1)
            except Exception as ex:  # noqa
                row[f"patch_s{sk}_err"] = str(ex)[:60]
        ca, cb = ops.group_norm_coeffs(x, gamma, beta)
        for sk in (1, 2, 4):
            row[f"patch_gn_s{sk}_us"] = round(timed(lambda: ops.conv2d_nhwc(x, w, bias, cfg=5, split_k=sk, gn=(ca, cb, True)), a.iters), 1)
        best_patch = min(row[f"patch_gn_s{sk}_us"] for sk in (1, 2, 4))
        row["unfused_total_us"] = round(row["gn_us"] + row["conv_auto_us"], 1)
        row["fused_total_us"] = round(row["coef_us"] + best_patch, 1)
        row["conv_auto_tf"] = round(flops / row["conv_auto_us"] / 1e6, 1)
        row["patch_tf"] = round(flops / min(row[f"patch_s{sk}_us"] for sk in (1, 2, 4)) / 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
