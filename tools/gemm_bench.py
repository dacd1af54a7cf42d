"""Micro-benchmark of the implicit-GEMM kernel on the SD1.5 UNet's layer shapes (CFG batch of 8 rows
at 512x512).  Prints TFLOP/s per (shape, tile config); used to pick tile heuristics / find slow shapes."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

SHAPES = [
    # name, B, H, W, Cin, Cout, taps, stride, act
    ("L0 conv3x3 320->320", 8, 64, 64, 320, 320, 9, 1, 0),
    ("L0 conv3x3 640->320", 8, 64, 64, 640, 320, 9, 1, 0),
    ("L0 conv3x3 960->320", 8, 64, 64, 960, 320, 9, 1, 0),
    ("L0 lin 320->960 (qkv)", 8, 64, 64, 320, 960, 1, 1, 0),
    ("L0 lin 320->320", 8, 64, 64, 320, 320, 1, 1, 0),
    ("L0 lin 320->2560 geglu", 8, 64, 64, 320, 2560, 1, 1, 2),
    ("L0 lin 1280->320", 8, 64, 64, 1280, 320, 1, 1, 0),
    ("L0->1 conv s2 320->320", 8, 64, 64, 320, 320, 9, 2, 0),
    ("L1 conv3x3 640->640", 8, 32, 32, 640, 640, 9, 1, 0),
    ("L1 conv3x3 320->640", 8, 32, 32, 320, 640, 9, 1, 0),
    ("L1 conv3x3 1280->640", 8, 32, 32, 1280, 640, 9, 1, 0),
    ("L1 conv3x3 1920->640", 8, 32, 32, 1920, 640, 9, 1, 0),
    ("L1 lin 640->1920 (qkv)", 8, 32, 32, 640, 1920, 1, 1, 0),
    ("L1 lin 640->640", 8, 32, 32, 640, 640, 1, 1, 0),
    ("L1 lin 640->5120 geglu", 8, 32, 32, 640, 5120, 1, 1, 2),
    ("L1 lin 2560->640", 8, 32, 32, 2560, 640, 1, 1, 0),
    ("L2 conv3x3 1280->1280", 8, 16, 16, 1280, 1280, 9, 1, 0),
    ("L2 conv3x3 640->1280", 8, 16, 16, 640, 1280, 9, 1, 0),
    ("L2 conv3x3 2560->1280", 8, 16, 16, 2560, 1280, 9, 1, 0),
    ("L2 conv3x3 1920->1280", 8, 16, 16, 1920, 1280, 9, 1, 0),
    ("L2 lin 1280->3840 (qkv)", 8, 16, 16, 1280, 3840, 1, 1, 0),
    ("L2 lin 1280->1280", 8, 16, 16, 1280, 1280, 1, 1, 0),
    ("L2 lin 1280->10240 geglu", 8, 16, 16, 1280, 10240, 1, 1, 2),
    ("L2 lin 5120->1280", 8, 16, 16, 5120, 1280, 1, 1, 0),
    ("L3 conv3x3 1280->1280", 8, 8, 8, 1280, 1280, 9, 1, 0),
    ("L3 conv3x3 2560->1280", 8, 8, 8, 2560, 1280, 9, 1, 0),
    ("L3 lin 1280->10240 geglu", 8, 8, 8, 1280, 10240, 1, 1, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0,1,2")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--flags", default="23", help="comma list of GEMM tuning flag values to A/B (knob 2)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    cfgs = [int(c) for c in a.cfgs.split(",")]
    res = []
    for name, B, H, W, Cin, Cout, taps, stride, act in SHAPES:
        if a.filter and a.filter not in name:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, taps * Cin, device="cuda") * (taps * Cin) ** -0.5).to(dt)
        bias = torch.randn(Cout, device="cuda")
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        M = B * Ho * Wo
        flops = 2.0 * M * Cout * taps * Cin
        lib = ops.L.load()
        row = dict(shape=name, M=M, N=Cout, K=taps * Cin, auto=lib.imd_conv_gemm_auto_cfg(M, Cout),
                   auto_split=lib.imd_conv_gemm_auto_split(M, Cout, taps * Cin, -1) if act != 2 else 1)
        for cfg, fl in [(c, int(f)) for f in a.flags.split(",") for c in cfgs + [-1]]:
            lib.imd_set_tuning(2, fl)
            try:
                sk = 0 if cfg == -1 else 1
                for _ in range(3):
                    ops.conv2d_nhwc(x, w, bias, taps=taps, stride=stride, act=act, cfg=cfg, split_k=sk)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv2d_nhwc(x, w, bias, taps=taps, stride=stride, act=act, cfg=cfg, split_k=sk)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                row[f"cfg{cfg}_f{fl}_us"] = round(us, 1)
                row[f"cfg{cfg}_f{fl}_tf"] = round(flops / us / 1e6, 1)
            except Exception as ex:   # noqa
                row[f"cfg{cfg}_err"] = str(ex)[:80]
        res.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
