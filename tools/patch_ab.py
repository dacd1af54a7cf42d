"""A/B of the halo-patch 3x3 conv kernel's two staging paths on the ResNet conv shapes of the 512x512 CFG batch (interleaved
repeats): every operand by LDS-DMA (default) vs through registers (knob 2 bit 9)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagdressing_amd import ops

SHAPES = [("L0 320->320", 8, 64, 64, 320, 320, 1), ("L0 640->320", 8, 64, 64, 640, 320, 1), ("L0 960->320", 8, 64, 64, 960, 320, 1),
          ("L1 320->640", 8, 32, 32, 320, 640, 1), ("L1 640->640", 8, 32, 32, 640, 640, 2), ("L1 1280->640", 8, 32, 32, 1280, 640, 2),
          ("L1 1920->640", 8, 32, 32, 1920, 640, 2), ("L2 1280->1280", 8, 16, 16, 1280, 1280, 4), ("L2 2560->1280", 8, 16, 16, 2560, 1280, 4),
          ("L0 b1 320->320", 2, 64, 64, 320, 320, 1), ("L0 512x640 320->320", 8, 80, 64, 320, 320, 1)]
lib = ops.L.load()

def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

dt = torch.bfloat16
for name, B, H, W, Cin, Cout, split in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt)
    bias = torch.randn(Cout, device="cuda")
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    row = dict(shape=name, split=split)
    outs = {}
    for rep in range(2):
        for nm, flags in (("dma", 23), ("regs", 23 | 512)):
            ops.L.check(lib.imd_set_tuning(2, flags))
            us = timed(lambda: ops.conv2d_nhwc(x, w, bias, cfg=5, split_k=split))
            row[nm + "_us"] = round(min(row.get(nm + "_us", 1e9), us), 1)
            outs[nm] = ops.conv2d_nhwc(x, w, bias, cfg=5, split_k=split)
    ops.L.check(lib.imd_set_tuning(2, 23))
    row["dma_tf"] = round(fl / row["dma_us"] / 1e6, 1); row["regs_tf"] = round(fl / row["regs_us"] / 1e6, 1)
    row["identical"] = bool(torch.equal(outs["dma"], outs["regs"]))
    print(json.dumps(row), flush=True)
