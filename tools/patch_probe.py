"""Halo-patch conv on the three map sizes it serves at the bench batch (CFG batch 8): one timing per shape (HIP events,
rotating operands).  Used with timing-only library variants (IMD_LIB_PATH) to separate the memory side from the matrix side."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
dt = torch.bfloat16
_w = torch.randn(4096, 4096, device="cuda").to(dt)
for _ in range(300): _w @ _w
torch.cuda.synchronize()
CFGS = [int(c) for c in os.environ.get("PATCH_PROBE_CFGS", "5").split(",")]
SPLITS = [int(c) for c in os.environ.get("PATCH_PROBE_SPLITS", "1").split(",")]
out = {}
for name, (B, H, W, Cin, Cout) in {"L0 320->320": (8, 64, 64, 320, 320), "L0 640->320": (8, 64, 64, 640, 320), "L0 960->320": (8, 64, 64, 960, 320), "L1 640->640": (8, 32, 32, 640, 640),
                                   "L1 1280->640": (8, 32, 32, 1280, 640), "L2 1280->1280": (8, 16, 16, 1280, 1280),
                                   "VAE 256x256 256->256": (2, 256, 256, 256, 256), "VAE 512x512 128->128": (1, 512, 512, 128, 128)}.items():
    xs = [torch.randn(B, H, W, Cin, device="cuda").to(dt) for _ in range(4)]
    ws = [(torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt) for _ in range(4)]
    bias = torch.randn(Cout, device="cuda")
    if os.environ.get("PATCH_PROBE_ZERO"):            # zero operands: same instruction stream, no data toggling (power probe)
        for t in xs + ws: t.zero_()
    for cfg in CFGS:
      for sk in SPLITS:
        i = [0]
        def f():
            j = i[0] % 4; i[0] += 1
            return ops.conv2d_nhwc(xs[j], ws[j], bias, taps=9, stride=1, cfg=cfg, split_k=sk)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 100
        out[f"{name} c{cfg}s{sk}"] = (round(us, 1), round(2.0 * B * H * W * Cout * 9 * Cin / us / 1e6))
    del xs, ws
print(json.dumps(out))
