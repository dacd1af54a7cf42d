"""Per-shape timings of the tile kernels for same-box A/Bs of two library builds (run once per build, IMD_LIB_PATH selects it):
    python tools/kernel_ab.py --what conv|linear [--cfg N] [--dtype bf16] [--iters 30]
3x3 convolutions of the bench step on tile config 5 (or --cfg) and the plain linears that run on tile config 17; rotating operand
sets (cold L2), HIP-event timed; one JSON line per shape."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

CONVS = [  # (name, B, H, W, Cin, Cout, split_k)
    ("L0 320->320", 8, 64, 64, 320, 320, 1), ("L0 640->320", 8, 64, 64, 640, 320, 1), ("L0 960->320", 8, 64, 64, 960, 320, 1),
    ("L1 320->640", 8, 32, 32, 320, 640, 1), ("L1 640->640", 8, 32, 32, 640, 640, 1), ("L1 1280->640 s2", 8, 32, 32, 1280, 640, 2),
    ("L1 1920->640 s2", 8, 32, 32, 1920, 640, 2), ("L2 1280->1280 s4", 8, 16, 16, 1280, 1280, 4), ("L2 2560->1280 s4", 8, 16, 16, 2560, 1280, 4),
    ("b1 L0 320->320", 2, 64, 64, 320, 320, 1), ("512x640 L0 320->320", 8, 80, 64, 320, 320, 1),
]
LINEARS = [  # (name, M, N, K, geglu)
    ("L1 ff-out 8192x640x2560", 8192, 640, 2560, False), ("L1 geglu 8192x5120x640", 8192, 5120, 640, True),
    ("L2 geglu 2048x10240x1280", 2048, 10240, 1280, True), ("L2 ff-out 2048x1280x5120", 2048, 1280, 5120, False),
    ("L1 qkv-like 8192x1920x640", 8192, 1920, 640, False), ("L0 32768x320x320", 32768, 320, 320, False),
]
ap = argparse.ArgumentParser()
ap.add_argument("--what", default="conv"); ap.add_argument("--cfg", type=int, default=-2); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=30); ap.add_argument("--epi", action="store_true", help="conv: the ResNet epilogue -- time-embedding vector, residual, GroupNorm statistics of the output"); ap.add_argument("--tag", default=os.environ.get("IMD_LIB_PATH", "cur").split("_")[-1].replace(".so", ""))
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16


def timed(fn, n):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


if a.what == "conv":
    cfg = 5 if a.cfg == -2 else a.cfg
    for name, B, H, W, Cin, Cout, sk in CONVS:
        xs = [torch.randn(B, H, W, Cin, device="cuda").to(dt) for _ in range(4)]
        ws = [(torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt) for _ in range(4)]
        b = torch.randn(Cout, device="cuda")
        kw = {}
        if a.epi:
            kw = dict(rowvec=torch.randn(B, Cout, device="cuda"), rowvec_stride=Cout, res=torch.randn(B, H, W, Cout, device="cuda").to(dt), gn_stats_groups=32)
        us = timed(lambda i: ops.conv2d_nhwc(xs[i % 4], ws[i % 4], b, cfg=cfg, split_k=sk, **kw), a.iters)
        print(json.dumps(dict(lib=a.tag, shape=name, cfg=cfg, us=round(us, 1), tflops=round(2.0 * B * H * W * Cout * 9 * Cin / us / 1e6, 1))), flush=True)
else:
    cfg = 17 if a.cfg == -2 else a.cfg
    for name, M, N, K, geglu in LINEARS:
        xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(4)]
        ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(4)]
        b = torch.randn(N, device="cuda")
        us = timed(lambda i: ops.linear(xs[i % 4], ws[i % 4], b, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, cfg=cfg, split_k=1), a.iters)
        print(json.dumps(dict(lib=a.tag, shape=name, cfg=cfg, us=round(us, 1), tflops=round(2.0 * M * N * K / us / 1e6, 1))), flush=True)
