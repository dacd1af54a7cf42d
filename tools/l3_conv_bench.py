"""8x8- and 16x16-level 3x3 convolutions of the CFG batch (16 rows at batch 8): every tile config x K split, time of the conv
launch + its split-K finish launch (HIP events, rotating inputs so weights come from HBM/MALL, not from a warm L2)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", default="0,1,2,3,5,18,24")
ap.add_argument("--splits", default="1,2,3,4,6,8,9,12,18")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--shapes", default="L3a,L3b,L2a")
a = ap.parse_args()
SH = {"L3c": (8, 8, 8, 2560, 1280), "L3one": (2, 8, 8, 1280, 1280), "L3a": (16, 8, 8, 1280, 1280), "L3b": (16, 8, 8, 2560, 1280), "L2a": (16, 16, 16, 1280, 1280), "L2b": (16, 16, 16, 2560, 1280),
      "L3s": (8, 8, 8, 1280, 1280)}
dt = torch.bfloat16
def linear_case(M, N, K):
    """the same product without the gather (plain [M, K] x [N, K]^T through tile config 17): what the gather costs"""
    xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(4)]
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(4)]
    row = dict(shape=f"linear {M}x{N}x{K}", tf_per_us=round(2.0 * M * N * K / 1e6, 1))
    for cfg in (17, 0, 1, 16):
        for sk in [int(s) for s in a.splits.split(",")]:
            try:
                i = [0]
                def f():
                    j = i[0] % 4; i[0] += 1
                    return ops.conv_gemm(xs[j], ws[j], M=M, N=N, Cin=K, cfg=cfg, split_k=sk)
                for _ in range(4): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters): f()
                e1.record(); torch.cuda.synchronize()
                row[f"c{cfg}s{sk}"] = round(e0.elapsed_time(e1) * 1e3 / a.iters, 1)
            except Exception as ex:   # noqa
                row[f"c{cfg}s{sk}"] = "x"
    print(json.dumps(row), flush=True)

# warm the clocks (a fresh box idles low): ~1 s of launches before anything is timed
_w = torch.randn(4096, 4096, device="cuda").to(dt)
for _ in range(300): _w @ _w
torch.cuda.synchronize()
for name in a.shapes.split(","):
    if name == "LIN":
        linear_case(1024, 1280, 11520); continue
    if name.startswith("LIN:"):
        linear_case(*[int(v) for v in name.split(":")[1:]]); continue
    B, H, W, Cin, Cout = SH[name]
    xs = [torch.randn(B, H, W, Cin, device="cuda").to(dt) for _ in range(4)]
    ws = [(torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt) for _ in range(4)]
    bias = torch.randn(Cout, device="cuda")
    flops = 2.0 * B * H * W * Cout * 9 * Cin
    row = dict(shape=name, M=B * H * W, N=Cout, K=9 * Cin)
    best = (1e9, None)
    for cfg in [int(c) for c in a.cfgs.split(",")] + [-1]:
        for sk in ([0] if cfg == -1 else [int(s) for s in a.splits.split(",")]):
            try:
                i = [0]
                def f():
                    j = i[0] % 4; i[0] += 1
                    return ops.conv2d_nhwc(xs[j], ws[j], bias, taps=9, stride=1, cfg=cfg, split_k=sk)
                for _ in range(4): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters): f()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.iters
                row[f"c{cfg}s{sk}"] = round(us, 1)
                if cfg != -1 and us < best[0]: best = (us, (cfg, sk))
            except Exception as ex:   # noqa
                row[f"c{cfg}s{sk}"] = "x"
    row["best"] = best; row["best_tf"] = round(flops / best[0] / 1e6, 1); row["w_TBs"] = round(Cout * 9 * Cin * 2 / best[0] / 1e6, 2)
    print(json.dumps(row), flush=True)
