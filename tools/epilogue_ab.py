"""Where does the time of the level-0 GEGLU projection (M = 32768, N = 2560, K = 320) go?  Same GEMM with different epilogues."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def timed(fn, iters=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for M, N, K in ((32768, 2560, 320), (8192, 5120, 640), (2048, 10240, 1280)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    row = dict(shape=f"{M},{N},{K}")
    for name, act in (("none", ops.ACT_NONE), ("silu", ops.ACT_SILU), ("gelu", ops.ACT_GELU), ("geglu", ops.ACT_GEGLU)):
        for cfg in (4, 0):
            row[f"{name}_c{cfg}"] = round(timed(lambda: ops.linear(x, w, b, act=act, cfg=cfg, split_k=1)), 1)
    row["none_nobias_c4"] = round(timed(lambda: ops.linear(x, w, None, cfg=4, split_k=1)), 1)
    print(json.dumps(row), flush=True)
