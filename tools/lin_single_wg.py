"""Critical path of ONE workgroup of the implicit-GEMM kernel: launches whose grid is a single tile (M = BM, N = BN), against
K; the difference to the full-grid time of the same K is contention / traffic, this is the latency chain."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def timed(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 2)

dt = torch.bfloat16
z = torch.zeros(1024, device="cuda")
row = dict(empty_torch_kernel=timed(lambda: z.add_(1.0)))
a = torch.randn(64, 320, device="cuda").to(dt); o = torch.empty_like(a)
row["add_64x320"] = timed(lambda: ops.add(a, a, out=o))
g = torch.ones(320, device="cuda"); b0 = torch.zeros(320, device="cuda")
row["layernorm_64x320"] = timed(lambda: ops.layer_norm(a, g, b0, out=o))
print(json.dumps(row), flush=True)
for cfg, BM, BN in ((4, 128, 128), (0, 128, 128), (2, 64, 64), (6, 64, 320)):
    row = dict(cfg=cfg)
    for K in (32, 64, 320, 1280, 5120):
        x = torch.randn(BM, K, device="cuda").to(dt)
        w = (torch.randn(BN, K, device="cuda") * K ** -0.5).to(dt)
        b = torch.randn(BN, device="cuda")
        r = torch.randn(BM, BN, device="cuda").to(dt)
        out = torch.empty(BM, BN, dtype=dt, device="cuda")
        row[f"K{K}"] = timed(lambda: ops.linear(x, w, b, res=r, out=out, cfg=cfg, split_k=1))
    print(json.dumps(row), flush=True)
