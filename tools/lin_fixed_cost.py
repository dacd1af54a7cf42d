"""Fixed-cost decomposition of the K = N projections: time against K (one K tile ... full), with / without the residual,
against the pure-traffic floor of the same bytes (elementwise add of two [M, N] 16-bit tensors = 3 x M x N x 2 bytes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def timed(fn, iters=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)

dt = torch.bfloat16
for M, N in ((32768, 320), (8192, 640), (2048, 1280)):
    rs = [torch.randn(M, N, device="cuda").to(dt) for _ in range(8)]
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=dt, device="cuda")
    row = dict(M=M, N=N)
    i = [0]
    def add():
        j = i[0] % 8; i[0] += 1
        ops.add(rs[j], rs[(j + 3) % 8], out=out)
    row["add_floor"] = timed(add)
    for K in (32, 64, 128, N // 2, N, 2 * N):
        xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(8)]
        ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(8)]
        for cfg in (4, 0, 2, 6):
            for res in (True, False):
                def go():
                    j = i[0] % 8; i[0] += 1
                    ops.linear(xs[j], ws[j], b, res=rs[j] if res else None, out=out, cfg=cfg, split_k=1)
                row[f"K{K}_c{cfg}_{'res' if res else 'nores'}"] = timed(go)
    print(json.dumps(row), flush=True)
