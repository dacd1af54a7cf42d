"""A/B of tile configs / K splits on the three K = N square projections (75 launches per denoising step)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def timed(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

for M, N, K in ((32768, 320, 320), (8192, 640, 640), (2048, 1280, 1280)):
    # rotate over 8 distinct operand sets so that inputs are not cache-resident from the previous launch (as in the real loop)
    xs = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(8)]
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(8)]
    rs = [torch.randn(M, N, device="cuda").to(torch.bfloat16) for _ in range(8)]
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    row = dict(shape=f"{M},{N},{K}")
    for cfg in (0, 1, 2, 3, 4, 6, 7, 8):
        for sk in (1, 2):
            i = [0]
            def go():
                j = i[0] % 8; i[0] += 1
                ops.linear(xs[j], ws[j], b, res=rs[j], out=out, cfg=cfg, split_k=sk)
            try:
                row[f"c{cfg}s{sk}"] = round(timed(go), 1)
            except Exception as e:
                row[f"c{cfg}s{sk}"] = str(e)[:30]
    print(json.dumps(row), flush=True)
