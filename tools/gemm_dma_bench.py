"""Tile config 16 (256^2 LDS-DMA tiles, gemm_dma.hip) against the table's choice on the large linears of the 32x32 / 16x16 levels."""
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
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)

dt = torch.bfloat16
for M, N, K, act, res in ((8192, 5120, 640, ops.ACT_GEGLU, False), (8192, 640, 2560, ops.ACT_NONE, True), (8192, 1920, 640, ops.ACT_NONE, False),
                          (2048, 10240, 1280, ops.ACT_GEGLU, False), (2048, 1280, 5120, ops.ACT_NONE, True), (2048, 3840, 1280, ops.ACT_NONE, False),
                          (32768, 2560, 320, ops.ACT_GEGLU, False), (32768, 320, 1280, ops.ACT_NONE, True), (512, 10240, 1280, ops.ACT_GEGLU, False)):
    xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(3)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(dt) if res else None
    out = torch.empty(M, N // 2 if act == ops.ACT_GEGLU else N, dtype=dt, device="cuda")
    i = [0]
    row = dict(shape=f"{M},{N},{K}", act=int(act), res=res)
    for name, cfg in (("table", -1), ("dma256", 16), ("table_b", -1), ("dma256_b", 16)):
        def go():
            j = i[0] % 3; i[0] += 1
            ops.linear(xs[j], w, b, res=r, act=act, out=out, cfg=cfg, split_k=0 if cfg < 0 else 1)
        row[name] = timed(go)
    row["dma256_tflops"] = round(2.0 * M * N * K / row["dma256_b"] / 1e6, 1)
    print(json.dumps(row), flush=True)
