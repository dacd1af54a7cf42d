"""LayerNorm / GroupNorm streaming rate at the three UNet levels of the CFG batch (rotating buffers, HIP events)."""
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
    return e0.elapsed_time(e1) * 1e3 / iters

for rows, C in ((32768, 320), (8192, 640), (2048, 1280)):
    xs = [torch.randn(rows, C, device="cuda").to(torch.bfloat16) for _ in range(8)]
    outs = [torch.empty_like(xs[0]) for _ in range(8)]
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    i = [0]
    def ln():
        j = i[0] % 8; i[0] += 1
        ops.layer_norm(xs[j], g, b, out=outs[j])
    def gn():
        j = i[0] % 8; i[0] += 1
        ops.group_norm(xs[j].view(8, rows // 8, C), g, b, silu=True, out=outs[j].view(8, rows // 8, C))
    mb = 2 * rows * C * 2 / 1e6
    t_ln, t_gn = timed(ln), timed(gn)
    print(json.dumps(dict(rows=rows, C=C, ln_us=round(t_ln, 1), ln_TBs=round(mb / t_ln / 1e6 * 1e6 / 1e6, 2), gn_us=round(t_gn, 1),
                          gn_TBs=round(1.5 * mb / t_gn, 2))))
