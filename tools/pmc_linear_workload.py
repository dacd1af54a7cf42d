"""One plain linear launch (a few repeats, rotating weights) for the PMC passes of tools/gpu.sh pmc:
    python tools/pmc_linear_workload.py --M 8192 --N 640 --K 2560 --cfg 17 [--split 1] [--dtype bf16] [--res] [--geglu]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, default=8192); ap.add_argument("--N", type=int, default=640); ap.add_argument("--K", type=int, default=2560)
ap.add_argument("--cfg", type=int, default=17); ap.add_argument("--split", type=int, default=1); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=6); ap.add_argument("--res", action="store_true"); ap.add_argument("--geglu", action="store_true")
ap.add_argument("--time", action="store_true", help="print HIP-event time per launch (rotating operands, 50 launches) instead of the short PMC run")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
xs = [torch.randn(a.M, a.K, device="cuda").to(dt) for _ in range(4)]
ws = [(torch.randn(a.N, a.K, device="cuda") * a.K ** -0.5).to(dt) for _ in range(4)]
b = torch.randn(a.N, device="cuda")
res = torch.randn(a.M, a.N, device="cuda").to(dt) if a.res else None
act = ops.ACT_GEGLU if a.geglu else ops.ACT_NONE
def go(i):
    return ops.linear(xs[i % 4], ws[i % 4], b, res=res, act=act, cfg=a.cfg, split_k=a.split)
n = 50 if a.time else a.iters
for i in range(3): go(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n): go(i)
e1.record(); torch.cuda.synchronize()
if a.time:
    us = e0.elapsed_time(e1) * 1e3 / n
    import json
    print(json.dumps(dict(M=a.M, N=a.N, K=a.K, cfg=a.cfg, split=a.split, dtype=a.dtype, us=round(us, 2), tflops=round(2.0 * a.M * a.N * a.K / us / 1e6, 1))))
