"""Row-resident kernel (tile config 12 / ln_eps) against the tiled kernel on the K = N = 320 projections of the 64x64 level
(M = 32768).  Operands rotate over 6 sets (cold, as in the loop).  Event timings are host-bound below ~9 us per launch:
run under `rocprofv3 --kernel-trace --stats` for the kernel durations."""
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

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
M, N, K = {'320': (32768, 320, 320), '640': (8192, 640, 640), '1280': (2048, 1280, 1280)}[os.environ.get('RL_K', '320')]
TILED, ROW = {320: (4, 12), 640: (2, 13), 1280: (2, 14)}[K]
xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(6)]
ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(dt) for _ in range(6)]
rs = [torch.randn(M, N, device="cuda").to(dt) for _ in range(6)]
b = torch.randn(N, device="cuda"); g = torch.ones(K, device="cuda"); be = torch.zeros(K, device="cuda")
out = torch.empty(M, N, dtype=dt, device="cuda"); nrm = torch.empty(M, K, dtype=dt, device="cuda")
i = [0]
row = {}
for name, cfg in (("tiled", TILED), ("row", ROW), ("tiled_b", TILED), ("row_b", ROW)):
    def go():
        j = i[0] % 6; i[0] += 1
        ops.linear(xs[j], ws[j], b, res=rs[j], out=out, cfg=cfg, split_k=1)
    row[name + "_res"] = timed(go)
    def go2():
        j = i[0] % 6; i[0] += 1
        ops.linear(xs[j], ws[j], b, out=out, cfg=cfg, split_k=1)
    row[name + "_nores"] = timed(go2)
def two():
    j = i[0] % 6; i[0] += 1
    ops.layer_norm(xs[j], g, be, out=nrm)
    ops.linear(nrm, ws[j], b, out=out, cfg=TILED, split_k=1)
def fused():
    j = i[0] % 6; i[0] += 1
    ops.linear(xs[j], ws[j], b, out=out, ln_eps=1e-5)
row["ln_then_linear"] = timed(two); row["ln_linear_fused"] = timed(fused)
row["ln_then_linear_b"] = timed(two); row["ln_linear_fused_b"] = timed(fused)
print(json.dumps(row), flush=True)
