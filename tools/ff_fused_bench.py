"""Fused feed-forward (ff_fused.hip) against the engine's four launches (layernorm, GEGLU projection, out projection + residual) at the
64x64 level (M = 32768, C = 320, inner = 1280); operands rotate over 4 sets."""
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

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
M, Cc, inner = int(os.environ.get("FF_M", 32768)), 320, 1280
xs = [torch.randn(M, Cc, device="cuda").to(dt) for _ in range(4)]
w1 = (torch.randn(2 * inner, Cc, device="cuda") * Cc ** -0.5).to(dt); b1 = torch.randn(2 * inner, device="cuda") * 0.1
w2 = (torch.randn(Cc, inner, device="cuda") * inner ** -0.5).to(dt); b2 = torch.randn(Cc, device="cuda") * 0.1
g = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda")
wi = torch.stack([w1[:inner], w1[inner:]], dim=1).reshape(2 * inner, Cc).contiguous(); bi = torch.stack([b1[:inner], b1[inner:]], dim=1).reshape(-1).contiguous()
packed = ops.pack_ff_fused(w1, b1, w2, b2, g, be)
out = torch.empty(M, Cc, dtype=dt, device="cuda")
i = [0]
def unfused():
    j = i[0] % 4; i[0] += 1
    n = ops.layer_norm(xs[j], g, be, 1e-5)
    gg = ops.linear(n, wi, bi, act=ops.ACT_GEGLU)
    ops.linear(gg, w2, b2, res=xs[j], out=out)
def fused():
    j = i[0] % 4; i[0] += 1
    ops.ff_geglu_fused(xs[j], packed, 1e-5, out=out)
row = dict(M=M, dtype=str(dt))
for rep in range(2):
    row[f"unfused_us_{rep}"] = timed(unfused); row[f"fused_us_{rep}"] = timed(fused)
fl = 2.0 * M * Cc * 2 * inner + 2.0 * M * inner * Cc
row["fused_tflops"] = round(fl / row["fused_us_1"] / 1e6, 1)
print(json.dumps(row), flush=True)
