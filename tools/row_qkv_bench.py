"""norm1 -> q/k/v projection of the 64x64 level: layernorm + tiled head-split GEMM (two launches) against row_qkv.hip (one)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

def timed(fn, iters=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)

dt = torch.bfloat16
B, HW, Cc, H, D = 8, 4096, 320, 8, 40
DPK, DPV = ops.attn_padded_dims(D); LP = ops.pad64(HW)
xs = [torch.randn(B * HW, Cc, device="cuda").to(dt) for _ in range(4)]
w = (torch.randn(3 * Cc, Cc, device="cuda") * Cc ** -0.5).to(dt)
g = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda")
w2, b2 = ops.fold_layernorm_affine(w, None, g, be)
q = torch.zeros(B, H, HW, DPK, dtype=dt, device="cuda"); k = torch.zeros_like(q); vt = torch.zeros(B, H, DPV, LP, dtype=dt, device="cuda")
heads = dict(C=Cc, H=H, D=D, dests=[(q, 0, DPK, HW, 0.2), (k, 0, DPK, HW, 1.0), (vt, 1, DPV, LP, 1.0)])
nrm = torch.empty(B * HW, Cc, dtype=dt, device="cuda")
i = [0]
def two():
    j = i[0] % 4; i[0] += 1
    ops.layer_norm(xs[j], g, be, out=nrm)
    ops.conv_gemm(nrm, w, M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, heads=heads)
def tiled_only():
    j = i[0] % 4; i[0] += 1
    ops.conv_gemm(xs[j], w, M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, heads=heads)
def row_only():
    j = i[0] % 4; i[0] += 1
    ops.conv_gemm(xs[j], w, M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, heads=heads, cfg=15)
def fused():
    j = i[0] % 4; i[0] += 1
    ops.conv_gemm(xs[j], w2, M=B * HW, N=3 * Cc, Cin=Cc, Hin=HW, Win=1, Hout=HW, Wout=1, bias=b2, heads=heads, ln_eps=1e-5)
row = {}
for rep in range(2):
    for name, fn in (("ln_then_tiled_qkv", two), ("tiled_qkv", tiled_only), ("row_qkv", row_only), ("ln_row_qkv_fused", fused)):
        row[f"{name}_{rep}"] = timed(fn)
print(json.dumps(row), flush=True)
