"""Level-0 hybrid attention (CFG batch 8: 4 garment rows + 4 plain rows, N = M = 4096) with and without the out-projection in the
launch (ABI v7): microseconds per [attention], [attention + imd_conv_gemm out-projection], [fused launch]."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
dt = torch.bfloat16
D, N, H, Bimg = 40, 4096, 8, 4
B, Cc = 2 * Bimg, 320
dpk, dpv = ops.attn_padded_dims(D)
g = torch.Generator(device="cuda").manual_seed(0)
def r(*s): return torch.randn(*s, generator=g, device="cuda").to(dt)
q = torch.zeros(B, H, N, dpk, dtype=dt, device="cuda"); q[..., :D] = r(B, H, N, D) * (D ** -0.5 * math.log2(math.e))
k = ops.k_buffer((B, H, N, dpk), D, dt, "cuda"); k[..., :D] = r(B, H, N, D)
vt = torch.zeros(B, H, dpv, N, dtype=dt, device="cuda"); vt[:, :, :D, :N] = r(B, H, D, N)
kr = ops.k_buffer((1, H, N, dpk), D, dt, "cuda"); kr[..., :D] = r(1, H, N, D)
vr = torch.zeros(1, H, dpv, N, dtype=dt, device="cuda"); vr[:, :, :D, :N] = r(1, H, D, N)
s2 = torch.cat([torch.ones(Bimg), torch.zeros(Bimg)]).cuda()
o = torch.empty(B, N, Cc, dtype=dt, device="cuda"); out = torch.empty_like(o)
wo = r(Cc, Cc) * Cc ** -0.5; bo = torch.randn(Cc, device="cuda"); res = r(B, N, Cc)
kw = dict(B=B, H=H, N=N, D=D, L1=N, L1P=N, k2=kr, v2t=vr, scale2=s2, L2=N, L2P=N, kv2_bdiv=B, k_pad_one=True)
def plain(): ops.attention(q, k, vt, o, **kw)
def two():
    ops.attention(q, k, vt, o, **kw)
    ops.linear(o.view(B * N, Cc), wo, bo, res=res.view(B * N, Cc))
def fused(): ops.attention(q, k, vt, o, proj=(wo, bo, res, out), **kw)
def lin(): ops.linear(o.view(B * N, Cc), wo, bo, res=res.view(B * N, Cc))
_w = torch.randn(4096, 4096, device="cuda").to(dt)
for _ in range(200): _w @ _w
res_us = {}
for name, f in (("attention", plain), ("out_projection_launch", lin), ("attention_then_projection", two), ("fused", fused), ("attention", plain), ("fused", fused)):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    res_us.setdefault(name, []).append(round(e0.elapsed_time(e1) * 10, 1))
print(json.dumps(res_us))
