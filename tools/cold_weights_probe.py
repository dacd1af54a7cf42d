"""What do COLD weights cost a launch?  In the denoising loop every layer's weights were last touched one UNet forward (12 ms, 1.7 GB of other
weights) ago: they come from HBM, not from L2 / the 256 MB Infinity Cache, while the tuner and the kernel A/Bs re-run a launch on a handful of
hot operand sets.  Times each shape with ONE weight set (hot) and with enough rotating sets to exceed 320 MB (cold); activations rotate over
four sets in both legs.    python tools/cold_weights_probe.py"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
dt = torch.bfloat16
SHAPES = [  # (name, kind, B, H, W, Cin, Cout)   kind: conv3 | lin
    ("L0 conv 320->320", "conv3", 8, 64, 64, 320, 320), ("L1 conv 640->640", "conv3", 8, 32, 32, 640, 640),
    ("L2 conv 1280->1280", "conv3", 8, 16, 16, 1280, 1280), ("L3 conv 1280->1280 (8x8)", "conv3", 8, 8, 8, 1280, 1280),
    ("L1 geglu 640->5120", "lin", 8, 32, 32, 640, 5120), ("L2 geglu 1280->10240", "lin", 8, 16, 16, 1280, 10240),
    ("L2 ff-out 5120->1280", "lin", 8, 16, 16, 5120, 1280), ("L0 proj 320->320", "lin", 8, 64, 64, 320, 320),
]
def timed(fn, n):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, kind, B, H, W, Cin, Cout in SHAPES:
    K = 9 * Cin if kind == "conv3" else Cin
    wbytes = Cout * K * 2
    nsets = min(256, max(2, math.ceil(320e6 / wbytes)))
    xs = [torch.randn(B, H, W, Cin, device="cuda").to(dt) for _ in range(4)]
    ws = [(torch.randn(Cout, K, device="cuda") * K ** -0.5).to(dt) for _ in range(nsets)]
    b = torch.randn(Cout, device="cuda")
    def go(i, cold):
        w = ws[i % nsets] if cold else ws[0]
        if kind == "conv3": ops.conv2d_nhwc(xs[i % 4], w, b, taps=9)
        else: ops.linear(xs[i % 4].view(-1, Cin), w, b)
    n = max(40, 2 * nsets)
    hot = timed(lambda i: go(i, False), n); cold = timed(lambda i: go(i, True), n)
    print(json.dumps(dict(shape=name, weight_mb=round(wbytes / 1e6, 1), sets=nsets, hot_us=round(hot, 1), cold_us=round(cold, 1), delta_us=round(cold - hot, 1))), flush=True)
    del ws, xs
    torch.cuda.empty_cache()
