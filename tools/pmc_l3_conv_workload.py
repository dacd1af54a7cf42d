import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
dt = torch.bfloat16
B, H, W, Cin, Cout = 8, 8, 8, 1280, 1280
x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
ws = [(torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt) for _ in range(4)]
b = torch.randn(Cout, device="cuda")
for i in range(8):
    ops.conv2d_nhwc(x, ws[i % 4], b, taps=9)
torch.cuda.synchronize()
