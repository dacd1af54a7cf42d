import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
dt = torch.bfloat16
B, H, W, Cin, Cout = 8, 64, 64, 320, 320
x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt)
b = torch.randn(Cout, device="cuda")
for _ in range(6):
    ops.conv2d_nhwc(x, w, b, taps=9, cfg=5, split_k=1)
torch.cuda.synchronize()
