#!/usr/bin/env python
"""Step 0 of the Winograd question at the level of the whole UNet (CPU, a few minutes): the full-width fp32 oracle UNet (TEST INFRASTRUCTURE,
oracle/sd15.py) with ONLY its stride-1 3x3 convolutions on maps >= 32 wide emulated in 16 bit -- direct form (what the shipped kernels compute)
vs Winograd F(2x2, 3x3) (tools/winograd_numerics.py) -- against the all-fp32 forward at t = 981 / 481 / 1.  Prints one JSON line per timestep.
Results: profiles/r6_winograd_closure.md."""
import sys, json, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from oracle import sd15, processors as OP
from tools.winograd_numerics import winograd_conv
torch.set_num_threads(8)
MODE = {"m": "fp32", "dt": torch.float16}
class EmuConv(nn.Conv2d):
    def forward(self, x):
        m, dt = MODE["m"], MODE["dt"]
        big = self.kernel_size == (3, 3) and self.stride == (1, 1) and x.shape[-1] >= 32 and x.shape[-1] % 2 == 0 and self.in_channels >= 320
        if m == "fp32" or not big:
            return super().forward(x)
        x16 = x.to(dt).float()
        if m == "direct":
            return F.conv2d(x16, self.weight.to(dt).float(), self.bias, padding=1)
        return winograd_conv(x16, self.weight, dt) + self.bias.view(1, -1, 1, 1)
with torch.no_grad(), OP.reference_sdpa_dispatch():
    torch.manual_seed(0)
    u = sd15.UNet2DConditionModel()
    n = 0
    for mod in u.modules():
        for name, ch in list(mod.named_children()):
            if type(ch) is nn.Conv2d and ch.kernel_size == (3, 3) and ch.stride == (1, 1):
                ch.__class__ = EmuConv; n += 1
    print("emulated conv modules", n, flush=True)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 4, 64, 64, generator=g); pe = torch.randn(1, 77, 768, generator=g) * 0.5
    for t in (981, 481, 1):
        MODE["m"] = "fp32"; ref = u(z, t, pe)
        row = {"t": t, "eps_sigma": round(ref.std().item(), 3)}
        for dt in (torch.float16, torch.bfloat16):
            MODE["dt"] = dt
            for m in ("direct", "winograd"):
                MODE["m"] = m
                o = u(z, t, pe)
                row[f"{str(dt)[6:]}_{m}"] = {"max": round((o - ref).abs().max().item(), 5), "rms": round((o - ref).pow(2).mean().sqrt().item(), 6)}
        print(json.dumps(row), flush=True)
