"""The VAE decode of the bench batch alone (4 latents 64x64 -> 4 images 512x512, AutoencoderKL.decode, IMAGDressing_v1_pipeline.py decode_latents):
wall time per decode and, under `rocprofv3 --kernel-trace --stats`, its kernels.   python tools/vae_decode_profile.py [--batch 4] [--iters 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd.vae import AutoencoderKL

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
vae = AutoencoderKL.random_init(seed=5, device=torch.device("cuda", 0), dtype=dt)
z = torch.randn(a.batch, 4, 64, 64, device="cuda")
with torch.no_grad():
    for _ in range(2):
        out = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        out = vae.decode(z, return_dict=False)[0]
    e1.record(); torch.cuda.synchronize()
print(json.dumps(dict(batch=a.batch, dtype=a.dtype, ms_per_decode=round(e0.elapsed_time(e1) / a.iters, 3), out=list(out.shape), finite=bool(torch.isfinite(out.float()).all()))))
