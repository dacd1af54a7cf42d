"""Time the HIP VAE at the BASELINE image size: decode of the 4-image batch (latent [4, 4, 64, 64] -> [4, 3, 512, 512]) and
encode of one garment image; FLOPs are the sum of 2*M*N*K over the GEMM / conv launches of one call."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops
from imagdressing_amd.vae import AutoencoderKL


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    vae = AutoencoderKL.random_init(seed=5, device="cuda", dtype=dt)
    z = torch.randn(a.batch, 4, a.res // 8, a.res // 8, device="cuda")
    img = torch.rand(1, 3, a.res, a.res, device="cuda") * 2 - 1
    out = {}
    for name, fn in (("decode", lambda: vae.decode(z, return_dict=False)[0]), ("encode", lambda: vae.encode(img).latent_dist.mean)):
        ops.GEMM_TRACE = []
        fn()
        fl = sum(2.0 * t["M"] * t["N"] * t["K"] for t in ops.GEMM_TRACE)
        ops.GEMM_TRACE = None
        ms = timed(fn, a.iters)
        out[name] = dict(ms=round(ms, 2), gflop=round(fl / 1e9, 1), tflops=round(fl / ms / 1e9, 1))
    print(json.dumps(dict(batch=a.batch, res=a.res, dtype=a.dtype, **out)))


if __name__ == "__main__":
    main()
