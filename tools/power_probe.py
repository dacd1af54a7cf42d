"""Sustained run of one kernel configuration while sampling the board's power and shader clock with rocm-smi (is the kernel
power- / clock-limited?).  python tools/power_probe.py --variant 10 [--zero] [--seconds 4]"""
import argparse, json, math, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=0, help="knob 0 value (0 = library default)")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--zero", action="store_true")
ap.add_argument("--seconds", type=float, default=4.0)
ap.add_argument("--what", default="attn", choices=["attn", "idle", "conv"])
ap.add_argument("--cfg", type=int, default=5, help="--what conv: tile config of the level-0 320 -> 320 3x3 conv (5 | 21 | 0 ...)")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
D, N, H, Bimg = 40, 4096, 8, 4
B = 2 * Bimg
dpk, dpv = ops.attn_padded_dims(D)
g = torch.Generator(device="cuda").manual_seed(0)
def r(*s): return torch.randn(*s, generator=g, device="cuda").to(dt) * (0.0 if a.zero else 1.0)
q = torch.zeros(B, H, N, dpk, dtype=dt, device="cuda"); q[..., :D] = r(B, H, N, D) * (D ** -0.5 * math.log2(math.e))
k = ops.k_buffer((B, H, N, dpk), D, dt, "cuda"); k[..., :D] = r(B, H, N, D)
vt = torch.zeros(B, H, dpv, N, dtype=dt, device="cuda"); vt[:, :, :D, :N] = r(B, H, D, N)
kr = ops.k_buffer((1, H, N, dpk), D, dt, "cuda"); kr[..., :D] = r(1, H, N, D)
vr = torch.zeros(1, H, dpv, N, dtype=dt, device="cuda"); vr[:, :, :D, :N] = r(1, H, D, N)
s2 = torch.cat([torch.ones(Bimg), torch.zeros(Bimg)]).cuda()
out = torch.empty(B, N, H * D, dtype=dt, device="cuda")
if a.variant:
    ops.L.check(ops.L.load().imd_set_tuning(0, a.variant))
def go():
    ops.attention(q, k, vt, out, B=B, H=H, N=N, D=D, L1=N, L1P=N, k2=kr, v2t=vr, scale2=s2, L2=N, L2P=N, kv2_bdiv=B, k_pad_one=True)
if a.what == "conv":           # the level-0 320 -> 320 3x3 conv of the bench batch (CFG batch 8, 64 x 64 maps), rotating operands
    xs = [torch.randn(8, 64, 64, 320, device="cuda").to(dt) * (0.0 if a.zero else 1.0) for _ in range(4)]
    ws = [(torch.randn(320, 2880, device="cuda") * 2880 ** -0.5).to(dt) * (0.0 if a.zero else 1.0) for _ in range(4)]
    cb = torch.zeros(320, device="cuda")
    ci = [0]
    def go():
        j = ci[0] % 4; ci[0] += 1
        ops.conv2d_nhwc(xs[j], ws[j], cb, taps=9, stride=1, cfg=a.cfg, split_k=1)
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(o)
            c = j.get("card0", {})
            samples.append({k2: v for k2, v in c.items() if "ower" in k2 or "sclk" in k2.lower()})
        except Exception as e:
            samples.append({"err": str(e)[:80]})
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
for _ in range(5): go()
torch.cuda.synchronize()
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < a.seconds:
    if a.what in ("attn", "conv"):
        for _ in range(50): go()
        n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
us = e0.elapsed_time(e1) * 1e3 / max(n, 1)
def num(x):
    try:
        return float(str(x).strip("()MhzW "))
    except ValueError:
        return None
tail = samples[len(samples) // 3:]
pw = [num(r.get("Current Socket Graphics Package Power (W)")) for r in tail]
ck = [num(r.get("sclk clock speed:")) for r in tail]
pw, ck = [v for v in pw if v], [v for v in ck if v]
print(json.dumps(dict(variant=a.variant, dtype=a.dtype, zero=a.zero, what=a.what, cfg=a.cfg if a.what == "conv" else None, launches=n, us_per_launch=round(us, 1),
                      power_w=round(sum(pw) / len(pw), 1) if pw else None, sclk_mhz=round(sum(ck) / len(ck), 1) if ck else None,
                      n_samples=len(tail), samples=tail[:6])))
