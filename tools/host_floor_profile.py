"""Where do the ~75 us per processor call go on the HOST?  (Round-5 review item 4b.)  cProfile of 2000 calls of the hybrid processor through the
plugin surface at the smallest UNet shape (C = 1280, N = M = 64: ~10 us of GPU work), plus wall time per call with and without a device sync."""
import cProfile, io, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd.adapter import attention_processor as AP
from imagdressing_amd.unet import Attention

dev, dt = torch.device("cuda"), torch.bfloat16
C, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 64)
g = torch.Generator().manual_seed(1)
w = {k: torch.randn(C, C, generator=g) * C ** -0.5 for k in ("wq", "wk", "wv", "wo")}
sd = {"a.to_q.weight": w["wq"], "a.to_k.weight": w["wk"], "a.to_v.weight": w["wv"], "a.to_out.0.weight": w["wo"], "a.to_out.0.bias": torch.zeros(C)}
attn = Attention(sd, "a", 8, str(dev), dt)
proc = AP.RefSAttnProcessor2_0("blk.attn1.processor", C)
attn.set_processor(proc)
x = torch.randn(1, N, C, generator=g).to(device=dev, dtype=dt)
sa = {"blk.attn1.processor": torch.randn(1, N, C, generator=g).to(dev)}
with torch.no_grad():
    for _ in range(20):
        attn(x, sa_hidden_states=sa, residual=x)
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        attn(x, sa_hidden_states=sa, residual=x)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        attn(x, sa_hidden_states=sa, residual=x)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(json.dumps({"C": C, "N": N, "host_issue_us_per_call": round(t_issue / n * 1e6, 2), "wall_us_per_call_with_final_sync": round(t_all / n * 1e6, 2)}))
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
