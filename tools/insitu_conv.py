"""In-situ time of every conv / linear launch of the bench step (HIP events around each imd_conv_gemm launch inside the running sampling
loop), per shape and tile config.   python tools/insitu_conv.py [--force KEY=CFG[:SPLIT] ...] [--top 30]
--force overrides the tuning table for one shape key ("M,N,K,taps,stride,ups|HxW"): same-process A/B of tile configs IN SITU."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagdressing_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--force", nargs="*", default=[]); ap.add_argument("--top", type=int, default=30); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--taps", type=int, default=0, help="only keys with this many taps (9 | 1)")
ap.add_argument("--batch", type=int, default=4, help="images per batch (1: the latency shapes)")
ap.add_argument("--grep", default="", help="only keys containing this text")
ap.add_argument("--width", type=int, default=512); ap.add_argument("--height", type=int, default=512)
a = ap.parse_args()
tab = ops._gemm_table()
for f in a.force:
    key, v = f.split("=")
    cfg, _, sp = v.partition(":")
    ent = dict(tab.get(key) or tab.get(key.split("|")[0]) or {})
    ent.update(cfg=int(cfg), cfg_nosplit=int(cfg), split=int(sp) if sp else ent.get("split", 1))
    tab[key] = ent
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, torch.bfloat16, 0)
inp = bench.synthetic_inputs(a.width, a.height, a.batch, dev, torch.bfloat16, 0, 1)
def run():
    return pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=a.width, height=a.height, num_inference_steps=a.steps,
                guidance_scale=7.5, num_images_per_prompt=a.batch, output_type="latent", **inp).images
run(); torch.cuda.synchronize()
ops.GEMM_EVENT_HOOK = {}
run(); torch.cuda.synchronize()
hook, ops.GEMM_EVENT_HOOK = ops.GEMM_EVENT_HOOK, None
rows = []
for (key, cfg, split), evs in hook.items():
    ms = [e0.elapsed_time(e1) * 1e3 for e0, e1 in evs]
    rows.append(dict(key=key, cfg=cfg, split=split, launches=len(ms), avg_us=round(sum(ms) / len(ms), 1), min_us=round(min(ms), 1), total_ms=round(sum(ms) / 1e3, 2)))
tot = sum(r["total_ms"] for r in rows)
print(json.dumps(dict(total_ms_in_conv_gemm_launches=round(tot, 2), steps=a.steps)))
for r in sorted(rows, key=lambda r: -r["total_ms"])[:a.top]:
    if a.taps and f",{a.taps}," not in r["key"]: continue
    if a.grep and a.grep not in r["key"]: continue
    print(json.dumps(r))
