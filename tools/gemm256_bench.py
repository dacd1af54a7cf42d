"""Microbenchmark of the feed-forward linears with their REAL epilogues (GEGLU / bias + residual) across tile configs, and the timing ablations of
gemm_dma256.hip (tuning knob 2 bits 5..7: 32 no epilogue, 64 no DMA wait / barrier, 128 no fragment reads / MFMAs -- WRONG results).
    python tools/gemm256_bench.py [--cfgs 17,25,30,31,32] [--ablate]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--cfgs", default="17,25,30,31")
ap.add_argument("--ablate", action="store_true")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--only", type=int, default=-1, help="index of the one shape to run (PMC passes)")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
lib = ops.L.load()
base_flags = lib.imd_get_tuning(2)


def timed(fn, iters):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)


SHAPES = ((8192, 5120, 640, ops.ACT_GEGLU, False, 1), (2048, 10240, 1280, ops.ACT_GEGLU, False, 1), (8192, 640, 2560, ops.ACT_NONE, True, 1),
          (8192, 640, 2560, ops.ACT_NONE, True, 2), (2048, 1280, 5120, ops.ACT_NONE, True, 3), (8192, 1920, 640, ops.ACT_NONE, False, 1),
          (2048, 3840, 1280, ops.ACT_NONE, False, 1), (8192, 5120, 640, ops.ACT_NONE, False, 1))
for si, (M, N, K, act, res, split) in enumerate(SHAPES):
    if a.only >= 0 and si != a.only: continue
    xs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(3)]
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(dt) if res else None
    out = torch.empty(M, N // 2 if act == ops.ACT_GEGLU else N, dtype=dt, device="cuda")
    i = [0]
    row = dict(shape=f"{M},{N},{K}", act="geglu" if act == ops.ACT_GEGLU else ("bias+res" if res else "bias"), split=split, gflop=round(2.0 * M * N * K / 1e9, 1))
    for rep in ("a", "b"):
        for cfg in [int(c) for c in a.cfgs.split(",")]:
            def go():
                j = i[0] % 3; i[0] += 1
                ops.linear(xs[j], w, b, res=r, act=act, out=out, cfg=cfg, split_k=split)
            try:
                us = timed(go, a.iters)
            except Exception as ex:      # noqa: BLE001
                us = None
            k = f"cfg{cfg}"
            row[k] = us if k not in row or row[k] is None else min(row[k], us)
    if a.ablate and split == 1:
        for cfg in (30,):
            for bits, name in ((32, "no_epilogue"), (64, "no_wait_no_barrier"), (128, "no_mfma"), (32 | 128, "staging_and_sync_only"), (32 | 64, "issue_reads_mfma_only")):
                ops.L.check(lib.imd_set_tuning(2, base_flags | bits))
                try:
                    def go():
                        j = i[0] % 3; i[0] += 1
                        ops.linear(xs[j], w, b, res=r, act=act, out=out, cfg=cfg, split_k=1)
                    row[f"cfg{cfg}_{name}"] = timed(go, a.iters)
                finally:
                    ops.L.check(lib.imd_set_tuning(2, base_flags))
    print(json.dumps(row), flush=True)
