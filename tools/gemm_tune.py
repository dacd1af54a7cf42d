"""Per-shape tuning of the implicit-GEMM kernel for the shapes ONE denoising step actually launches.

    python tools/gemm_tune.py [--batch 4 --res 512 --dtype bf16] -> gpurun_out/gemm_tuning.json

Records every conv_gemm shape of one CFG-batched UNet forward (+ the garment pass), times each distinct shape
with every (tile config, K split) candidate on synthetic operands (interleaved repeats, min of medians), and
writes the winners.  Copy the result to imagdressing_amd/gemm_tuning.json to use it."""
import argparse, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagdressing_amd import ops

CFGS = [0, 4, 2, 1, 5, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]      # 12-15: row-resident kernels (K = 320 / 640 / 1280 linears, 320 -> 960 qkv), 16: 256^2 LDS-DMA tile kernel; refused elsewhere
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]


def collect_shapes(args, dt):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import configs
    dev = torch.device("cuda", 0)
    pipe, kw = configs.build(args.config, dev, dt, args.batch or None, steps=1, width=args.width, height=args.height)
    ops.GEMM_TRACE = []
    pipe(**kw)
    torch.cuda.synchronize()
    tr, ops.GEMM_TRACE = ops.GEMM_TRACE, None
    del pipe
    torch.cuda.empty_cache()
    uniq = {}
    for t in tr:
        key = f"{t['M']},{t['N']},{t['K']},{t['taps']},{t['stride']},{t['ups']}"
        if t["taps"] == 9:           # (same M, N, K for different maps: see ops.conv_gemm)
            key += f"|{t['Hout']}x{t['Wout']}"
        e = uniq.setdefault(key, dict(t, count=0, any_splittable=False, any_unsplittable=False))
        e["count"] += 1
        e["any_splittable"] |= t["splittable"]
        e["any_unsplittable"] |= not t["splittable"]
    return uniq


def time_candidate(t, cfg, split, dt, iters):
    B = t["M"] // (t["Hout"] * t["Wout"])
    x = torch.randn(B, t["Hin"], t["Win"], t["Cin"], device="cuda").to(dt)
    w = (torch.randn(t["N"], t["K"], device="cuda") * t["K"] ** -0.5).to(dt)
    bias = torch.randn(t["N"], device="cuda")
    out = torch.empty(t["M"], t["N"], dtype=dt, device="cuda")
    def go():
        ops.conv_gemm(x, w, M=t["M"], N=t["N"], Cin=t["Cin"], taps=t["taps"], Hin=t["Hin"], Win=t["Win"], Hout=t["Hout"],
                      Wout=t["Wout"], stride=t["stride"], ups=bool(t["ups"]), bias=bias, out=out, cfg=cfg, split_k=split)
    for _ in range(2):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def only_linear(args, shapes, dt):
    path = os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "gemm_tuning.json")
    with open(path) as f:
        doc = json.load(f)
    table = doc.get("shapes", {})
    log = []
    for key, t in sorted(shapes.items(), key=lambda kv: -kv[1]["count"] * kv[1]["M"] * kv[1]["N"] * kv[1]["K"]):
        tkey = key if key in table else key.split("|")[0]       # (3x3 shapes are traced with their map, the table may hold the plain key)
        if tkey not in table or (t["taps"] != 1 and not args.keys) or (args.keys and tkey not in args.keys.split(";") and key not in args.keys.split(";")):
            continue
        ent = table[tkey]
        shipped = (ent["cfg"], ent["split"]) if t["any_splittable"] else (ent["cfg_nosplit"], 1)
        cands = {}
        for rep in range(2):
            if args.cands:          # a focused re-time: the shipped choice against the named tile configs (un-split, and at the shipped K split)
                cc = [int(c) for c in args.cands.split(",")]
                others = [(c, 1) for c in cc] + ([(c, shipped[1]) for c in cc] if shipped[1] > 1 else [])
                if args.splits and t["any_splittable"]:
                    others += [(c, sp) for c in cc for sp in (int(x) for x in args.splits.split(",")) if sp > 1 and (t["K"] + 63) // 64 // sp >= 4]
                others = list(dict.fromkeys(others))
                if t["K"] % 64 or t["K"] < args.min_k:
                    continue
            else:
                others = [(c, 1) for c in (0, 4, 9, 10, 11, 12, 13, 14, 15, 16, 17, 19)]
            for cand in [shipped] + [c for c in others if c != shipped]:
                try:
                    us = time_candidate(t, cand[0], cand[1], dt, args.iters)
                except Exception:          # noqa
                    continue
                cands.setdefault(cand, []).append(us)
        if shipped not in cands:
            continue
        best = min(cands, key=lambda k: min(cands[k]))
        rec = dict(key=key, count=t["count"], shipped=list(shipped), shipped_us=round(min(cands[shipped]), 1), best=list(best), best_us=round(min(cands[best]), 1))
        if best != shipped and min(cands[best]) < 0.97 * min(cands[shipped]):
            if t["any_splittable"]:
                ent["cfg"], ent["split"] = best
            if best[1] == 1:
                ent["cfg_nosplit"] = best[0]
            rec["changed"] = True
        log.append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    doc["shapes"] = table
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    with open(args.out.replace(".json", "_log.jsonl"), "w") as f:
        for r in log:
            f.write(json.dumps(r) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, help="BASELINE configuration whose shapes are tuned: 1 (bench line), 3 (IPA + "
                    "ControlNet, batch 8), 5 (768x576 inpainting + ControlNet, 4 images); see tools/configs.py")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--width", type=int, default=512, help="config 1 only: generated width (the reference scripts' default geometry is 512 x 640)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--merge", action="store_true", help="start from the shipped table and add / overwrite only the traced shapes")
    ap.add_argument("--skip-known", action="store_true", help="with --merge: leave shapes that the shipped table already holds alone")
    ap.add_argument("--quick", action="store_true", help="one repeat, K splits {1, 2, 4}")
    ap.add_argument("--keep-margin", type=float, default=0.0, help="with --merge: keep a shape's shipped choice unless the winner is faster by this fraction")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--out", default="gpurun_out/gemm_tuning.json")
    ap.add_argument("--only-conv3x3", action="store_true",
                    help="re-time only the 3x3 stride-1 convolutions (candidates 0, 4, 5) and merge into the shipped table")
    ap.add_argument("--cands", default="", help="with --only-linear: comma list of tile configs to time against the shipped choice (e.g. 25,27)")
    ap.add_argument("--min-k", type=int, default=0, help="with --cands: only shapes with K >= this")
    ap.add_argument("--splits", default="", help="with --cands: also time every candidate at these K splits (comma list, e.g. 2,4) where the shape is splittable")
    ap.add_argument("--keys", default="", help="with --only-linear --cands: re-time exactly these table keys (';'-separated; 3x3 keys allowed)")
    ap.add_argument("--only-linear", action="store_true",
                    help="re-time only the plain linear layers (taps = 1) with the shipped choice against the unsplit candidates 0, 4, 9, 10, 11 and "
                         "the round-2 kernels 12..16; an entry changes only when the winner is > 3 %% faster than the shipped choice")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    shapes = collect_shapes(args, dt)
    if args.only_linear:
        return only_linear(args, shapes, dt)
    result, log = {}, []
    cfgs = CFGS
    splits = [1, 2, 4] if args.quick else SPLITS
    if args.merge:
        with open(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "gemm_tuning.json")) as f:
            result = json.load(f).get("shapes", {})
        if args.skip_known:
            shapes = {k: t for k, t in shapes.items() if k not in result}
    if args.only_conv3x3:
        with open(os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "gemm_tuning.json")) as f:
            result = json.load(f).get("shapes", {})
        shapes = {k: t for k, t in shapes.items() if t["taps"] == 9 and t["stride"] == 1 and t["Cin"] % 32 == 0}      # (round 3: fused-upsample convs too)
        cfgs = [5, 21, 22, 23, 18]           # round 4: the three halo-patch tilings + the gathering LDS-DMA tiles; the shipped choice is re-timed beside them
        if args.keep_margin == 0.0:
            args.keep_margin = 0.03
    total_before = total_after = 0.0
    for key, t in sorted(shapes.items(), key=lambda kv: -kv[1]["count"] * kv[1]["M"] * kv[1]["N"] * kv[1]["K"]):
        flops = 2.0 * t["M"] * t["N"] * t["K"]
        cands = {}
        ktiles = (t["K"] + 63) // 64
        shipped_cfg = result.get(key, {}).get("cfg") if args.only_conv3x3 else None
        for rep in range(1 if args.quick else 2):
            for cfg in (cfgs + [shipped_cfg] if shipped_cfg is not None and shipped_cfg not in cfgs else cfgs):
                for split in splits:
                    if split > 1 and (ktiles // split < 8 or not t["any_splittable"]):
                        continue
                    try:
                        us = time_candidate(t, cfg, split, dt, args.iters)
                    except Exception as ex:          # noqa
                        continue
                    cands.setdefault((cfg, split), []).append(us)
        best = min(cands, key=lambda k: min(cands[k]))
        best_ns = min((k for k in cands if k[1] == 1), key=lambda k: min(cands[k]))
        old = result.get(key) if args.keep_margin > 0 else None
        if old is not None:          # hysteresis against timing noise: a shipped choice stays unless the winner beats it by the margin
            sh = (old["cfg"], old["split"]) if t["any_splittable"] else (old["cfg_nosplit"], 1)
            if sh in cands and min(cands[best]) > (1.0 - args.keep_margin) * min(cands[sh]):
                best = sh
            shn = (old["cfg_nosplit"], 1)
            if shn in cands and min(cands[best_ns]) > (1.0 - args.keep_margin) * min(cands[shn]):
                best_ns = shn
        auto_cfg = ops.L.load().imd_conv_gemm_auto_cfg(t["M"], t["N"])
        auto_split = ops.L.load().imd_conv_gemm_auto_split(t["M"], t["N"], t["K"], auto_cfg) if t["any_splittable"] else 1
        auto_us = min(cands.get((auto_cfg, auto_split), cands.get((auto_cfg, 1), [float("nan")])))
        result[key] = dict(cfg=best[0], split=best[1], cfg_nosplit=best_ns[0])
        total_before += auto_us * t["count"]; total_after += min(cands[best]) * t["count"]
        log.append(dict(key=key, count=t["count"], best=list(best), best_us=round(min(cands[best]), 1),
                        best_tf=round(flops / min(cands[best]) / 1e6, 1), heuristic=[auto_cfg, auto_split], heuristic_us=round(auto_us, 1)))
        print(json.dumps(log[-1]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(dict(note="measured by tools/gemm_tune.py on MI355X", config=args.config, batch=args.batch, dtype=args.dtype,
                       shapes=result, log=log, total_us_heuristic=round(total_before, 1), total_us_tuned=round(total_after, 1)), f, indent=1)
    print("total per traced pass: heuristic %.1f us -> tuned %.1f us" % (total_before, total_after))


if __name__ == "__main__":
    main()
