#!/usr/bin/env python
"""bench.py -- 512x512 / 50-step images per second of the IMAGDressing-v1 denoising hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: the garment UNet once + 50 DDIM steps of the
CFG-batched denoising UNet (hybrid attention with the garment branch on the cond rows) for
``--batch`` images per GPU sharing one garment (BASELINE.json configs[1]: SD1.5 bf16, 512x512,
50 steps, batch 4 on one MI355X, garment cross-attn only).  Inputs (text / garment-token
embeddings, garment latent, initial latents) are synthetic, seeded, and resident in HBM before the
timed region; weights are random-init of the SD1.5 architecture (no checkpoints exist offline).
At N > 1 every rank runs its own shard of the images (weak scaling: per-GPU batch fixed) and the
garment features are computed on rank 0 and broadcast (RCCL) once per step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU sharing one garment")
    ap.add_argument("--res", type=int, default=512, help="width = height of the generated image (BASELINE configs[1]: 512)")
    ap.add_argument("--width", type=int, default=0, help="override --res (the reference scripts' default geometry is --width 512 --height 640)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-geometry-secondary", action="store_true",
                    help="skip the extra short timed run at the reference scripts' own default geometry, 512 wide x 640 high (inference_IMAGdressing.py:182-183)")
    ap.add_argument("--no-parity", action="store_true", help="skip the measured parity of both element types against the committed fp32-oracle UNet forward")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency figures (eager vs HIP-graph replay of the step)")
    ap.add_argument("--no-flops", action="store_true", help="skip the algorithmic FLOP count of one bench step (one extra untimed step)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the timed runs of the other single-GPU BASELINE configurations (configs[2]: IP-Adapter + ControlNet, batch 8; "
                         "configs[4]: 768x576 ControlNet inpainting, batch 4)")
    ap.add_argument("--no-power", action="store_true", help="skip the rocm-smi power / clock reading of the roofline kernel")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 --pmc (falls back to the committed figure)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=2)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU rehearsals of N > 1)")
    ap.add_argument("--device", type=int, default=-1, help="force the HIP device index of every rank (rehearsal of N > 1 on one GPU)")
    ap.add_argument("--decode", dest="decode", action="store_true", default=True,
                    help="run the HIP VAE decoder on the final latents inside the timed region (default: the metric is IMAGES/s, "
                         "IMAGDressing_v1_pipeline.py:544-546)")
    ap.add_argument("--no-decode", dest="decode", action="store_false", help="stop at the final latents (round-1 definition of a step)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second timed run in the other 16-bit element type (fp16 when --dtype bf16 and vice versa)")
    ap.add_argument("--attn-qw", type=int, default=0, help="tuning knob 0 of the library (0 = library default)")
    ap.add_argument("--gemm-flags", type=int, default=-1, help="tuning knob 2 of the library (-1 = library default)")
    ap.add_argument("--splitk-in-kernel", action="store_true", help="A/B: split-K slices summed by each tile's last-arriving workgroup (opt-in, slower)")
    ap.add_argument("--no-fused-ff", action="store_true", help="A/B: norm3 -> GEGLU feed-forward -> + residual of the 64x64 level as four launches")
    ap.add_argument("--no-row-linear", action="store_true",
                    help="A/B: the 64x64-level K = N = 320 projections on the tiled kernel and LayerNorm -> attn2.to_q as two launches")
    return ap.parse_args()


def build_pipeline(device, dtype, rank):
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter import attention_processor as AP
    from imagdressing_amd.adapter.resampler import Resampler
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from imagdressing_amd.scheduler import DDIMScheduler
    unet = E.UNet2DConditionModel.random_init(seed=0, device=device, dtype=dtype)
    ref_unet = E.UNet2DConditionModel.random_init(seed=1, device=device, dtype=dtype)
    boc = unet.cfg["block_out_channels"]
    procs = {}
    g = torch.Generator(device="cpu").manual_seed(2)
    for name in unet.attn_processors.keys():
        if name.startswith("mid_block"):
            hs = boc[-1]
        elif name.startswith("up_blocks"):
            hs = list(reversed(boc))[int(name[len("up_blocks.")])]
        else:
            hs = boc[int(name[len("down_blocks.")])]
        if name.endswith("attn1.processor"):
            p = AP.RefSAttnProcessor2_0(name, hs)                  # inference_IMAGdressing.py:80
            with torch.no_grad():                                  # fan-in scaled synthetic weights
                p.to_k_ref.weight.copy_(torch.randn(hs, hs, generator=g) * hs ** -0.5)
                p.to_v_ref.weight.copy_(torch.randn(hs, hs, generator=g) * hs ** -0.5)
        else:
            p = AP.CAttnProcessor2_0(name, hidden_size=hs, cross_attention_dim=768)   # :82
        procs[name] = p
    unet.set_attn_processor(procs)
    ref_unet.set_attn_processor({n: AP.CacheAttnProcessor2_0() for n in ref_unet.attn_processors.keys()})   # :93-94
    torch.manual_seed(3)
    proj = Resampler(dim=768, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=768, ff_mult=4)   # :55-64
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)          # :119-127
    pipe = IMAGDressing_v1(vae=None, reference_unet=ref_unet, unet=unet, tokenizer=None, text_encoder=None, image_encoder=None,
                           ImgProj=proj, scheduler=sch, safety_checker=None, feature_extractor=None)
    return pipe


def synthetic_inputs(width, height, batch, device, dtype, rank, world):
    """Synthetic conditioning of one garment + `batch * world` initial latents; the garment has the generation resolution
    (the reference resizes it to the output size, inference_IMAGdressing.py:165-167)."""
    gen = torch.Generator(device="cpu").manual_seed(1234)
    lh, lw = height // 8, width // 8
    n_total = batch * world
    inp = dict(
        prompt_embeds=(torch.randn(1, 77, 768, generator=gen) * 0.5).to(device),
        negative_prompt_embeds=(torch.randn(1, 77, 768, generator=gen) * 0.5).to(device),
        ref_clip_hidden_states=(torch.randn(1, 257, 1280, generator=gen) * 0.5).to(device=device, dtype=dtype),
        ref_image_latents=(torch.randn(1, 4, lh, lw, generator=gen)).to(device),
        # per-image seeds 42, 43, ... drawn on the CPU (identical on every vendor); rank r owns its block
        latents=torch.stack([torch.randn(4, lh, lw, generator=torch.Generator().manual_seed(42 + i))
                             for i in range(n_total)]).to(device),
    )
    return inp


def attn_flops_hybrid_level0(batch, N, M, C):
    """Algorithmic FLOPs of ONE launch of the fused attention kernel at UNet level 0 in the CFG batch:
    `batch` cond rows run self + garment attention (4 N^2 C + 4 N M C), `batch` uncond rows run self only
    (projections are separate GEMM launches and are not counted here)."""
    return batch * (4.0 * N * N * C + 4.0 * N * M * C) + batch * (4.0 * N * N * C)


def attn_bytes_hybrid_level0(batch, N, M, C=320, H=8, dpk=48):
    """Algorithmic HBM bytes of the same launch (16-bit operands): Q and K rows (padded to 48 per head as stored), V^T rows, the garment
    K / V^T once per head, O written."""
    B, D = 2 * batch, C // H
    return 2 * (B * H * N * dpk * 2) + B * H * D * N * 2 + H * (M * dpk + D * M) * 2 + B * N * C * 2


def cpu_baseline(args):
    """Reference-semantics CPU port (oracle/: reference processors' math + restated diffusers UNet, fp32
    torch) timed on this host: `cpu-baseline-steps` DDIM steps at batch 1 = 2 B=1 UNet forwards each
    (IMAGDressing_v1_pipeline.py:499-518), extrapolated linearly to 50 steps (the one-off garment pass is
    < 2 % of a run and is left out of the sample, which makes the CPU number slightly optimistic).
    Attention products go through ``F.scaled_dot_product_attention`` exactly as the reference's processors call it
    (adapter/attention_processor.py:589,607; ``oracle.processors.reference_sdpa_dispatch``) -- the oracle's explicit
    [B, 8, N, N] softmax is 5-10x slower on a CPU and would understate the reference (round-5 review)."""
    from imagdressing_amd import unet as E
    from oracle import processors as OP
    from oracle import sd15
    from oracle.ddim import DDIMOracle
    torch.manual_seed(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))     # more torch threads than ~16 only adds contention at batch 1
    torch.set_num_threads(cores)
    lat_h, lat_w = (args.height or args.res) // 8, (args.width or args.res) // 8
    with torch.no_grad(), OP.reference_sdpa_dispatch():
        u = sd15.UNet2DConditionModel()       # default-initialised fp32 weights (values do not affect timing)
        names = list(u.attn_processors.keys())
        boc = sd15.SD15["block_out_channels"]
        procs, sa = {}, {}
        lvl_tokens = {lv: (lat_h >> lv) * (lat_w >> lv) for lv in range(4)}
        for n in names:
            if n.startswith("mid_block"):
                hs, lv = boc[-1], 3
            elif n.startswith("up_blocks"):
                i = int(n[len("up_blocks.")]); hs, lv = list(reversed(boc))[i], 3 - i
            else:
                i = int(n[len("down_blocks.")]); hs, lv = boc[i], i
            if n.endswith("attn1.processor"):
                procs[n] = OP.RefSAttn(n, hs)
                sa[n] = torch.randn(1, lvl_tokens[lv], hs)
            else:
                procs[n] = OP.CAttn(n, hs, 768)
        u.set_attn_processor(procs)
        sch = DDIMOracle(); ts = sch.set_timesteps(args.ddim_steps)
        z = torch.randn(1, 4, lat_h, lat_w); pe = torch.randn(1, 77, 768) * 0.5; ne = torch.randn(1, 77, 768) * 0.5
        per_step, note = [], ""
        t0 = time.time()
        u(z, ts[0], ne)        # one untimed forward: first-touch of the 3.4 GB of fp32 weights and the allocator's arenas is not the path's cost
        warm_s = time.time() - t0
        for i in range(args.cpu_baseline_steps):
            t = ts[i]
            t0 = time.time()
            ec = u(z, t, pe, cross_attention_kwargs={"sa_hidden_states": sa})      # cond pass, garment branch on
            t_c = time.time() - t0
            if t_c > 60.0:      # keep the default bench run bounded on slow hosts
                per_step.append(t_c * (1.0 + 0.8))
                note = "; uncond pass not run (cond pass > 60 s), estimated as 0.8 x cond"
                break
            t0 = time.time()
            eu = u(z, t, ne)                                                         # uncond pass
            t_u = time.time() - t0
            z = sch.step(eu + 7.5 * (ec - eu), t, z)
            per_step.append(t_c + t_u)
        dt = sum(per_step) / len(per_step)
    return dict(value=1.0 / (dt * args.ddim_steps), unit="images/s", cores=cores, kind="port",
                attention="F.scaled_dot_product_attention, as the reference (adapter/attention_processor.py:589,607)",
                sample=f"{len(per_step)} of {args.ddim_steps} DDIM steps at batch 1 (reference loop semantics: cond + uncond fp32 UNet "
                       f"forward per step, {dt:.2f} s/step mean of {[round(t, 2) for t in per_step]}, torch {torch.__version__} on {cores} threads), "
                       f"extrapolated x{args.ddim_steps}{note}; one untimed warm-up forward first ({warm_s:.1f} s)")


def kernel_level_baseline(device, dtype, cores):
    """BASELINE.md section 4.1 / SURVEY 8(d) CPU-baseline plan (i): the hybrid processor ALONE -- ``RefSAttnProcessor2_0`` semantics
    (oracle/processors.py::hybrid_self_attention, pinned on the reference source, with its two attention products through
    F.scaled_dot_product_attention like the reference's :589,607; fp32 torch on `cores` threads, 2 warm-up + 5 timed calls)
    beside the HIP processor (three launches: q/k/v projection, fused two-softmax attention, out-projection + bias; HIP events over 20
    calls after 3 warm-up, garment K / V cached as in the loop) at the four (C, N = M) shapes of the 512x512 UNet, batch 1, garment
    branch on.  FLOPs per call: 8 N C^2 + 4 N^2 C + 4 N M C (the garment K / V projection is once per garment: excluded on both sides).
    The HIP figure is what a CALLER of the processor sees: below ~75 us it is the Python plugin surface + three launches, not the kernels."""
    from imagdressing_amd.adapter import attention_processor as AP
    from imagdressing_amd.unet import Attention
    from oracle import processors as OP
    torch.set_num_threads(cores)
    rows = []
    g = torch.Generator().manual_seed(77)
    for C, N in ((320, 4096), (640, 1024), (1280, 256), (1280, 64)):
        M = N
        w = {k: torch.randn(C, C, generator=g) * C ** -0.5 for k in ("wq", "wk", "wv", "wo", "wkr", "wvr")}
        bo = torch.randn(C, generator=g) * 0.1
        x, ref = torch.randn(1, N, C, generator=g), torch.randn(1, M, C, generator=g)
        fl = 8.0 * N * C * C + 4.0 * N * N * C + 4.0 * N * M * C
        with torch.no_grad():
            def cpu_call():      # (under reference_sdpa_dispatch below: SDPA as the reference's processor calls it)
                return OP.hybrid_self_attention(x, w["wq"], w["wk"], w["wv"], w["wo"], bo, 8, ref=ref, wk_ref=w["wkr"], wv_ref=w["wvr"], scale=1.0)
            with OP.reference_sdpa_dispatch():
                for _ in range(2):
                    cpu_call()
                t0 = time.perf_counter()
                for _ in range(5):
                    cpu_out = cpu_call()
                cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
            sd = {"a.to_q.weight": w["wq"], "a.to_k.weight": w["wk"], "a.to_v.weight": w["wv"], "a.to_out.0.weight": w["wo"], "a.to_out.0.bias": bo}
            attn = Attention(sd, "a", 8, str(device), dtype)
            proc = AP.RefSAttnProcessor2_0("blk.attn1.processor", C)
            proc.to_k_ref.weight.copy_(w["wkr"]); proc.to_v_ref.weight.copy_(w["wvr"])
            attn.set_processor(proc)
            xd, sa = x.to(device=device, dtype=dtype), {"blk.attn1.processor": ref.to(device)}
            for _ in range(3):
                out = attn(xd, sa_hidden_states=sa)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = attn(xd, sa_hidden_states=sa)
            e1.record(); torch.cuda.synchronize()
            hip_us = e0.elapsed_time(e1) / 20 * 1e3
            err = (out.float().cpu() - cpu_out).abs().max().item()
        rows.append({"C": C, "N": N, "M": M, "gflop": round(fl / 1e9, 3), "cpu_ms": round(cpu_ms, 3), "cpu_gflops": round(fl / cpu_ms / 1e6, 1),
                     "hip_us": round(hip_us, 2), "hip_tflops": round(fl / hip_us / 1e6, 1), "speedup": round(cpu_ms * 1e3 / hip_us, 1),
                     "max_abs_diff_hip_vs_cpu": round(err, 5)})
    return {"what": "hybrid attention processor alone (RefSAttnProcessor2_0, garment branch on, batch 1): fp32 CPU port of the reference processor (SDPA, as the reference) vs "
                    f"the HIP processor ({str(dtype).replace('torch.', '')}) called through the plugin surface (host-bound below ~75 us per call), per UNet level of the 512x512 geometry",
            "cores": cores, "shapes": rows}


def run_other_config(cid, what, device, dtype, args, runs=3):
    """Timed runs of another single-GPU BASELINE configuration (tools/configs.py builds pipeline + synthetic inputs resident in HBM):
    one warm-up run, then `runs` timed runs of the whole call -- garment pass, ControlNet + UNet loop, VAE decode of the images."""
    import importlib
    configs = importlib.import_module("tools.configs")
    from imagdressing_amd import ops
    from imagdressing_amd.vae import AutoencoderKL
    pipe, kw = configs.build(cid, device, dtype, None, args.ddim_steps)
    if args.decode:
        pipe.vae = AutoencoderKL.random_init(seed=5, device=device, dtype=dtype)
        kw["output_type"] = "pt"
    B = kw["num_images_per_prompt"]
    out = pipe(**kw).images
    torch.cuda.synchronize()
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        out = pipe(**kw).images
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best, mean = min(times), sum(times) / len(times)
    res = {"workload": what + f"; {args.dtype}, random-init weights" + (", VAE decode included" if args.decode else ""),
           "value": round(B / mean, 4), "unit": "images/s", "images": B, "runs": runs, "warmup": 1,
           "ms_per_run": round(mean * 1e3, 2), "ms_per_run_best": round(best * 1e3, 2), "ms_per_ddim_step": round(mean * 1e3 / args.ddim_steps, 3),
           "outputs_finite": bool(torch.isfinite(out).all().item()), "output_shape": list(out.shape)}
    del pipe, out, kw
    return res


def self_launch(args):
    """``python bench.py --gpus N`` without a launcher: re-exec under ``torch.distributed.run`` with one rank per GPU
    (what the driver does explicitly), rendezvous on 127.0.0.1.  Returns the child's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if args.device < 0 and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this driver (RCCL / CUDA-tensor sharing)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)     # "nccl" == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.backend)

    from imagdressing_amd import ops
    if args.attn_qw:
        ops.L.check(ops.L.load().imd_set_tuning(0, args.attn_qw))
    if args.gemm_flags >= 0:
        ops.L.check(ops.L.load().imd_set_tuning(2, args.gemm_flags))
    if args.splitk_in_kernel:
        ops.SPLITK_IN_KERNEL = True
    if args.no_fused_ff:
        ops.FUSED_FF = False
    if args.no_row_linear:
        ops.FUSED_LN = False
        for ent in ops._gemm_table().values():
            for k in ("cfg", "cfg_nosplit"):
                if ent.get(k) in (12, 15):
                    ent[k] = 4
                if ent.get(k) in (13, 14):
                    ent[k] = 2
    W0 = args.width or args.res
    H0 = args.height or args.res
    is_headline_geometry = (W0 == 512 and H0 == 512)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def run_timed(dtype, width, height, batch, steps, warmup, decode, graph=False, hook_attention=True, keep_output=False):
        """warmup + EXACTLY `steps` timed steps in `dtype`; -> dict(elapsed s [max over ranks], per-rank seconds, hybrid-attention
        event times, VAE-decode event times, outputs finite[, last output])"""
        pipe = build_pipeline(device, dtype, rank)
        if decode:
            from imagdressing_amd.vae import AutoencoderKL
            pipe.vae = AutoencoderKL.random_init(seed=5, device=device, dtype=dtype)      # inference_IMAGdressing.py:42
        pipe.enable_step_graph(graph)
        inp = synthetic_inputs(width, height, batch, device, dtype, rank, world)
        N0 = (width // 8) * (height // 8)
        dec_events = []
        if decode:           # bracket the decode on the launch stream: latent-out time = step - decode
            orig_decode = pipe._decode

            def timed_decode(latents, output_type, generator=None):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig_decode(latents, output_type, generator)
                e1.record()
                dec_events.append((e0, e1))
                return r
            pipe._decode = timed_decode

        def one_step():
            return pipe(prompt=None, null_prompt=None, negative_prompt=None, ref_image=None, width=width, height=height,
                        num_inference_steps=args.ddim_steps, guidance_scale=7.5, num_images_per_prompt=batch * world,
                        image_scale=1.0, output_type="pt" if decode else "latent", shard_over_ranks=world > 1, **inp).images
        for _ in range(warmup):
            out = one_step()
        dec_events.clear()
        # roofline hook: bracket every level-0 hybrid-attention launch of the timed region with HIP events
        # (B == 2 * batch: the launch attn_flops_hybrid_level0 prices -- `batch` two-phase + `batch` one-phase rows.  The first hybrid block of a
        # step runs the cond rows only with its first phase stored twice (round 6, imd_attn_params.out_dup): fewer FLOPs, not this launch.)
        # (round 6) every FOURTH such launch is bracketed (50 of the 200 per bench step): an event pair between two kernels costs the stream ~6 us in front of
        # the launch and ~1.5 us behind it (tools/rocprof_sequence.py, profiles/r6r_*: 47 us of gaps per DDIM step with every launch bracketed)
        seen_l0 = [0]

        def match_l0(B, H, N, D, L1, L2):
            if not (D == 40 and N == N0 and L2 == N0 and B == 2 * batch):
                return False
            seen_l0[0] += 1
            return seen_l0[0] % 4 == 0
        hook = {"match": match_l0, "events": []}
        if hook_attention and not graph:
            ops.ATTN_EVENT_HOOK = hook
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = one_step()
        barrier()
        elapsed = mine = time.perf_counter() - t0
        ops.ATTN_EVENT_HOOK = None
        per_rank = [mine]
        if world > 1:
            import torch.distributed as dist
            tall = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
            dist.all_gather(tall, torch.tensor([mine], dtype=torch.float64, device=device))
            per_rank = [float(t.item()) for t in tall]
            elapsed = max(per_rank)
        res = dict(elapsed=elapsed, per_rank=per_rank, finite=bool(torch.isfinite(out).all().item()),
                   att_ms=[a.elapsed_time(b) for a, b in hook["events"]], dec_ms=[a.elapsed_time(b) for a, b in dec_events], N0=N0)
        if keep_output:
            res["out"] = out.detach().float().cpu()
        del pipe, out
        ops.clear_workspaces()
        torch.cuda.empty_cache()
        return res

    def live_traffic():
        """HBM bytes of ONE level-0 hybrid-attention launch from the PMC counters, measured NOW when rocprofv3 is on PATH: two
        separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/attn_bench.py --default-only with --kernel-trace only, corrected as
        MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE tallies 128-B requests as 64 B: x2; units of 1 KiB)."""
        import csv
        import glob
        import shutil
        import signal
        import subprocess
        import tempfile
        exe = shutil.which("rocprofv3")
        if exe is None or world > 1 or args.no_live_traffic or not is_headline_geometry or args.batch != 4:
            return None
        vals = {}
        tmp = tempfile.mkdtemp(prefix="imd_pmc_", dir="/tmp")
        try:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "a", "--", sys.executable,
                       os.path.join(ROOT, "tools", "attn_bench.py"), "--default-only", "--iters", "2", "--dtype", args.dtype]
                pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                      start_new_session=True)
                try:
                    pr.wait(timeout=150)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)
                    return None
                per = []
                for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if "attn40_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                                per.append(float(row["Counter_Value"]))
                if not per:
                    return None
                vals[counter] = sum(per) / len(per)
            return dict(traffic=int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024),
                        source=f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/attn_bench.py --default-only, measured in this run: "
                               f"FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB x2 (gfx950 correction) + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KiB per launch")
        except Exception:          # noqa: BLE001  (a profiler problem must not cost the bench line)
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def roofline_of(r, width, height, batch, measure_traffic=False):
        att_ms = r["att_ms"]
        if not att_ms:
            return None
        avg_s = sum(att_ms) / len(att_ms) * 1e-3
        fl = attn_flops_hybrid_level0(batch, r["N0"], r["N0"], 320)
        fused_proj = bool(ops.FUSED_OUT_PROJ and ops.attention_proj_supported(8, r["N0"], 40))
        if fused_proj:                 # the launch also carries to_out[0] + bias + residual of the block (ABI v7): 2 B N C^2
            fl += 2.0 * (2 * batch) * r["N0"] * 320 * 320
        ach = fl / avg_s / 1e12
        traffic = tsrc = None
        if measure_traffic:
            lt = live_traffic()
            if lt is not None:
                traffic, tsrc = lt["traffic"], lt["source"]
        if traffic is None:            # committed PMC passes of this kernel (same shape only)
            for rel in (("profiles", "pmc_r3", "attn_level0_traffic.json"), ("profiles", "pmc_r2", "attn_level0_traffic.json")):
                tpath = os.path.join(ROOT, *rel)
                if batch == 4 and width == 512 and height == 512 and os.path.isfile(tpath):
                    with open(tpath) as f:
                        traffic = json.load(f).get("traffic_bytes")
                    tsrc = "/".join(rel) + " (rocprofv3 --pmc passes of this kernel and shape; not re-measured in this run)"
                    break
        return dict(bound="mfma", kernel="fused hybrid attention (d = 40), UNet level 0, CFG batch" + (" + out-projection + residual in the same launch" if fused_proj else ""),
                    achieved=round(ach, 2), peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_PEAK_TFLOPS, 4),
                    traffic=traffic, traffic_source=tsrc, launches=len(att_ms), avg_launch_ms=round(avg_s * 1e3, 4),
                    flops_per_launch=fl, algorithmic_bytes_per_launch=attn_bytes_hybrid_level0(batch, r["N0"], r["N0"]))

    # ---- multi-GPU self-description (what proves that N ranks on N distinct devices took part, and what the one collective costs) ----
    multi = None
    if world > 1:
        import torch.distributed as dist
        pr = torch.cuda.get_device_properties(dev_index)
        ident = str(getattr(pr, "uuid", "")) or f"{pr.name}/pci{getattr(pr, 'pci_bus_id', '?')}:{getattr(pr, 'pci_device_id', '?')}"
        idents = [None] * world
        dist.all_gather_object(idents, (rank, dev_index, ident))
        distinct = len({(d, i) for _, d, i in idents})
        if distinct < world and args.device < 0:
            raise SystemExit(f"bench.py: {world} ranks but only {distinct} distinct devices {idents}: two ranks share a GPU (pass --device only for rehearsals)")
        # the ONE collective of the path: the packed garment-feature broadcast (23.1 MB at 512x512 in 16 bits), HIP-event timed
        nel = sum(t * c for c, t in ((320, 5 * (W0 // 8) * (H0 // 8)), (640, 5 * (W0 // 16) * (H0 // 16)), (1280, 5 * (W0 // 32) * (H0 // 32)),
                                     (1280, (W0 // 64) * (H0 // 64)))) + 1
        buf = torch.zeros(nel, dtype=dtype, device=device)
        bms = []
        for _ in range(4):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); dist.broadcast(buf, src=0); e1.record(); torch.cuda.synchronize()
            bms.append(e0.elapsed_time(e1))
        multi = dict(ranks_seen=len(idents), distinct_devices=distinct, devices=[f"rank{r}:cuda{d}:{i}" for r, d, i in idents],
                     backend=args.backend + (" (RCCL)" if args.backend == "nccl" else ""), broadcast_bytes=nel * 2,
                     broadcast_ms=round(min(bms[1:]), 3))

    primary = run_timed(dtype, W0, H0, args.batch, args.steps, args.warmup, args.decode)
    secondary = None
    other = "fp16" if args.dtype == "bf16" else "bf16"
    odt = torch.float16 if other == "fp16" else torch.bfloat16
    if not args.no_secondary:
        r2 = run_timed(odt, W0, H0, args.batch, args.steps, args.warmup, args.decode)
        roof2 = roofline_of(r2, W0, H0, args.batch)
        secondary = {"dtype": other, "value": round(args.batch * world * args.steps / r2["elapsed"], 4), "unit": "images/s",
                     "ms_per_step": round(r2["elapsed"] / args.steps * 1e3, 2), "outputs_finite": r2["finite"],
                     "roofline_frac": None if roof2 is None else roof2["frac"],
                     "note": "same binary, same workload, the other 16-bit element type (same MFMA rate)"}
    geometry = None
    if not args.no_geometry_secondary and is_headline_geometry:
        gsteps = args.steps                       # same timed steps / warm-up as the headline (round 4; was 2 after 1)
        rg = run_timed(dtype, 512, 640, args.batch, gsteps, args.warmup, args.decode)
        roofg = roofline_of(rg, 512, 640, args.batch)
        geometry = {"workload": "the reference scripts' own default geometry: width 512 x height 640, garment 640x512 (inference_IMAGdressing.py:182-183), "
                                f"latent 80x64, N = M = 5120 / 1280 / 320 / 80; {args.dtype}, batch {args.batch}/GPU, {args.ddim_steps} DDIM steps",
                    "value": round(args.batch * world * gsteps / rg["elapsed"], 4), "unit": "images/s", "steps": gsteps, "warmup": args.warmup,
                    "ms_per_step": round(rg["elapsed"] / gsteps * 1e3, 2), "outputs_finite": rg["finite"],
                    "roofline_frac": None if roofg is None else roofg["frac"],
                    "hybrid_attention_tflops": None if roofg is None else roofg["achieved"]}

    # ---- everything below runs on rank 0 of a single-GPU run only (untimed diagnostics) ----
    parity = latency = flops = None
    if world == 1 and not args.no_parity:
        try:
            from tests.unet_fixture import ipa_controlnet_forward_inputs, measure_unet_parity_timesteps, unet_forward_inputs
            import torch as _t
            gold = _t.load(os.path.join(ROOT, "tests", "golden", "unet_forward_full.pt"), weights_only=False)["latent_64x64"]
            gold_t = _t.load(os.path.join(ROOT, "tests", "golden", "unet_forward_timesteps.pt"), weights_only=False)
            base_inputs = unet_forward_inputs(64, 64, gold)
            ipa_inputs = ipa_controlnet_forward_inputs(gold_t["ipa_controlnet"], base=base_inputs)
            parity = {"what": "full-width (859.5 M parameters) cond + uncond UNet forwards on the 64x64 latent at t = 981, 481 and 1 (the first, a middle and the "
                              "last step of the 50-step schedule) against committed fp32-oracle outputs (tests/golden/unet_forward_full.pt, "
                              "unet_forward_timesteps.pt; inputs regenerated from seeds and digest-checked), measured in this process: "
                              "`refs` = RefS + CAttn processors, garment branch on the cond row (configs[1]); `ipa_controlnet` = LoraRefS + LoRAIP "
                              "processors (rank-128 LoRA, 77 + 4 tokens) with the engine's own pose-ControlNet residuals added (configs[2]).  "
                              "oracle/sd15.py is the unpinned restatement of diffusers 0.24, its processors are pinned on the reference source; error of eps (std ~0.6)",
                      "north_star_bar": "atol 1e-2"}
            for nm, d_ in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
                r = measure_unet_parity_timesteps(device, d_, inputs=base_inputs, ipa_inputs=ipa_inputs)
                # legacy single-timestep view (t = 481, configs[1] processors) + the worst element over everything measured
                parity[nm] = dict(r["refs"]["t481"], timesteps=r, max_abs=max(r["refs"]["max_abs"], r["ipa_controlnet"]["max_abs"]),
                                  finite=all(v["finite"] for c in r.values() for k, v in c.items() if k.startswith("t")))
                parity[nm]["meets_atol_1e-2"] = bool(parity[nm]["max_abs"] <= 1e-2)
            parity["meets_atol_1e-2"] = [nm for nm in ("fp16", "bf16") if parity[nm]["meets_atol_1e-2"]]
            del ipa_inputs
            # full-LENGTH trajectories against the committed fp32-oracle latents (tests/golden/trajectory.pt): the whole pipeline call --
            # Resampler, garment pass, 20 / 50 DDIM steps, CFG -- for BASELINE configs[0] (20 steps, seed 42) and configs[1] (50 steps,
            # seeds 42 / 43; then the batch-4 call and its HIP-graph replay, rows 0 / 1 against the same goldens)
            try:
                from tests import trajectory_fixture as TF
                traj = {"what": "final latent (and latents after a few steps) of the whole pipeline call against oracle/pipeline.py::denoise on the fp32 oracle "
                                "(tests/golden/trajectory.pt).  HEADLINE figures are relative to the oracle latent's own scale (its sigma is ~ 24 with the "
                                "random-init weights, so an absolute 1e-2 on a latent means nothing): rel_rms = rms error / rms of the oracle latent, "
                                "max_abs_over_sigma = worst element / sigma, frac_within_1e-2_sigma = share of elements within 1e-2 sigma.  The trajectory "
                                "composes the UNet 20 / 50 times at guidance 7.5; measured: the per-forward error does not grow along it"}
                for nm, d_ in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
                    r = TF.measure_trajectory_parity(device, d_, cases=("configs0_20step", "configs1_50step"), base=base_inputs)
                    traj[nm] = {"configs0_20step_final": r["configs0_20step"]["seed42"]["final"],
                                "configs1_50step_final": {k: v["final"] for k, v in r["configs1_50step"].items() if k.startswith("seed")},
                                "configs1_50step_batch4": r["configs1_50step"].get("batch4"), "configs1_50step_batch4_graph": r["configs1_50step"].get("batch4_graph"),
                                "worst_final_rel_rms": max(r["configs0_20step"]["worst_final_rel_rms"], r["configs1_50step"]["worst_final_rel_rms"]),
                                "worst_final_max_abs_over_sigma": max([r["configs0_20step"]["seed42"]["final"]["max_abs_over_sigma"]] +
                                                                      [v["final"]["max_abs_over_sigma"] for k, v in r["configs1_50step"].items() if k.startswith("seed")]),
                                "detail": r}
                TF.clear_input_cache()
                parity["trajectory"] = traj
            except Exception as e:       # noqa: BLE001
                parity["trajectory"] = {"error": f"{type(e).__name__}: {e}"}
            del base_inputs
            ops.clear_workspaces(); torch.cuda.empty_cache()
        except Exception as e:       # noqa: BLE001
            parity = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_latency:
        try:
            le = run_timed(dtype, W0, H0, 1, 3, 1, args.decode, graph=False, hook_attention=False, keep_output=True)
            lg = run_timed(dtype, W0, H0, 1, 3, 1, args.decode, graph=True, hook_attention=False, keep_output=True)
            latency = {"what": f"ONE image (the reference's literal usage, batch_size = 1, IMAGDressing_v1_pipeline.py:389), {W0}x{H0}, {args.ddim_steps} DDIM steps, "
                               f"{args.dtype}, garment pass" + (" and VAE decode" if args.decode else "") + " included; mean of 3 after 1 warm-up",
                       "eager_ms": round(le["elapsed"] / 3 * 1e3, 2), "graph_ms": round(lg["elapsed"] / 3 * 1e3, 2),
                       "graph": "HIP-graph replay of the DDIM step (pipe.enable_step_graph(); step 0 eager, step 1 captured, steps 1..S-1 replayed)",
                       "bit_identical": bool(torch.equal(le["out"], lg["out"]))}
        except Exception as e:       # noqa: BLE001
            latency = {"error": f"{type(e).__name__}: {e}"}
            ops.ATTN_EVENT_HOOK = None
    if world == 1 and not args.no_flops:
        try:
            ops.FLOP_COUNTER = {}
            run_timed(dtype, W0, H0, args.batch, 1, 0, args.decode, hook_attention=False)
            cnt, ops.FLOP_COUNTER = ops.FLOP_COUNTER, None
            tot = sum(cnt.values())
            flops = {"what": "algorithmic FLOPs of ONE bench step (garment pass + CFG-batched DDIM steps" + (" + VAE decode" if args.decode else "") +
                             "): 2 M N K per GEMM / convolution launch, 4 N L d per attention key set, padding and recomputation excluded",
                     "per_bench_step": tot, "attention": cnt.get("attention", 0.0), "gemm_conv": cnt.get("gemm_conv", 0.0),
                     "per_ddim_step_approx": tot / args.ddim_steps,
                     "mfma_frac_end_to_end": round(tot / (primary["elapsed"] / args.steps) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        except Exception as e:       # noqa: BLE001
            ops.FLOP_COUNTER = None
            flops = {"error": f"{type(e).__name__}: {e}"}

    other_configs = None
    if world == 1 and not args.no_configs and is_headline_geometry:
        other_configs = {}
        for key, cid, what in (("configs2", 3, "BASELINE configs[2]: + IP-Adapter face (LoRAIP, 77 + 4 tokens) + ControlNet-OpenPose, LoraRefS / LoRAIP processors with "
                                               "rank-128 LoRA folded into the weights, 512x512, 50 DDIM steps, batch 8, guidance 7.0 "
                                               "(inference_IMAGdressing_ipa_controlnetpose.py:218-237)"),
                               ("configs4", 5, "BASELINE configs[4] on ONE GPU: ControlNet-inpainting path, 768x576 (latent 96x72), 50 DDIM steps, 4 images "
                                               "(= 32 / 8 GPUs), guidance 5.0, inpaint blend every step, 16-bit attention (the MX-fp8 attention kernel is slower: opt-in) "
                                               "(inference_IMAGdressing_controlnetinpainting.py:213-229)")):
            try:
                other_configs[key] = run_other_config(cid, what, device, dtype, args)
            except Exception as e:       # noqa: BLE001
                other_configs[key] = {"error": f"{type(e).__name__}: {e}"}
            ops.clear_workspaces(); torch.cuda.empty_cache()

    power = None
    if world == 1 and not args.no_power and is_headline_geometry and args.batch == 4:
        # board power / shader clock while the level-0 hybrid-attention kernel runs back to back (rocm-smi, 20 Hz, ~2 s): the kernel is
        # POWER-capped -- with random operands the board sits at its cap and the governor lowers the clock, so the 2.5 PFLOP/s
        # datasheet peak (2.4 GHz) is not reachable by ANY kernel with this operand activity; roofline.frac stays priced against it
        try:
            import shutil
            import subprocess
            if shutil.which("rocm-smi"):
                pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "power_probe.py"), "--seconds", "2", "--dtype", args.dtype],
                                    capture_output=True, text=True, timeout=120)
                rec = json.loads([l for l in pr.stdout.splitlines() if l.startswith("{")][-1])
                power = {"what": "rocm-smi while the level-0 hybrid-attention launch (the roofline kernel, same operands as tools/attn_bench.py) runs back to back for 2 s",
                         "board_power_w": rec["power_w"], "board_power_cap_w": 1400.0, "sclk_mhz": rec["sclk_mhz"], "sclk_max_mhz": 2400.0,
                         "us_per_launch_back_to_back": rec["us_per_launch"]}
        except Exception as e:       # noqa: BLE001
            power = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        images = args.batch * world * args.steps
        elapsed, dec_ms = primary["elapsed"], primary["dec_ms"]
        roof = roofline_of(primary, W0, H0, args.batch, measure_traffic=True)
        if roof is not None and power is not None and power.get("sclk_mhz"):
            roof["power"] = power
            roof["frac_of_peak_at_sustained_clock"] = round(roof["achieved"] / (MFMA_PEAK_TFLOPS * power["sclk_mhz"] / 2400.0), 4)
        geo = f"{W0}x{H0}"
        # both 16-bit element types side by side, each with ITS measured parity (the reference computes in fp16, BASELINE.json's config names
        # bf16): same instruction streams except the convert instructions -- the fp16 build runs the dense kernels at a ~4 % lower shader
        # clock (profiles/r4b_power_fp16_vs_bf16.jsonl), which is the whole speed difference
        by_dtype = None
        if secondary is not None:
            def ent(dt_name, value, ms, frac, finite):
                par = (parity or {}).get(dt_name) if isinstance(parity, dict) else None
                return {"value": value, "unit": "images/s", "ms_per_step": ms, "hybrid_attention_roofline_frac": frac, "outputs_finite": finite,
                        "parity_max_abs_eps_error": None if not par else par.get("max_abs"),
                        "meets_north_star_atol_1e-2": None if not par else par.get("meets_atol_1e-2")}
            by_dtype = {args.dtype: ent(args.dtype, round(images / elapsed, 4), round(elapsed / args.steps * 1e3, 2), None if roof is None else roof["frac"], primary["finite"]),
                        other: ent(other, secondary["value"], secondary["ms_per_step"], secondary["roofline_frac"], secondary["outputs_finite"])}
            ok = [k for k, v in by_dtype.items() if v["meets_north_star_atol_1e-2"]]
            by_dtype["parity_qualified"] = ({"dtype": max(ok, key=lambda k: by_dtype[k]["value"]), "value": max(by_dtype[k]["value"] for k in ok), "unit": "images/s"}
                                            if ok else None)
        line = {
            "metric": "512x512 50-step images/sec (whole node)", "value": round(images / elapsed, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: SD1.5 {args.dtype}, {geo}, {args.ddim_steps} DDIM steps, "
                                   f"batch {args.batch}/GPU sharing one garment, garment cross-attn only (RefS + CAttn processors), "
                                   "random-init weights" + (", VAE decode of the final latents included" if args.decode else ""),
                       "images_per_gpu": args.batch, "global_batch": args.batch * world, "guidance_scale": 7.5,
                       "parallelism": f"dp{world} (image shards; garment features broadcast once per batch)"},
            "outputs_finite": primary["finite"],
            # (round 6) the figures a truncated record tail loses, repeated up front: the other element type's value, which of the two meets the
            # north-star tolerance, the sustained shader clock of the roofline kernel
            "value_fp16": None if by_dtype is None else by_dtype.get("fp16", {}).get("value"),
            "value_bf16": None if by_dtype is None else by_dtype.get("bf16", {}).get("value"),
            "parity_qualified": None if by_dtype is None else by_dtype.get("parity_qualified"),
            "sclk_mhz": None if not power else power.get("sclk_mhz"),
            "roofline_frac": None if roof is None else roof.get("frac"),
            "roofline": roof,
            "decode_ms_per_step": (round(sum(dec_ms) / max(len(dec_ms), 1), 2) if dec_ms else None),
            "latent_out_ms_per_step": (round(elapsed / args.steps * 1e3 - sum(dec_ms) / max(len(dec_ms), 1), 2) if dec_ms else None),
            "secondary": secondary,
            "by_dtype": by_dtype,
            "other_configs": other_configs,
            "default_geometry_512x640": geometry,
            "parity": parity if parity is not None else {"note": "measured on single-GPU runs only (tests/unet_fixture.py::measure_unet_parity)"},
            "latency_b1": latency,
            "flops": flops,
        }
        if world > 1:
            multi["ms_per_step_per_rank"] = [round(t / args.steps * 1e3, 2) for t in primary["per_rank"]]
            line["multi_gpu"] = multi
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
            try:
                line["cpu_baseline"]["kernel_level"] = kernel_level_baseline(device, dtype, line["cpu_baseline"]["cores"])
            except Exception as e:       # noqa: BLE001
                line["cpu_baseline"]["kernel_level"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
