"""Shared machinery of the four ``IMAGDressing_v1`` pipeline classes.

The reference pipelines subclass diffusers' ``StableDiffusionPipeline`` /
``StableDiffusionControlNetInpaintPipeline`` (un-vendored).  Here the pipeline is a plain class that
keeps the reference's constructor kwargs, ``__call__`` kwargs, ``set_scale`` / ``set_ipa_scale`` and
return type, and re-designs the loop (IMAGDressing_v1_pipeline.py:463-541) MI355X-first:

* the reference issues two batch-1 UNet calls per step (cond with garment tokens, uncond without);
  here ONE UNet call runs a [2B] batch -- rows [0, B) cond, rows [B, 2B) uncond -- with the garment
  branch switched per row (``sa_batch_mask``), for B images that share the garment;
* garment features are harvested once per garment from the garment UNet run at batch 1 (the
  reference runs it at batch 2 and discards half, quirk 8 of SURVEY.md), and their K/V projections
  are cached inside the processors for the whole loop;
* CFG + DDIM step + (inpaint blend) + the next step's UNet input are one elementwise kernel over
  an fp32 latent state;
* under ``torch.distributed`` the B images are sharded over ranks and the garment features are
  broadcast from rank 0 (one collective per garment, none in the loop).

Encoders outside the hot path (CLIP text / vision, VAE -- SURVEY.md 8f "next") are duck-typed torch
modules supplied by the caller, or bypassed with pre-computed tensors (``prompt_embeds``,
``negative_prompt_embeds``, ``ref_clip_hidden_states``, ``ref_image_latents``, ``latents``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ... import ops
from ...adapter.attention_processor import (IPAttnProcessor2_0, LoRAIPAttnProcessor2_0, LoraRefSAttnProcessor2_0,
                                            RefSAttnProcessor2_0)
from ...unet import nchw_to_nhwc8

bf16 = torch.bfloat16


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]] = None


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """diffusers ``randn_tensor``: CPU generators sample on the CPU (cross-vendor reproducible)."""
    device = torch.device(device or "cpu")
    if isinstance(generator, (list, tuple)):
        return torch.cat([randn_tensor((1,) + tuple(shape[1:]), g, device, dtype) for g in generator], 0)
    gdev = generator.device if generator is not None else device
    if gdev.type != device.type:
        return torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)


class PipelineBase:
    vae_scale_factor = 8

    def _init_common(self, *, vae, reference_unet, unet, tokenizer, text_encoder, image_encoder, ImgProj, scheduler,
                     safety_checker=None, feature_extractor=None, controlnet=None):
        from ...hub import PendingModel

        def ready(m):          # a ``from_pretrained`` handle that never met ``.to(device=...)``: build it now (default device)
            return m._build() if isinstance(m, PendingModel) else m
        vae, reference_unet, unet, controlnet = ready(vae), ready(reference_unet), ready(unet), ready(controlnet)
        text_encoder, image_encoder = ready(text_encoder), ready(image_encoder)
        for role, m in (("unet", unet), ("reference_unet", reference_unet), ("controlnet", controlnet)):
            if isinstance(m, torch.nn.Module) and not hasattr(m, "forward_nhwc"):        # e.g. a stock diffusers UNet2DConditionModel
                raise TypeError(
                    f"IMAGDressing_v1({role}=...): got a {type(m).__module__}.{type(m).__name__}.  These pipelines drive the MI355X engine "
                    "UNet (imagdressing_amd.unet.UNet2DConditionModel / ControlNetModel; NHWC, fused CFG batch) -- build it from the same "
                    "checkpoint with `UNet2DConditionModel.from_pretrained(dir, subfolder='unet').to(dtype=torch.float16, device='cuda')` "
                    "imported from imagdressing_amd.unet, or put <repo>/compat on sys.path so that `from diffusers import "
                    "UNet2DConditionModel` resolves to it (INTEGRATION.md section 1).  The attention PROCESSORS alone also work on a "
                    "stock diffusers UNet (unet.set_attn_processor), but then the stock diffusers pipeline has to drive it.")
        self.vae, self.reference_unet, self.unet = vae, reference_unet, unet
        self.tokenizer, self.text_encoder, self.image_encoder = tokenizer, text_encoder, image_encoder
        self.ImgProj, self.scheduler, self.controlnet = ImgProj, scheduler, controlnet
        # the reference scripts pass the *classes* here (inference_IMAGdressing.py:133-134): tolerated, unused
        self.safety_checker, self.feature_extractor = safety_checker, feature_extractor
        self._cross_attention_kwargs = None
        self._garment_cache = None
        if vae is not None and hasattr(vae, "config") and hasattr(vae.config, "block_out_channels"):
            self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)

    # ---- reference surface ----
    @property
    def cross_attention_kwargs(self):
        return self._cross_attention_kwargs

    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    def set_tuning(self, attn_variant: Optional[int] = None, attn_xcd: Optional[bool] = None, gemm_flags: Optional[int] = None):
        """Kernel tuning of THIS pipeline's launches (head-dim-40 attention variant, XCD-aware attention work order, GEMM tuning bits 0..4),
        carried per call in the params blocks (``ops.tuning_scope``, ``IMD_TUNING_PER_CALL``): two pipelines of one process may differ and
        the library's process-wide knobs (``imd_set_tuning``) are not touched.  All None (the default) = the process-wide settings."""
        self._tuning = dict(attn_variant=attn_variant, attn_xcd=attn_xcd, gemm_flags=gemm_flags)
        self.release_step_graph()            # a captured step graph bakes its kernels in
        return self

    def enable_step_graph(self, flag: bool = True):
        """Opt in to HIP-graph replay of the DDIM denoising step (see ``denoise``): same kernels, same arithmetic, one graph launch
        per step instead of ~500 kernel launches.  Worth it where the loop is host-bound (small batches); ignored for UniPC, step
        callbacks, traces and per-step ControlNet gating."""
        self._step_graph = bool(flag)
        if not flag:
            self.release_step_graph()
        return self

    def release_step_graph(self):
        """Drop the captured step graph, the side stream it was recorded on and every scratch buffer keyed by that stream
        (``ops.clear_workspaces(stream=...)``): the graph pins a private memory pool, and the per-stream workspaces (attention operand
        buffers, GroupNorm partials, a >= 16 MB split-K slab) would otherwise live as long as the process."""
        side = self.__dict__.pop("_graph_stream", None)
        self.__dict__.pop("_last_step_graph", None)
        if side is not None:
            side.synchronize()
            ops.clear_workspaces(stream=side.cuda_stream)

    def __del__(self):
        try:
            self.release_step_graph()
        except Exception:        # noqa: BLE001  (interpreter shutdown)
            pass

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm.auto import tqdm
            return tqdm(iterable, total=total, disable=getattr(self, "_progress_disabled", True))
        except Exception:   # pragma: no cover
            class _N:
                def __enter__(s): return s
                def __exit__(s, *a): return False
                def update(s, *a): pass
            return _N()

    def set_progress_bar_config(self, disable=False, **kw):
        self._progress_disabled = disable

    # ---- encoders outside the hot path (duck-typed torch modules) ----
    def encode_prompt(self, prompt, device, num_images_per_prompt=1, do_classifier_free_guidance=True,
                      negative_prompt=None, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
        def enc(text):
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("no tokenizer/text_encoder: pass prompt_embeds / negative_prompt_embeds")
            ids = self.tokenizer(text, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                 return_tensors="pt").input_ids
            return self.text_encoder(ids.to(device))[0]
        if prompt_embeds is None:
            prompt_embeds = enc(prompt)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt_embeds = enc(negative_prompt if negative_prompt is not None else "")
        return prompt_embeds, negative_prompt_embeds

    def _clip_hidden(self, clip_image, device):
        dt = next(self.image_encoder.parameters()).dtype
        return self.image_encoder(clip_image.to(device, dtype=dt), output_hidden_states=True).hidden_states[-2]

    def prepare_latents(self, batch_size, num_channels_latents, width, height, dtype, device, generator, latents=None):
        # NB the reference passes (width, height) in this order to a (height, width) signature, and so keeps
        # shape [B, 4, width//8, height//8] semantics consistent with its own call (:440-448); we take them named.
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=torch.float32)
        return latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma

    # ---- garment features (A2 of SURVEY 8a) ----
    @torch.no_grad()
    def garment_features(self, ref_image_latents: torch.Tensor, cloth_proj_embed: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Garment UNet once at t = 0 with the 16 resampler tokens as context; returns the (post-LayerNorm)
        input of every attention layer, [1, M, C] each (IMAGDressing_v1_pipeline.py:465-480)."""
        with ops.tuning_scope(**(getattr(self, "_tuning", None) or {})):
            return self._garment_features(ref_image_latents, cloth_proj_embed)

    def _garment_features(self, ref_image_latents, cloth_proj_embed):
        dt = self.reference_unet.dtype
        x = nchw_to_nhwc8(ref_image_latents[:1].to(self.device), dt)
        ehs = cloth_proj_embed[-1:].to(device=self.device, dtype=dt).contiguous()     # the cond half ([1] of the CFG pair)
        self.reference_unet.forward_nhwc(x, 0, ehs)
        out = {}
        for name, proc in self.reference_unet.attn_processors.items():
            out[name] = proc.cache["hidden_states"]
        return out

    # ---- the loop ----
    @torch.no_grad()
    def denoise(self, **kw) -> torch.Tensor:
        """:meth:`_denoise` inside this pipeline's tuning scope (:meth:`set_tuning`)."""
        with ops.tuning_scope(**(getattr(self, "_tuning", None) or {})):
            return self._denoise(**kw)

    def _denoise(self, *, latents: torch.Tensor, prompt_embeds: torch.Tensor, negative_prompt_embeds: torch.Tensor,
                sa_hidden_states: Dict[str, torch.Tensor], num_inference_steps: int, guidance_scale: float,
                control: Optional[dict] = None, inpaint: Optional[dict] = None,
                callback: Optional[Callable] = None, callback_steps: int = 1, trace: Optional[list] = None,
                eta: float = 0.0, generator=None, variance_noise: Optional[List[torch.Tensor]] = None, t_start: int = 0) -> torch.Tensor:
        """latents [B, 4, h, w] fp32 -> final latents [B, 4, h, w] fp32.

        ``eta`` > 0 (DDIM only; other schedulers ignore it, like ``prepare_extra_step_kwargs``, IMAGDressing_v1_pipeline.py:102-119):
        the stochastic step, noise per step = ``variance_noise[i]`` [B, 4, h, w] or a draw of that shape in the UNet's element type
        from ``generator`` (what ``DDIMScheduler.step`` does with the reference's ``noise_pred``).
        ``t_start``: skip the first t_start timesteps of the schedule (inpainting ``strength`` < 1,
        ..._controlnet_inpainting.py:316-319 -> diffusers ``get_timesteps``); ``latents`` is then the noised image latent.

        ``control`` = dict(image=[1|B, 3, H, W] in [0,1] or NHWC8 bf16, prompt_embeds=[1,77,768],
        negative_prompt_embeds=[1,77,768], scale=float, keep=[float]*steps)
        ``inpaint`` = dict(mask=[B|1,1,h,w], image_latents=[B|1,4,h,w], noise=[B,4,h,w])
        """
        dev = self.device
        B, Cl, h, w = latents.shape
        HW = h * w
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps, device=dev)
        timesteps = [int(t) for t in sch.timesteps][int(t_start) * getattr(sch, "order", 1):]
        if not timesteps:
            raise ValueError(f"no denoising steps left (num_inference_steps={num_inference_steps}, t_start={t_start})")
        z = latents.to(device=dev, dtype=torch.float32).permute(0, 2, 3, 1).reshape(B, HW, Cl).contiguous()
        dt = self.unet.dtype
        x_in = torch.zeros(2 * B, h, w, 8, dtype=dt, device=dev)
        x_in[..., :Cl] = torch.cat([z, z]).view(2 * B, h, w, Cl)
        # rows [0,B): prompt (+garment), rows [B,2B): negative prompt, no garment -> ehs rows shared per half
        ehs = torch.cat([prompt_embeds[:1], negative_prompt_embeds[:1]]).to(device=dev, dtype=dt).contiguous()
        mask_rows = torch.cat([torch.ones(B), torch.zeros(B)]).to(device=dev, dtype=torch.float32)
        # sa_pair_layout: the mask above IS "garment on for rows [0, B), off for rows [B, 2B)" -- together with cfg_pair (identical latents in
        # the two halves) it lets the engine run the first hybrid block's self-attention phase once per image (unet.call_pair_half)
        cak = {"sa_hidden_states": sa_hidden_states, "sa_batch_mask": mask_rows, "sa_pair_layout": True}
        ctrl_img = ctrl_ehs = None
        if control is not None:
            img = control["image"]
            ctrl_img = img if (img.dim() == 4 and img.shape[-1] == 8 and img.dtype == dt) else nchw_to_nhwc8(img.to(dev), dt)
            ctrl_ehs = torch.cat([control["prompt_embeds"][:1], control["negative_prompt_embeds"][:1]]).to(device=dev, dtype=dt).contiguous()
        inp = None
        if inpaint is not None:
            def nhwc(t, c):
                t = t.to(device=dev, dtype=torch.float32)
                if t.shape[0] != B:
                    t = t.expand(B, -1, -1, -1)
                return t.permute(0, 2, 3, 1).reshape(B, HW, c).contiguous()
            inp = dict(mask=nhwc(inpaint["mask"], 1).view(B, HW).contiguous(), z_img=nhwc(inpaint["image_latents"], Cl),
                       noise=nhwc(inpaint["noise"], Cl))
        multistep = hasattr(sch, "step_guided")        # UniPC: latent updates are host-computed linear combinations
        if multistep and inp is not None:
            raise NotImplementedError("the inpainting blend is defined on the DDIM step (…inpainting.py:487-500)")
        keeps = None if control is None else [control.get("keep", [1.0] * len(timesteps))[i] for i in range(len(timesteps))]
        stochastic = float(eta) > 0.0 and not multistep
        if variance_noise is not None and stochastic and len(variance_noise) < len(timesteps):
            raise ValueError(f"variance_noise has {len(variance_noise)} entries for {len(timesteps)} steps")
        ctrl_scale = 0.0 if control is None else float(control.get("scale", 1.0))

        def ddim_step(t, i=None, coefs=None):
            """ControlNet + UNet + CFG / DDIM / blend / next UNet input for one timestep; ``t`` a Python int (eager) or a device
            scalar with ``coefs`` the device-side schedule coefficients (graph replay: nothing step-specific is baked in)."""
            down = mid = None
            if control is not None:
                down, mid = self.controlnet.forward_nhwc(x_in, t, ctrl_ehs, ctrl_img, ctrl_scale * (keeps[0] if i is None else keeps[i]))
            eps = self.unet.forward_nhwc(x_in, t, ehs, cak, down, mid, cfg_pair=True)     # x_in = [z; z]: both halves see the same latent
            kw = {}
            if inp is not None:
                kw = dict(mask=inp["mask"], z_img=inp["z_img"], noise=inp["noise"])
            if coefs is not None:
                ops.ddim_cfg_step(z, eps, x_in.view(2 * B, HW, 8), guidance=float(guidance_scale), coefs=coefs, **kw)
            else:
                if inp is not None:
                    kw["a_next"] = sch.alpha(timesteps[i + 1]) if i < len(timesteps) - 1 else None
                if stochastic:
                    vn = variance_noise[i] if variance_noise is not None else randn_tensor((B, Cl, h, w), generator=generator, device=dev, dtype=dt)
                    kw["var_noise"] = vn.to(device=dev, dtype=torch.float32).permute(0, 2, 3, 1).reshape(B, HW, Cl).contiguous()
                    kw["sigma"] = sch.sigma(timesteps[i], eta)
                ops.ddim_cfg_step(z, eps, x_in.view(2 * B, HW, 8), guidance=float(guidance_scale), a_t=sch.alpha(timesteps[i]),
                                  a_prev=sch.alpha_prev(timesteps[i]), **kw)

        # The time-embedding chain depends on the timestep only: one pass over the whole schedule here, every forward of the loop picks its row
        # (unet._Encoder.precompute_time_embeddings; 7 launches per UNet / ControlNet forward gone)
        encs = [m for m in [self.unet] + ([self.controlnet] if control is not None else []) if hasattr(m, "precompute_time_embeddings")]
        tables = [e.precompute_time_embeddings(timesteps, dev) for e in encs] if ops.TEMB_TABLE else []
        if not tables:
            encs = []

        def run_steps():
            nonlocal z
            use_graph = (getattr(self, "_step_graph", False) and not multistep and not stochastic and callback is None and trace is None
                         and len(timesteps) > 2 and (keeps is None or len(set(keeps)) == 1) and ops.ATTN_EVENT_HOOK is None)
            if use_graph:
                # HIP-graph replay of the denoising step (opt-in, ``enable_step_graph``): step 0 runs eagerly on the pipeline's side stream
                # (it also fills the step-invariant K / V caches of the processors), step 1 is CAPTURED (not executed) into a graph whose
                # only per-step inputs are two device scalars -- the timestep and the six schedule coefficients -- and the graph is then
                # replayed for steps 1 .. S-1: ~500 kernel launches per step become one hipGraphLaunch (the loop is host-bound at batch 1).
                steps_n = len(timesteps)
                t_table = torch.tensor(timesteps, dtype=torch.float32).to(dev)
                rows = []
                for i, t in enumerate(timesteps):
                    a_next = (sch.alpha(timesteps[i + 1]) if i < steps_n - 1 else None) if inp is not None else None
                    rows.append(ops.ddim_coefs(sch.alpha(t), sch.alpha_prev(t), a_next))
                coef_table = torch.tensor(rows, dtype=torch.float32).to(dev)
                t_dev = torch.zeros(1, dtype=torch.float32, device=dev)
                coef_dev = torch.zeros(6, dtype=torch.float32, device=dev)
                side = self.__dict__.get("_graph_stream")
                if side is None:
                    side = self._graph_stream = torch.cuda.Stream(device=dev)
                cur = torch.cuda.current_stream(dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    t_dev.copy_(t_table[0:1]); coef_dev.copy_(coef_table[0])
                    temb_bufs = [torch.empty_like(tab[0:1]) for tab in tables]     # fixed addresses inside the captured step, refreshed like t_dev
                    for e, buf, tab in zip(encs, temb_bufs, tables):
                        buf.copy_(tab[0:1])
                        e.use_time_embedding(buf)
                    ddim_step(t_dev, coefs=coef_dev)
                    g = torch.cuda.CUDAGraph()
                    g.capture_begin()
                    try:
                        ddim_step(t_dev, coefs=coef_dev)
                    finally:
                        g.capture_end()
                    for i in range(1, steps_n):
                        t_dev.copy_(t_table[i:i + 1]); coef_dev.copy_(coef_table[i])
                        for buf, tab in zip(temb_bufs, tables):
                            buf.copy_(tab[i:i + 1])
                        g.replay()
                cur.wait_stream(side)
                self._last_step_graph = g          # keep the executable graph alive until the next call (replays may still be in flight)
                return z.view(B, h, w, Cl).permute(0, 3, 1, 2).contiguous()
            for i, t in enumerate(timesteps):
                if multistep:
                    down = mid = None
                    if control is not None:
                        down, mid = self.controlnet.forward_nhwc(x_in, t, ctrl_ehs, ctrl_img, ctrl_scale * keeps[i])
                    eps = self.unet.forward_nhwc(x_in, t, ehs, cak, down, mid, cfg_pair=True)
                    z = sch.step_guided(eps.view(2 * B, HW, Cl), z, float(guidance_scale))
                    # emit the next 16-bit UNet input (both CFG halves) from z: the fused step with eps = 0, alpha = 1 is the identity on z
                    ops.ddim_cfg_step(z, ops.workspace("zero_eps", (2 * B, HW, Cl), torch.float32, dev), x_in.view(2 * B, HW, 8),
                                      guidance=1.0, a_t=1.0, a_prev=1.0)
                else:
                    ddim_step(t, i)
                if trace is not None:
                    trace.append(z.clone())
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, z.view(B, h, w, Cl).permute(0, 3, 1, 2))
            return z.view(B, h, w, Cl).permute(0, 3, 1, 2).contiguous()

        try:
            return run_steps()
        finally:
            for e in encs:
                e.clear_time_embeddings()

    # ---- shared front / back end ----
    def _cloth_tokens(self, ref_clip_image, ref_clip_hidden_states, device):
        """(cloth_proj_embed, cloth_null_embeds) [1,16,768] each (:409-415).  The null tokens are computed for
        API parity only -- the garment UNet's null half is discarded by the reference (:476-480)."""
        if ref_clip_hidden_states is None:
            ref_clip_hidden_states = self._clip_hidden(ref_clip_image, device)
        return self.ImgProj(ref_clip_hidden_states.to(device))

    def _ref_latents(self, ref_image, ref_image_latents):
        if ref_image_latents is not None:
            return ref_image_latents
        p = next(self.vae.parameters())
        return self.vae.encode(ref_image.to(dtype=p.dtype, device=p.device)).latent_dist.mean * 0.18215   # :457-458

    def _decode(self, latents, output_type, generator=None):
        if output_type == "latent":
            return StableDiffusionPipelineOutput(images=latents, nsfw_content_detected=None)
        p = next(self.vae.parameters())
        image = self.vae.decode((latents / self.vae.config.scaling_factor).to(p.dtype), return_dict=False)[0]
        image = (image.float() / 2 + 0.5).clamp(0, 1)
        if output_type == "pt":
            return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)
        arr = (image.permute(0, 2, 3, 1).cpu().numpy() * 255).round().astype("uint8")
        if output_type == "np":
            return StableDiffusionPipelineOutput(images=arr, nsfw_content_detected=None)
        from PIL import Image
        return StableDiffusionPipelineOutput(images=[Image.fromarray(a) for a in arr], nsfw_content_detected=None)

    def _shard(self, latents, shard: bool):
        """Data-parallel sharding of the image batch over torch.distributed ranks (no-op single process)."""
        from ... import dist as imd_dist
        return imd_dist.shard_rows(latents) if shard else latents

    def _sa_states(self, ref_latents, cloth_tokens, shard: bool):
        from ... import dist as imd_dist
        if shard and imd_dist.world_size() > 1:
            return imd_dist.garment_features_broadcast(self, ref_latents, cloth_tokens)
        return self.garment_features(ref_latents, cloth_tokens)


def controlnet_keep(num_steps: int, start: float, end: float) -> List[float]:
    """diffusers' per-step ControlNet gate (..._pipeline_ipa_controlnet.py:582-590)."""
    return [1.0 - float(i / num_steps < start or (i + 1) / num_steps > end) for i in range(num_steps)]


def first(x):
    return x[0] if isinstance(x, (list, tuple)) else x


def to_image_tensor(image, device, normalize: bool, size=None, multiple: int = 8) -> torch.Tensor:
    """PIL / ndarray / tensor -> [B, 3, H, W] float in [0, 1] (or [-1, 1] when ``normalize``).

    ``size`` = (height, width) requested by the caller: like diffusers' ``prepare_image`` /
    ``VaeImageProcessor.preprocess`` (call sites ..._controlnet.py:480-501, ..._inpainting.py:352-369) the image is resized
    to it, rounded DOWN to a multiple of ``multiple`` (the VAE scale factor) -- PIL inputs with PIL's Lanczos filter (the
    processor's default ``resample``), tensors / arrays with ``F.interpolate``'s default (nearest), as the library does.
    The pipelines then read the output size back from this tensor (:501), so it always divides by 8."""
    import numpy as np
    hw = None
    if size is not None:
        hw = (int(size[0]) // multiple * multiple, int(size[1]) // multiple * multiple)
    if isinstance(image, torch.Tensor):
        t = image.float()
        if t.dim() == 3:
            t = t.unsqueeze(0)
    else:
        if not isinstance(image, (list, tuple)):
            image = [image]
        ims = []
        for im in image:
            if hasattr(im, "convert"):
                im = im.convert("RGB")
                if hw is not None and (im.height, im.width) != hw:
                    from PIL import Image
                    im = im.resize((hw[1], hw[0]), resample=Image.LANCZOS)
            ims.append(np.asarray(im, dtype=np.float32) / 255.0)
        t = torch.from_numpy(np.stack(ims)).permute(0, 3, 1, 2)
        if normalize:
            t = t * 2.0 - 1.0
    if hw is not None and tuple(t.shape[-2:]) != hw:
        t = torch.nn.functional.interpolate(t, size=hw)
    return t.to(device)


def set_scale_by_type(unet, cls, **attrs):
    for proc in unet.attn_processors.values():
        if isinstance(proc, cls):
            for k, v in attrs.items():
                setattr(proc, k, v)


__all__ = ["controlnet_keep", "first", "to_image_tensor", "PipelineBase", "StableDiffusionPipelineOutput", "randn_tensor", "set_scale_by_type",
           "RefSAttnProcessor2_0", "LoraRefSAttnProcessor2_0", "LoRAIPAttnProcessor2_0", "IPAttnProcessor2_0"]
