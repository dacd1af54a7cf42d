"""CLIP text and vision encoders on the HIP kernels -- the step immediately before the denoising loop
(SURVEY.md section 8f, rank 2).

Replace the `transformers` modules the reference builds at /root/reference/inference_IMAGdressing.py:44-47
(`CLIPTextModel` of SD1.5 = CLIP ViT-L/14 text tower; `CLIPVisionModelWithProjection` of h94/IP-Adapter = OpenCLIP ViT-H/14)
for exactly the calls the pipelines make:

    text_encoder(input_ids)[0]                                              IMAGDressing_v1_pipeline.py:246-262 (encode_prompt)
    image_encoder(pixels, output_hidden_states=True).hidden_states[-2]      IMAGDressing_v1_pipeline.py:404-411
    image_encoder(pixels).image_embeds                                      (IP-Adapter FaceID-Plus shortcut path)
    .config.hidden_size / .config.projection_dim, .dtype, .device

Built from `transformers`-layout state dicts (keys with or without the `text_model.` / `vision_model.` prefixes).
Pre-LN transformer blocks: `imd_layernorm` -> fused QKV projection with the head-split epilogue (bias, softmax scale folded
into Q) -> `imd_attention` (d = 64 / 80; causal mask for the text tower) -> out projection + residual -> `imd_layernorm` ->
fc1 + quick-GELU / GELU -> fc2 + residual.  Token / position embedding lookup and the ViT sequence assembly are small HIP
kernels (`imd_embed_tokens`, `imd_vit_assemble`); the 14x14 stride-14 patch convolution is a GEMM over unfolded patches.
"""
from __future__ import annotations

import math
import types
from typing import Dict, Optional

import torch

from . import ops
from .hub import PretrainedMixin
from .unet import LinearOp, NormParams

bf16 = torch.bfloat16
LOG2E = 1.4426950408889634

TEXT_CONFIG = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=49407, projection_dim=768)
VISION_CONFIG = dict(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                     patch_size=14, num_channels=3, hidden_act="gelu", layer_norm_eps=1e-5, projection_dim=1024)
_ACT = {"quick_gelu": ops.ACT_QUICK_GELU, "gelu": ops.ACT_GELU}


def _strip(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


class _Layer:
    def __init__(self, sd, p, heads, act, eps, device, dtype):
        cat = lambda n: torch.cat([sd[f"{p}.self_attn.{x}_proj.{n}"] for x in ("q", "k", "v")])      # noqa: E731
        self.wqkv = cat("weight").detach().to(device=device, dtype=dtype).contiguous()
        self.bqkv = cat("bias").detach().to(device=device, dtype=torch.float32).contiguous()
        self.out = LinearOp(sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"], device, dtype)
        self.ln1, self.ln2 = NormParams(sd, f"{p}.layer_norm1", device), NormParams(sd, f"{p}.layer_norm2", device)
        self.fc1 = LinearOp(sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"], device, dtype)
        self.fc2 = LinearOp(sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"], device, dtype)
        self.heads, self.act, self.eps = heads, act, eps

    def __call__(self, h: torch.Tensor, causal: bool) -> torch.Tensor:
        B, N, Cc = h.shape
        H = self.heads
        D = Cc // H
        dt, dev = h.dtype, h.device
        dpk, dpv = ops.attn_padded_dims(D)
        LP = ops.pad64(N)
        x = ops.layer_norm(h.view(B * N, Cc), self.ln1.weight, self.ln1.bias, eps=self.eps)
        q = ops.workspace("clip_q", (B, H, N, dpk), dt, dev)
        k = ops.workspace("clip_k", (B, H, N, dpk), dt, dev)
        vt = ops.workspace("clip_vt", (B, H, dpv, LP), dt, dev)          # zero-initialised: key padding stays 0
        ops.conv_gemm(x, self.wqkv, M=B * N, N=3 * Cc, Cin=Cc, Hin=N, Win=1, Hout=N, Wout=1, bias=self.bqkv,
                      heads=dict(C=Cc, H=H, D=D, dests=[(q, 0, dpk, N, D ** -0.5 * LOG2E), (k, 0, dpk, N, 1.0), (vt, 1, dpv, LP, 1.0)]))
        o = torch.empty(B, N, Cc, dtype=dt, device=dev)
        ops.attention(q, k, vt, o, B=B, H=H, N=N, D=D, L1=N, L1P=LP, causal=causal)
        h = self.out(o.view(B * N, Cc), res=h.view(B * N, Cc))
        x = ops.layer_norm(h, self.ln2.weight, self.ln2.bias, eps=self.eps)
        x = self.fc1(x, act=self.act)
        return self.fc2(x, res=h).view(B, N, Cc)


class CLIPTextModel(PretrainedMixin):
    """`transformers.CLIPTextModel` surface used by the pipelines: `model(input_ids)[0]` = last_hidden_state [B, T, C]."""

    _config_keys = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                    "max_position_embeddings", "hidden_act", "layer_norm_eps")

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None, device="cuda", dtype=bf16):
        device = torch.device(device)
        ops.ensure_device(device)
        cfg = dict(TEXT_CONFIG, **(config or {}))
        sd = _strip(state_dict, "text_model.")
        self.cfg, self.config = cfg, types.SimpleNamespace(**cfg)
        self._ctor_config = config
        self.device, self.dtype = device, dtype
        self.tok = sd["embeddings.token_embedding.weight"].detach().to(device=device, dtype=dtype).contiguous()
        self.pos = sd["embeddings.position_embedding.weight"].detach().to(device=device, dtype=dtype).contiguous()
        act = _ACT[cfg["hidden_act"]]
        self.layers = [_Layer(sd, f"encoder.layers.{i}", cfg["num_attention_heads"], act, cfg["layer_norm_eps"], device, dtype)
                       for i in range(cfg["num_hidden_layers"])]
        self.final_ln = NormParams(sd, "final_layer_norm", device)

    def parameters(self):
        yield self.tok

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, attention_mask=None, output_hidden_states: bool = False, **unused):
        if attention_mask is not None:
            raise NotImplementedError("CLIPTextModel: padding masks are not used by the reference (config.use_attention_mask is unset)")
        ops.ensure_device(input_ids.device)
        ids = input_ids.to(torch.int64).contiguous()
        B, T = ids.shape
        h = ops.embed_tokens(self.tok, self.pos, ids)
        hs = [h]
        for lyr in self.layers:
            h = lyr(h, causal=True)
            hs.append(h)
        Cc = h.shape[-1]
        last = ops.layer_norm(h.view(B * T, Cc), self.final_ln.weight, self.final_ln.bias, eps=self.cfg["layer_norm_eps"]).view(B, T, Cc)
        eos = self.cfg.get("eos_token_id", 2)
        idx = ids.argmax(dim=-1) if eos == 2 else (ids == eos).int().argmax(dim=-1)
        pooled = last[torch.arange(B, device=last.device), idx]
        out = _Output(last_hidden_state=last, pooler_output=pooled, hidden_states=tuple(hs) if output_hidden_states else None)
        return out


class CLIPVisionModelWithProjection(PretrainedMixin):
    """`transformers.CLIPVisionModelWithProjection` surface used by the pipelines: `.hidden_states[-2]`, `.image_embeds`."""

    _config_keys = ("hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "image_size",
                    "patch_size", "projection_dim", "hidden_act", "layer_norm_eps")

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None, device="cuda", dtype=bf16):
        device = torch.device(device)
        ops.ensure_device(device)
        cfg = dict(VISION_CONFIG, **(config or {}))
        proj_w = state_dict["visual_projection.weight"]
        sd = _strip(state_dict, "vision_model.")
        self.cfg, self.config = cfg, types.SimpleNamespace(**cfg)
        self._ctor_config = config
        self.device, self.dtype = device, dtype
        Cc, ps, nc = cfg["hidden_size"], cfg["patch_size"], cfg["num_channels"]
        self.kdim = nc * ps * ps
        self.kpad = (self.kdim + 7) // 8 * 8
        w = sd["embeddings.patch_embedding.weight"].detach().float().reshape(Cc, self.kdim)      # [C, c*ps*ps]: (c, ky, kx) order
        w = torch.nn.functional.pad(w, (0, self.kpad - self.kdim))
        self.patch = LinearOp(w, None, device, dtype)
        self.cls = sd["embeddings.class_embedding"].detach().to(device=device, dtype=dtype).contiguous()
        self.pos = sd["embeddings.position_embedding.weight"].detach().to(device=device, dtype=dtype).contiguous()
        self.pre_ln = NormParams(sd, "pre_layrnorm", device)
        self.post_ln = NormParams(sd, "post_layernorm", device)
        act = _ACT[cfg["hidden_act"]]
        self.layers = [_Layer(sd, f"encoder.layers.{i}", cfg["num_attention_heads"], act, cfg["layer_norm_eps"], device, dtype)
                       for i in range(cfg["num_hidden_layers"])]
        self.proj = LinearOp(proj_w, None, device, dtype)

    def parameters(self):
        yield self.cls

    @torch.no_grad()
    def __call__(self, pixel_values: torch.Tensor, output_hidden_states: bool = False, **unused):
        ops.ensure_device(pixel_values.device)
        cfg = self.cfg
        B, nc, Hh, Ww = pixel_values.shape
        ps, Cc = cfg["patch_size"], cfg["hidden_size"]
        gh, gw = Hh // ps, Ww // ps
        P = gh * gw
        if P + 1 != self.pos.shape[0]:
            raise ValueError(f"CLIPVisionModel: {Hh}x{Ww} pixels give {P} patches, the position table has {self.pos.shape[0] - 1}")
        # unfold the non-overlapping patches (pure data movement): [B, c, gh, ps, gw, ps] -> [B*P, c*ps*ps], K padded to a multiple of 8
        x = pixel_values.to(self.dtype).view(B, nc, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * P, self.kdim)
        xp = torch.zeros(B * P, self.kpad, dtype=self.dtype, device=x.device)
        xp[:, :self.kdim] = x
        patches = self.patch(xp).view(B, P, Cc)
        h = ops.vit_assemble(patches, self.cls, self.pos)
        N = P + 1
        h = ops.layer_norm(h.view(B * N, Cc), self.pre_ln.weight, self.pre_ln.bias, eps=cfg["layer_norm_eps"]).view(B, N, Cc)
        hs = [h]
        for lyr in self.layers:
            h = lyr(h, causal=False)
            hs.append(h)
        pooled = ops.layer_norm(h[:, 0].contiguous(), self.post_ln.weight, self.post_ln.bias, eps=cfg["layer_norm_eps"])
        embeds = self.proj(pooled)
        return _Output(image_embeds=embeds, last_hidden_state=h, hidden_states=tuple(hs) if output_hidden_states else None)


class _Output(dict):
    """attribute + index access like a transformers ModelOutput (`out[0]`, `out.hidden_states`)"""

    def __init__(self, **kw):
        super().__init__({k: v for k, v in kw.items() if v is not None})
        self.__dict__.update(kw)

    def __getitem__(self, i):
        if isinstance(i, int):
            return list(self.values())[i]
        return super().__getitem__(i)
