"""SD1.5 VAE (`AutoencoderKL`) on the HIP kernels -- the step immediately before and after the denoising loop
(SURVEY.md section 8f, rank 1).

Replaces the un-vendored `diffusers==0.24.0` `AutoencoderKL` the reference constructs at
/root/reference/inference_IMAGdressing.py:47-48 and calls at
/root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:457-458 (`vae.encode(x).latent_dist.mean`) and :544
(`vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]`).  Same surface for those calls:
`.config.scaling_factor`, `.config.block_out_channels`, `.encode(x).latent_dist.{mean, mode(), sample()}`,
`.decode(z, return_dict=False)`, `.parameters()`, `.dtype`, `.device`, `enable_slicing()` / `disable_slicing()`.
Built from a diffusers-layout state dict (keys unchanged).

Everything runs through the hot path's own kernels: NHWC 16-bit activations, `imd_conv_gemm` (halo-patch 3x3 conv,
fused nearest-2x upsample, bottom/right-only padding for the encoder's stride-2 convs, bias / residual epilogues),
`imd_groupnorm` (+SiLU).  The mid block's single-head d = 512 attention is two GEMMs around `imd_softmax_rows`
(fp32 scores, the reference's upcast softmax): S = Q K^T / sqrt(C), P = softmax(S), O = P V with V^T written by the
projection's head-split epilogue.  No fallback: CPU tensors raise.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import torch

from . import ops
from .hub import PretrainedMixin
from .unet import ConvOp, LinearOp, NormParams, nchw_to_nhwc8

bf16 = torch.bfloat16

VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                  layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)


def vae_param_shapes(cfg: Optional[dict] = None) -> Dict[str, tuple]:
    """name -> shape of every tensor of the diffusers AutoencoderKL state dict (83,653,863 parameters at the SD1.5 config)."""
    c = dict(VAE_CONFIG, **(cfg or {}))
    boc, lc, layers = c["block_out_channels"], c["latent_channels"], c["layers_per_block"]
    sh: Dict[str, tuple] = {}

    def conv(p, cin, cout, k):
        sh[f"{p}.weight"] = (cout, cin, k, k); sh[f"{p}.bias"] = (cout,)

    def norm(p, ch):
        sh[f"{p}.weight"] = (ch,); sh[f"{p}.bias"] = (ch,)

    def resnet(p, cin, cout):
        norm(f"{p}.norm1", cin); conv(f"{p}.conv1", cin, cout, 3); norm(f"{p}.norm2", cout); conv(f"{p}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{p}.conv_shortcut", cin, cout, 1)

    def mid(p, ch):
        norm(f"{p}.attentions.0.group_norm", ch)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            sh[f"{p}.attentions.0.{n}.weight"] = (ch, ch); sh[f"{p}.attentions.0.{n}.bias"] = (ch,)
        resnet(f"{p}.resnets.0", ch, ch); resnet(f"{p}.resnets.1", ch, ch)

    conv("encoder.conv_in", c["in_channels"], boc[0], 3)
    for i, ch in enumerate(boc):
        for j in range(layers):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", boc[max(i - 1, 0)] if j == 0 else ch, ch)
        if i < len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1]); conv("encoder.conv_out", boc[-1], 2 * lc, 3)
    rev = list(reversed(boc))
    conv("decoder.conv_in", lc, rev[0], 3)
    mid("decoder.mid_block", rev[0])
    for i, ch in enumerate(rev):
        for j in range(layers + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", rev[max(i - 1, 0)] if j == 0 else ch, ch)
        if i < len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    norm("decoder.conv_norm_out", rev[-1]); conv("decoder.conv_out", rev[-1], c["out_channels"], 3)
    conv("quant_conv", 2 * lc, 2 * lc, 1); conv("post_quant_conv", lc, lc, 1)
    return sh


def _pad_out4(w: torch.Tensor, b: torch.Tensor):
    """pad a conv's output channels to a multiple of 4 with zero filters (the GEMM's N granularity)"""
    cout = w.shape[0]
    cp = (cout + 3) // 4 * 4
    if cp == cout:
        return w, b
    return (torch.cat([w, w.new_zeros((cp - cout,) + tuple(w.shape[1:]))]), torch.cat([b, b.new_zeros(cp - cout)]))


class _Resnet:
    def __init__(self, sd, p, groups, device, dtype):
        self.norm1, self.norm2 = NormParams(sd, f"{p}.norm1", device), NormParams(sd, f"{p}.norm2", device)
        self.conv1 = ConvOp(sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], device, dtype)
        self.conv2 = ConvOp(sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], device, dtype)
        self.shortcut = None
        if f"{p}.conv_shortcut.weight" in sd:
            self.shortcut = ConvOp(sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"], device, dtype)
        self.groups = groups

    def __call__(self, x):
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, groups=self.groups, eps=1e-6, silu=True)
        h = self.conv1(h)
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, groups=self.groups, eps=1e-6, silu=True)
        return self.conv2(h, res=x if self.shortcut is None else self.shortcut(x))


class _MidAttention:
    """Single head, d = C: GroupNorm -> q / k / v projections (biased) -> softmax(q k^T / sqrt(C)) v -> out proj + residual."""

    def __init__(self, sd, p, ch, groups, device, dtype):
        self.norm = NormParams(sd, f"{p}.group_norm", device)
        self.to_q = LinearOp(sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"], device, dtype)
        self.to_k = LinearOp(sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"], device, dtype)
        self.to_v = LinearOp(sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"], device, dtype)
        self.to_out = LinearOp(sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"], device, dtype)
        self.ch, self.groups = ch, groups

    def __call__(self, x):
        B, H, W, Cc = x.shape
        N = H * W
        dt = x.dtype
        h = ops.group_norm(x, self.norm.weight, self.norm.bias, groups=self.groups, eps=1e-6, silu=False).view(B * N, Cc)
        q = self.to_q(h)
        k = self.to_k(h)
        NP = ops.pad64(N)
        vt = torch.zeros((B, 1, Cc, NP), dtype=dt, device=x.device) if NP != N else torch.empty((B, 1, Cc, NP), dtype=dt, device=x.device)
        # V^T [B, 1, C, N] straight from the projection's epilogue (keys contiguous: it is the [N][K] operand of O = P V)
        ops.conv_gemm(h, self.to_v.weight, M=B * N, N=Cc, Cin=Cc, Hout=N, Wout=1, Hin=N, Win=1, bias=self.to_v.bias,
                      heads=dict(C=Cc, H=1, D=Cc, dests=[(vt, 1, Cc, NP, 1.0)]))
        o = torch.empty((B * N, Cc), dtype=dt, device=x.device)
        scores = ops.workspace("vae_scores", (N, N), torch.float32, x.device)
        probs = ops.workspace("vae_probs", (N, NP), dt, x.device)       # zero-initialised: columns N..NP stay 0
        for b in range(B):
            qb, kb = q[b * N:(b + 1) * N], k[b * N:(b + 1) * N]
            ops.conv_gemm(qb, kb, M=N, N=N, Cin=Cc, out=scores, out_f32=True, out_scale=Cc ** -0.5)      # S = q k^T / sqrt(C)
            ops.softmax_rows(scores, out=probs)
            ops.conv_gemm(probs, vt[b, 0], M=N, N=Cc, Cin=NP, out=o[b * N:(b + 1) * N])                   # O = P V
        return self.to_out(o, res=x.view(B * N, Cc)).view(B, H, W, Cc)


class _Mid:
    def __init__(self, sd, p, ch, groups, device, dtype):
        self.r0 = _Resnet(sd, f"{p}.resnets.0", groups, device, dtype)
        self.attn = _MidAttention(sd, f"{p}.attentions.0", ch, groups, device, dtype)
        self.r1 = _Resnet(sd, f"{p}.resnets.1", groups, device, dtype)

    def __call__(self, x):
        return self.r1(self.attn(self.r0(x)))


class DiagonalGaussian:
    """`latent_dist` of `encode()`: mean / logvar in NCHW fp32 (diffusers DiagonalGaussianDistribution surface)."""

    def __init__(self, mean, logvar):
        self.mean, self.logvar = mean, logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device if generator is None else generator.device,
                            dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise


class AutoencoderKL(PretrainedMixin):
    _config_keys = ("block_out_channels", "layers_per_block", "latent_channels", "norm_num_groups", "scaling_factor")

    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None, device="cuda", dtype=bf16):
        device = torch.device(device)
        ops.ensure_device(device)
        cfg = dict(VAE_CONFIG, **(config or {}))
        self.cfg = cfg
        self._ctor_config = config
        self.config = types.SimpleNamespace(**cfg)
        self.device, self.dtype = device, dtype
        sd = state_dict
        boc, g, layers = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["layers_per_block"]
        self.lc = cfg["latent_channels"]

        def conv(p, pad4=False):
            w, b = sd[f"{p}.weight"], sd[f"{p}.bias"]
            if pad4:
                w, b = _pad_out4(w, b)
            return ConvOp(w, b, device, dtype)
        # encoder
        self.e_conv_in = conv("encoder.conv_in")
        self.e_down = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", g, device, dtype) for j in range(layers)]
            ds = conv(f"encoder.down_blocks.{i}.downsamplers.0.conv") if i < len(boc) - 1 else None
            self.e_down.append((res, ds))
        self.e_mid = _Mid(sd, "encoder.mid_block", boc[-1], g, device, dtype)
        self.e_norm_out = NormParams(sd, "encoder.conv_norm_out", device)
        self.e_conv_out = conv("encoder.conv_out")
        self.quant_conv = conv("quant_conv")
        # decoder
        self.post_quant_conv = conv("post_quant_conv")
        self.d_conv_in = conv("decoder.conv_in")
        self.d_mid = _Mid(sd, "decoder.mid_block", boc[-1], g, device, dtype)
        self.d_up = []
        for i in range(len(boc)):
            res = [_Resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", g, device, dtype) for j in range(layers + 1)]
            us = conv(f"decoder.up_blocks.{i}.upsamplers.0.conv") if i < len(boc) - 1 else None
            self.d_up.append((res, us))
        self.d_norm_out = NormParams(sd, "decoder.conv_norm_out", device)
        self.d_conv_out = conv("decoder.conv_out", pad4=True)
        self.groups = g
        self.use_slicing = False
        self._probe = torch.empty(1, dtype=dtype, device=device)

    @classmethod
    def random_init(cls, seed: int = 5, config: Optional[dict] = None, device="cuda", dtype=bf16):
        """seeded synthetic weights (fan-in scaled, norm weights ~ 1): there are no checkpoints in this environment"""
        g = torch.Generator(device=device).manual_seed(seed)
        sd = {}
        for name, shape in vae_param_shapes(config).items():
            r = torch.randn(shape, generator=g, device=device)
            if name.endswith(".bias"):
                sd[name] = r * 0.02
            elif "norm" in name:
                sd[name] = 1.0 + r * 0.05
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                sd[name] = r * fan_in ** -0.5
        return cls(sd, config, device, dtype)

    # ---- diffusers surface -------------------------------------------------------------------------------------
    def parameters(self):
        yield self._probe

    def enable_slicing(self):
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    # ---- compute ------------------------------------------------------------------------------------------------
    def encode_nhwc(self, x8: torch.Tensor) -> torch.Tensor:
        """x8 [B, H, W, 8] 16-bit (channels 3..7 zero) -> moments [B, H/8, W/8, 2*latent] 16-bit"""
        x = self.e_conv_in(x8)
        for res, ds in self.e_down:
            for r in res:
                x = r(x)
            if ds is not None:
                x = ds(x, stride=2, pad_br_only=True)
        x = self.e_mid(x)
        x = ops.group_norm(x, self.e_norm_out.weight, self.e_norm_out.bias, groups=self.groups, eps=1e-6, silu=True)
        return self.quant_conv(self.e_conv_out(x))

    def decode_nhwc(self, z8: torch.Tensor) -> torch.Tensor:
        """z8 [B, h, w, 8] 16-bit (channels latent..7 zero) -> image [B, 8h, 8w, 4] 16-bit (channel 3 is padding)"""
        x = self.d_conv_in(nhwc_pad8(self.post_quant_conv(z8)))
        x = self.d_mid(x)
        for res, us in self.d_up:
            for r in res:
                x = r(x)
            if us is not None:
                x = us(x, ups=True)
        x = ops.group_norm(x, self.d_norm_out.weight, self.d_norm_out.bias, groups=self.groups, eps=1e-6, silu=True)
        return self.d_conv_out(x)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [B, 3, H, W] in [-1, 1] (any float dtype, on the GPU) -> object with `.latent_dist`"""
        ops.ensure_device(x.device)
        outs = []
        for xb in (x.split(1) if self.use_slicing and x.shape[0] > 1 else (x,)):
            m = self.encode_nhwc(nchw_to_nhwc8(xb.float(), self.dtype))
            outs.append(m.float().permute(0, 3, 1, 2).contiguous())
        m = torch.cat(outs)
        dist = DiagonalGaussian(m[:, :self.lc], m[:, self.lc:2 * self.lc])
        return types.SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        """z [B, latent, h, w] (already divided by scaling_factor) -> image [B, 3, 8h, 8w] in the model dtype"""
        ops.ensure_device(z.device)
        outs = []
        for zb in (z.split(1) if self.use_slicing and z.shape[0] > 1 else (z,)):
            y = self.decode_nhwc(nchw_to_nhwc8(zb.float(), self.dtype))
            outs.append(y[..., :self.cfg["out_channels"]].permute(0, 3, 1, 2).contiguous())
        img = torch.cat(outs)
        return types.SimpleNamespace(sample=img) if return_dict else (img,)


def nhwc_pad8(x: torch.Tensor) -> torch.Tensor:
    """pad the channel dim of an NHWC tensor to a multiple of 8 with zeros (the GEMM's K granularity)"""
    c = x.shape[-1]
    if c % 8 == 0:
        return x
    out = torch.zeros(x.shape[:-1] + ((c + 7) // 8 * 8,), dtype=x.dtype, device=x.device)
    out[..., :c] = x
    return out
