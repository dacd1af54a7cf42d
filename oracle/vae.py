"""TEST INFRASTRUCTURE -- fp32 CPU restatement of the SD1.5 VAE (`diffusers==0.24.0` `AutoencoderKL`), the module the
reference pipeline calls immediately before and after the denoising loop:

    /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:457-458   vae.encode(ref_image).latent_dist.mean * 0.18215
    /root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:544       vae.decode(latents / scaling_factor)[0]

diffusers is a pinned, un-vendored dependency (requirements.txt:12) that cannot be installed here, and the reference
holds no tests or golden vectors at this boundary: **parity unpinned**.  The restatement follows the published
architecture (config of `runwayml/stable-diffusion-v1-5/vae`: in/out 3, latent 4, block_out_channels (128, 256, 512, 512),
layers_per_block 2, norm_num_groups 32, SiLU, scaling_factor 0.18215) with diffusers' state-dict key names, and is
anchored on the 83,653,863-parameter count.  Pieces:

  ResnetBlock2D (temb-free, eps 1e-6): GN32 -> SiLU -> conv3x3 -> GN32 -> SiLU -> conv3x3, + (1x1 conv_shortcut) skip
  mid block: resnet, single-head (d = C) attention with GroupNorm, biased q/k/v/out projections and residual, resnet
  Encoder: conv_in, 4 down blocks (2 resnets each; stride-2 conv after F.pad(x, (0, 1, 0, 1)) on the first three),
           mid, GN + SiLU + conv_out (2 x latent channels), quant_conv 1x1 -> (mean, logvar)
  Decoder: post_quant_conv 1x1, conv_in, mid, 4 up blocks (3 resnets each; nearest-2x + conv3x3 after the first three),
           GN + SiLU + conv_out

Only tests/ (and the smoke / cpu_baseline helpers) may import this module."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                  layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)


class Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return h + (x if self.conv_shortcut is None else self.conv_shortcut(x))


class MidAttention(nn.Module):
    """diffusers Attention(heads = 1, dim_head = C, bias = True, residual_connection = True, upcast_softmax = True)."""

    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Identity()])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax((q @ k.transpose(1, 2)).float() * (C ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([MidAttention(ch, groups)])
        self.resnets = nn.ModuleList([Resnet(ch, ch, groups), Resnet(ch, ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Conv(nn.Module):       # holder for the `...samplers.0.conv` key level
    def __init__(self, ch, stride):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=0 if stride == 2 else 1)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Conv(cout, 2)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))
        return x


class UpBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, up):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Conv(cout, 1)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return x


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([DownBlock(boc[max(i - 1, 0)], boc[i], cfg["layers_per_block"], g, i < len(boc) - 1)
                                          for i in range(len(boc))])
        self.mid_block = Mid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rev[0], 3, padding=1)
        self.mid_block = Mid(rev[0], g)
        self.up_blocks = nn.ModuleList([UpBlock(rev[max(i - 1, 0)], rev[i], cfg["layers_per_block"] + 1, g, i < len(boc) - 1)
                                        for i in range(len(boc))])
        self.conv_norm_out = nn.GroupNorm(g, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = dict(VAE_CONFIG, **(cfg or {}))
        lc = self.cfg["latent_channels"]
        self.encoder = Encoder(self.cfg)
        self.decoder = Decoder(self.cfg)
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)

    @torch.no_grad()
    def encode_moments(self, x):
        """-> (mean, logvar) of the diagonal Gaussian posterior; `latent_dist.mean` of the reference call is `mean`."""
        m = self.quant_conv(self.encoder(x))
        mean, logvar = m.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def n_params(m):
    return sum(p.numel() for p in m.parameters())


def seeded_state_dict(cfg=None, seed=0):
    """fan-in scaled synthetic weights (activations stay O(1)); norm weights ~ 1"""
    m = AutoencoderKL(cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith(".bias"):
            t = torch.randn(v.shape, generator=g) * 0.02
        elif "norm" in k:
            t = 1.0 + torch.randn(v.shape, generator=g) * 0.05
        else:
            fan_in = math.prod(v.shape[1:])
            t = torch.randn(v.shape, generator=g) / math.sqrt(fan_in)
        sd[k] = t
    return sd
