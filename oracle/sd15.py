"""Plain-torch fp32 restatement of the diffusers==0.24.0 SD1.5 ``UNet2DConditionModel`` and
``ControlNetModel`` arithmetic (TEST INFRASTRUCTURE; **parity unpinned** -- diffusers is a
third-party dependency pinned at ``/root/reference/requirements.txt:12`` and is not vendored,
and the reference holds no tests or golden vectors at this boundary).

Anchors: the reference's call sites ``dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:466,
499,511`` (UNet), ``..._ipa_controlnet.py:651`` / ``..._controlnet_inpainting.py:425``
(ControlNet), and the published SD1.5 config (block_out_channels=(320,640,1280,1280),
layers_per_block=2, 8 heads, cross_attention_dim=768, GroupNorm(32), GEGLU FF,
flip_sin_to_cos=True, freq_shift=0, use_linear_projection=False).

Module and parameter names mirror diffusers so that ``state_dict()`` keys are the ones real
checkpoints carry (``ref_unet.*`` keys of ``IMAGDressing-v1_512.pt``,
inference_IMAGdressing.py:103-115) and so the reference's processors, which read
``attn.to_q / to_k / to_v / to_out / heads ...`` (attention_processor.py:545-625), can be
installed unchanged through ``set_attn_processor``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15 = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2, heads=8, cross_attention_dim=768, norm_num_groups=32,
    down_attn=(True, True, True, False), sample_size=64,
)


def timestep_embedding(t: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class DefaultAttnProcessor:
    """diffusers ``AttnProcessor2_0`` math (what a ControlNet / un-patched UNet runs)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        from .processors import sdpa
        ehs = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = attn.to_q(hidden_states), attn.to_k(ehs), attn.to_v(ehs)
        return attn.to_out[0](sdpa(q, k, v, attn.heads))


class Attention(nn.Module):
    """Attribute surface of diffusers ``Attention`` that processors touch."""

    def __init__(self, query_dim, cross_dim, heads):
        super().__init__()
        self.heads = heads
        kdim = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(kdim, query_dim, bias=False)
        self.to_v = nn.Linear(kdim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = DefaultAttnProcessor()

    def prepare_attention_mask(self, mask, *a, **k):
        return mask

    def set_processor(self, p):
        self.processor = p

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ehs, cak):
        # cross_attention_kwargs go to BOTH attn1 and attn2 (diffusers-0.24 BasicTransformerBlock)
        x = self.attn1(self.norm1(x), encoder_hidden_states=None, **cak) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=ehs, **cak) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, cross_dim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)

    def forward(self, x, ehs, cak):
        b, c, h, w = x.shape
        res = x
        y = self.proj_in(self.norm(x))
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            y = blk(y, ehs, cak)
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return self.proj_out(y) + res


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, groups, heads, cross_dim, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_ch, groups) for i in range(2)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(2)])
        self.has_attn = has_attn
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.add_down = add_down

    def forward(self, x, temb, ehs, cak):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs, cak)
            outs.append(x)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb_ch, groups, heads, cross_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, cross_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch, groups) for _ in range(2)])

    def forward(self, x, temb, ehs, cak):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ehs, cak)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, in_ch, prev_ch, out_ch, temb_ch, groups, heads, cross_dim, has_attn, add_up):
        super().__init__()
        rs = []
        for i in range(3):
            skip = in_ch if i == 2 else out_ch
            rin = prev_ch if i == 0 else out_ch
            rs.append(ResnetBlock2D(rin + skip, out_ch, temb_ch, groups))
        self.resnets = nn.ModuleList(rs)
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(out_ch, heads, cross_dim, groups) for _ in range(3)])
        self.has_attn = has_attn
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(out_ch)])
        self.add_up = add_up

    def forward(self, x, skips, temb, ehs, cak):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs, cak)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class _AttnProcMixin:
    @property
    def attn_processors(self) -> Dict[str, object]:
        procs: Dict[str, object] = {}

        def rec(name, mod):
            if isinstance(mod, Attention):
                procs[f"{name}.processor"] = mod.processor
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        for name, mod in self.named_children():
            rec(name, mod)
        return procs

    def set_attn_processor(self, processor):
        def rec(name, mod):
            if isinstance(mod, Attention):
                mod.set_processor(processor if not isinstance(processor, dict) else processor[f"{name}.processor"])
            for sub, child in mod.named_children():
                rec(f"{name}.{sub}", child)

        for name, mod in self.named_children():
            rec(name, mod)


class UNet2DConditionModel(nn.Module, _AttnProcMixin):
    def __init__(self, cfg: Optional[dict] = None):
        super().__init__()
        cfg = dict(SD15, **(cfg or {}))
        self.cfg = cfg
        boc = cfg["block_out_channels"]
        g, heads, cd = cfg["norm_num_groups"], cfg["heads"], cfg["cross_attention_dim"]
        temb_ch = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(boc[0], temb_ch), linear_2=nn.Linear(temb_ch, temb_ch)))
        # registration order down_blocks, up_blocks, mid_block == diffusers (decides
        # ``unet.attn_processors`` order and hence ``adapter_modules.{idx}`` checkpoint keys)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        out = boc[0]
        for i, ch in enumerate(boc):
            cin, out = out, ch
            self.down_blocks.append(DownBlock(cin, out, temb_ch, g, heads, cd, cfg["down_attn"][i], i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], temb_ch, g, heads, cd)
        rev = list(reversed(boc))
        up_attn = list(reversed(cfg["down_attn"]))
        out = rev[0]
        for i, ch in enumerate(rev):
            prev, out = out, ch
            inp = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(inp, prev, out, temb_ch, g, heads, cd, up_attn[i], i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def time_embed(self, t, batch):
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1).expand(batch)
        e = timestep_embedding(t, self.cfg["block_out_channels"][0])
        return self.time_embedding["linear_2"](F.silu(self.time_embedding["linear_1"](e)))

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None):
        cak = dict(cross_attention_kwargs or {})
        temb = self.time_embed(timestep, sample.shape[0])
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states, cak)
            skips += outs
        if down_block_additional_residuals is not None:          # ControlNet residual add
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
        x = self.mid_block(x, temb, encoder_hidden_states, cak)
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            x = blk(x, skips, temb, encoder_hidden_states, cak)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, out_ch, cond_ch=3, chans=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(cond_ch, chans[0], 3, padding=1)
        self.blocks = nn.ModuleList([])
        for i in range(len(chans) - 1):
            self.blocks.append(nn.Conv2d(chans[i], chans[i], 3, padding=1))
            self.blocks.append(nn.Conv2d(chans[i], chans[i + 1], 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(chans[-1], out_ch, 3, padding=1)

    def forward(self, c):
        e = F.silu(self.conv_in(c))
        for b in self.blocks:
            e = F.silu(b(e))
        return self.conv_out(e)


class ControlNetModel(nn.Module, _AttnProcMixin):
    """SD1.5 ControlNet: encoder copy + 13 zero-convs (call site ..._ipa_controlnet.py:651-659)."""

    def __init__(self, cfg: Optional[dict] = None):
        super().__init__()
        cfg = dict(SD15, **(cfg or {}))
        self.cfg = cfg
        boc = cfg["block_out_channels"]
        g, heads, cd = cfg["norm_num_groups"], cfg["heads"], cfg["cross_attention_dim"]
        temb_ch = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = nn.ModuleDict(dict(linear_1=nn.Linear(boc[0], temb_ch), linear_2=nn.Linear(temb_ch, temb_ch)))
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(boc[0])
        self.down_blocks = nn.ModuleList([])
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(boc[0], boc[0], 1)])
        out = boc[0]
        for i, ch in enumerate(boc):
            cin, out = out, ch
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock(cin, out, temb_ch, g, heads, cd, cfg["down_attn"][i], not last))
            for _ in range(2 + (0 if last else 1)):
                self.controlnet_down_blocks.append(nn.Conv2d(out, out, 1))
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)
        self.mid_block = MidBlock(boc[-1], temb_ch, g, heads, cd)

    time_embed = UNet2DConditionModel.time_embed

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0):
        temb = self.time_embed(timestep, sample.shape[0])
        x = self.conv_in(sample) + self.controlnet_cond_embedding(controlnet_cond)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, temb, encoder_hidden_states, {})
            skips += outs
        x = self.mid_block(x, temb, encoder_hidden_states, {})
        down = [zc(s) * conditioning_scale for s, zc in zip(skips, self.controlnet_down_blocks)]
        mid = self.controlnet_mid_block(x) * conditioning_scale
        return down, mid
