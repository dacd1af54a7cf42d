"""SD1.5 ``UNet2DConditionModel`` / ``ControlNetModel`` execution engine on the HIP kernels.

The reference calls the un-vendored diffusers==0.24.0 classes of the same names
(/root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:466,499,511;
..._pipeline_ipa_controlnet.py:651-659).  This engine keeps their *surface* -- constructor from a
diffusers-layout state dict, ``.config``, ``.attn_processors`` / ``.set_attn_processor`` with the same
processor names and order (down, up, mid: it decides the ``adapter_modules.{idx}`` checkpoint keys,
inference_IMAGdressing.py:86,117), ``forward(sample, timestep, encoder_hidden_states,
cross_attention_kwargs, down_block_additional_residuals, mid_block_additional_residual)`` -- and
re-implements the body MI355X-first:

* activations live in HBM as NHWC / token-major bf16, so a feature map IS the [B, HW, C] token
  matrix of its transformer block (no layout shuffles between conv and attention);
* every conv / linear is the implicit-GEMM MFMA kernel with its bias, time-embedding add,
  residual add, SiLU, GEGLU or head-split fused in the epilogue;
* nearest-2x upsampling is folded into the next conv's gather; the 22 per-resnet
  ``time_emb_proj`` linears run as ONE GEMM per forward;
* attention layers call ``attn.processor(attn, hidden_states, ...)`` exactly like diffusers does --
  the processors in ``imagdressing_amd.adapter.attention_processor`` are the plugin surface.

Nothing here falls back to torch ops for the math: tensors are only storage handles.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

from . import ops
from .hub import PretrainedMixin

FUSED_LN_CHANNELS = (320, 640, 1280)     # csrc/row_linear.hip, row_linear_k640.hip: channel counts the fused LayerNorm -> linear kernels exist for

bf16 = torch.bfloat16

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, sample_size=64,
    down_attn=(True, True, True, False), time_cond_proj_dim=None,
)


# ---------------------------------------------------------------------------------------------
# parameter inventory (diffusers key layout) -- used for random init and for validating checkpoints
# ---------------------------------------------------------------------------------------------
def _resnet_shapes(p, cin, cout, temb):
    s = {f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,),
         f"{p}.conv1.weight": (cout, cin, 3, 3), f"{p}.conv1.bias": (cout,),
         f"{p}.time_emb_proj.weight": (cout, temb), f"{p}.time_emb_proj.bias": (cout,),
         f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,),
         f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,)}
    if cin != cout:
        s[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        s[f"{p}.conv_shortcut.bias"] = (cout,)
    return s


def _transformer_shapes(p, ch, cross):
    s = {f"{p}.norm.weight": (ch,), f"{p}.norm.bias": (ch,),
         f"{p}.proj_in.weight": (ch, ch, 1, 1), f"{p}.proj_in.bias": (ch,),
         f"{p}.proj_out.weight": (ch, ch, 1, 1), f"{p}.proj_out.bias": (ch,)}
    b = f"{p}.transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        s[f"{b}.{n}.weight"] = (ch,); s[f"{b}.{n}.bias"] = (ch,)
    for a, kd in (("attn1", ch), ("attn2", cross)):
        s[f"{b}.{a}.to_q.weight"] = (ch, ch)
        s[f"{b}.{a}.to_k.weight"] = (ch, kd)
        s[f"{b}.{a}.to_v.weight"] = (ch, kd)
        s[f"{b}.{a}.to_out.0.weight"] = (ch, ch); s[f"{b}.{a}.to_out.0.bias"] = (ch,)
    s[f"{b}.ff.net.0.proj.weight"] = (8 * ch, ch); s[f"{b}.ff.net.0.proj.bias"] = (8 * ch,)
    s[f"{b}.ff.net.2.weight"] = (ch, 4 * ch); s[f"{b}.ff.net.2.bias"] = (ch,)
    return s


def _encoder_shapes(cfg):
    boc = cfg["block_out_channels"]
    temb, cross = boc[0] * 4, cfg["cross_attention_dim"]
    s = {"conv_in.weight": (boc[0], cfg["in_channels"], 3, 3), "conv_in.bias": (boc[0],),
         "time_embedding.linear_1.weight": (temb, boc[0]), "time_embedding.linear_1.bias": (temb,),
         "time_embedding.linear_2.weight": (temb, temb), "time_embedding.linear_2.bias": (temb,)}
    out = boc[0]
    for i, ch in enumerate(boc):
        cin, out = out, ch
        for j in range(2):
            s.update(_resnet_shapes(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb))
            if cfg["down_attn"][i]:
                s.update(_transformer_shapes(f"down_blocks.{i}.attentions.{j}", out, cross))
        if i != len(boc) - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out, out, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out,)
    s.update(_resnet_shapes("mid_block.resnets.0", boc[-1], boc[-1], temb))
    s.update(_resnet_shapes("mid_block.resnets.1", boc[-1], boc[-1], temb))
    s.update(_transformer_shapes("mid_block.attentions.0", boc[-1], cross))
    return s


def unet_param_shapes(cfg=None) -> Dict[str, tuple]:
    cfg = dict(SD15_CONFIG, **(cfg or {}))
    boc = cfg["block_out_channels"]
    temb, cross = boc[0] * 4, cfg["cross_attention_dim"]
    s = _encoder_shapes(cfg)
    rev = list(reversed(boc))
    up_attn = list(reversed(cfg["down_attn"]))
    out = rev[0]
    for i, ch in enumerate(rev):
        prev, out = out, ch
        inp = rev[min(i + 1, len(boc) - 1)]
        for j in range(3):
            skip = inp if j == 2 else out
            rin = prev if j == 0 else out
            s.update(_resnet_shapes(f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb))
            if up_attn[i]:
                s.update(_transformer_shapes(f"up_blocks.{i}.attentions.{j}", out, cross))
        if i != len(boc) - 1:
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out, out, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out,)
    s["conv_norm_out.weight"] = (boc[0],); s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg["out_channels"], boc[0], 3, 3); s["conv_out.bias"] = (cfg["out_channels"],)
    return s


def controlnet_param_shapes(cfg=None, cond_channels=3, embed_channels=(16, 32, 96, 256)) -> Dict[str, tuple]:
    cfg = dict(SD15_CONFIG, **(cfg or {}))
    boc = cfg["block_out_channels"]
    s = _encoder_shapes(cfg)
    e = "controlnet_cond_embedding"
    s[f"{e}.conv_in.weight"] = (embed_channels[0], cond_channels, 3, 3); s[f"{e}.conv_in.bias"] = (embed_channels[0],)
    for i in range(len(embed_channels) - 1):
        a, b = embed_channels[i], embed_channels[i + 1]
        s[f"{e}.blocks.{2 * i}.weight"] = (a, a, 3, 3); s[f"{e}.blocks.{2 * i}.bias"] = (a,)
        s[f"{e}.blocks.{2 * i + 1}.weight"] = (b, a, 3, 3); s[f"{e}.blocks.{2 * i + 1}.bias"] = (b,)
    s[f"{e}.conv_out.weight"] = (boc[0], embed_channels[-1], 3, 3); s[f"{e}.conv_out.bias"] = (boc[0],)
    chans = [boc[0]]
    for i, ch in enumerate(boc):
        chans += [ch, ch] + ([ch] if i != len(boc) - 1 else [])
    for i, ch in enumerate(chans):
        s[f"controlnet_down_blocks.{i}.weight"] = (ch, ch, 1, 1); s[f"controlnet_down_blocks.{i}.bias"] = (ch,)
    s["controlnet_mid_block.weight"] = (boc[-1], boc[-1], 1, 1); s["controlnet_mid_block.bias"] = (boc[-1],)
    return s


def random_state_dict(shapes: Dict[str, tuple], seed: int, device="cpu", dtype=torch.float32, zero_convs=False):
    """Seeded synthetic weights with fan-in scaling (keeps activations O(1)); norm weights ~ 1.
    Generated on ``device`` (a CPU generator gives streams that are identical on every host)."""
    device = torch.device(device)
    g = torch.Generator(device=device).manual_seed(seed)

    def randn(shape):
        return torch.randn(shape, generator=g, device=device)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith(".bias"):
            t = randn(shape) * 0.02
        elif ".norm" in name or name.startswith("conv_norm_out"):
            t = 1.0 + randn(shape) * 0.05
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = randn(shape) * (1.0 / math.sqrt(fan_in))
            if zero_convs and name.startswith(("controlnet_down_blocks", "controlnet_mid_block")):
                t = t * 0.1
        sd[name] = t.to(dtype=dtype)
    return sd


# ---------------------------------------------------------------------------------------------
# packed layers
# ---------------------------------------------------------------------------------------------
def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class LinearOp:
    """Packed ``nn.Linear``: weight [N, K] bf16 in HBM, bias fp32.  Callable on [..., K] tensors."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], device, dtype=bf16):
        self.weight = weight.detach().to(device=device, dtype=dtype).contiguous()
        self.bias = None if bias is None else _f32(bias, device)
        self.out_features, self.in_features = self.weight.shape

    def __call__(self, x: torch.Tensor, **kw) -> torch.Tensor:
        x2 = x.to(self.weight.dtype).contiguous().view(-1, self.in_features)
        return ops.linear(x2, self.weight, self.bias, **kw).view(*x.shape[:-1], self.out_features)


class ConvOp:
    """Packed conv: weight [Cout][ky][kx][Cin_padded] bf16."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], device, dtype=bf16):
        cout, cin, kh, kw = weight.shape
        assert (kh, kw) in ((1, 1), (3, 3))
        cin_p = (cin + 7) // 8 * 8
        w = weight.detach().to(device=device, dtype=torch.float32).permute(0, 2, 3, 1)   # [Cout, kh, kw, Cin]
        if cin_p != cin:
            w = torch.nn.functional.pad(w, (0, cin_p - cin))
        cout_p = (cout + 3) // 4 * 4
        self.weight = w.reshape(cout, kh * kw * cin_p).to(dtype).contiguous()
        self.bias = None if bias is None else _f32(bias, device)
        self.cin, self.cin_p, self.cout, self.taps = cin, cin_p, cout, kh * kw
        assert cout_p == cout, "output channels must be a multiple of 4"

    def __call__(self, x, **kw):
        return ops.conv2d_nhwc(x, self.weight, self.bias, taps=self.taps, **kw)


class NormParams:
    def __init__(self, sd, prefix, device):
        self.weight = _f32(sd[prefix + ".weight"], device)
        self.bias = _f32(sd[prefix + ".bias"], device)


class _Identity:
    def __call__(self, x):
        return x


class Attention:
    """Attribute surface of diffusers ``Attention`` that processors read
    (adapter/attention_processor.py:545-625): heads, to_q/to_k/to_v, to_out[0] (+bias), to_out[1]."""

    def __init__(self, sd, prefix, heads, device, dtype=bf16):
        self.heads = heads
        self.dtype = dtype
        self.to_q = LinearOp(sd[f"{prefix}.to_q.weight"], None, device, dtype)
        self.to_k = LinearOp(sd[f"{prefix}.to_k.weight"], None, device, dtype)
        self.to_v = LinearOp(sd[f"{prefix}.to_v.weight"], None, device, dtype)
        self.to_out = [LinearOp(sd[f"{prefix}.to_out.0.weight"], sd[f"{prefix}.to_out.0.bias"], device, dtype), _Identity()]
        self.query_dim = self.to_q.out_features
        self.is_cross = self.to_k.in_features != self.query_dim
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = None

    def prepare_attention_mask(self, mask, *a, **k):
        return mask

    def set_processor(self, p):
        self.processor = p

    def __call__(self, hidden_states, encoder_hidden_states=None, residual=None, layernorm=None, **cross_attention_kwargs):
        """``layernorm`` = (NormParams, eps): ``hidden_states`` is the block's un-normalised state.  Processors of this
        package that declare ``fused_layernorm`` run the norm inside their Q projection (row-resident kernel, cross-attention
        on 320 channels); for everything else -- foreign processors included -- it is applied here, so a processor always
        sees what the diffusers protocol promises unless it opted in."""
        proc = self.processor
        if layernorm is not None:
            nrm, eps = layernorm
            # cross-attention: norm -> to_q on 320 / 640 / 1280 channels; self-attention: norm -> q / k / v on 320 channels (the
            # processor falls back to a separate LayerNorm launch for shapes its kernels do not cover)
            if (getattr(proc, "fused_layernorm", False) and getattr(proc, "fused_residual", False) and ops.FUSED_LN
                    and (hidden_states.shape[-1] in FUSED_LN_CHANNELS if encoder_hidden_states is not None
                         else hidden_states.shape[-1] == 320)):
                cross_attention_kwargs = dict(cross_attention_kwargs, imd_layernorm=(nrm.weight, nrm.bias, eps))
            else:
                hidden_states = ops.layer_norm(hidden_states, nrm.weight, nrm.bias, eps)
        if getattr(proc, "fused_residual", False):
            return proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=None,
                        imd_residual=residual, **cross_attention_kwargs)
        out = proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=None,
                   **cross_attention_kwargs)
        out = out.to(self.dtype).contiguous()
        return out if residual is None else ops.add(out, residual)


class TransformerBlock:
    def __init__(self, sd, p, ch, heads, device, dtype=bf16):
        self.norm1 = NormParams(sd, f"{p}.norm1", device)
        self.norm2 = NormParams(sd, f"{p}.norm2", device)
        self.norm3 = NormParams(sd, f"{p}.norm3", device)
        self.attn1 = Attention(sd, f"{p}.attn1", heads, device, dtype)
        self.attn2 = Attention(sd, f"{p}.attn2", heads, device, dtype)
        # GEGLU: interleave (value_j, gate_j) rows so a lane holds both halves of a pair
        w, b = sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"]
        inner = w.shape[0] // 2
        wi = torch.stack([w[:inner], w[inner:]], dim=1).reshape(2 * inner, w.shape[1])
        bi = torch.stack([b[:inner], b[inner:]], dim=1).reshape(2 * inner)
        self.ff_in = LinearOp(wi, bi, device, dtype)
        self.ff_out = LinearOp(sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"], device, dtype)
        # 64x64 level (C = 320): norm3 -> GEGLU feed-forward -> + residual as ONE launch (csrc/ff_fused.hip); operands packed once
        self.ff_fused = None
        if ch == 320 and inner == 4 * ch:
            self.ff_fused = ops.pack_ff_fused(w.to(device), b.to(device), sd[f"{p}.ff.net.2.weight"].to(device),
                                              sd[f"{p}.ff.net.2.bias"].to(device), self.norm3.weight, self.norm3.bias, dtype=dtype)

    def __call__(self, h, ehs, cak):
        h = self.attn1(h, encoder_hidden_states=None, residual=h, layernorm=(self.norm1, 1e-5), **cak)
        return self.after_attn1(h, ehs, cak)

    def after_attn1(self, h, ehs, cak):
        h = self.attn2(h, encoder_hidden_states=ehs, residual=h, layernorm=(self.norm2, 1e-5), **cak)
        B, L, Cc = h.shape
        if self.ff_fused is not None and ops.FUSED_FF and B * L >= ops.FUSED_FF_MIN_ROWS:
            return ops.ff_geglu_fused(h.view(B * L, Cc), self.ff_fused, 1e-5).view(B, L, Cc)
        n = ops.layer_norm(h, self.norm3.weight, self.norm3.bias)
        g = ops.linear(n.view(B * L, Cc), self.ff_in.weight, self.ff_in.bias, act=ops.ACT_GEGLU)
        return ops.linear(g, self.ff_out.weight, self.ff_out.bias, res=h.view(B * L, Cc)).view(B, L, Cc)


class Transformer2D:
    def __init__(self, sd, p, ch, heads, groups, device, dtype=bf16):
        self.norm = NormParams(sd, f"{p}.norm", device)
        self.proj_in = ConvOp(sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"], device, dtype)
        self.proj_out = ConvOp(sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"], device, dtype)
        self.transformer_blocks = [TransformerBlock(sd, f"{p}.transformer_blocks.0", ch, heads, device, dtype)]
        self.groups = groups

    def __call__(self, x, ehs, cak):
        B, H, W, Cc = x.shape
        # norm -> proj_in: one launch where proj_in runs on a row-resident kernel and x came with its producer's statistics (ops.conv_gemm gn_in)
        h = self.proj_in(x, gn_in=(self.norm.weight, self.norm.bias, 1e-6, False, self.groups)).view(B, H * W, Cc)
        for blk in self.transformer_blocks:
            h = blk(h, ehs, cak)
        # (gn_stats_groups: the next resnet's norm1 finds its statistics on the result where the projection's kernel writes them -- the 8x8 level)
        return self.proj_out(h.view(B, H, W, Cc), res=x, gn_stats_groups=self.groups)

    # ---- first hybrid block of a CFG batch (round 6) ----
    def pair_half_ok(self, x_half, cak) -> bool:
        """Can this transformer run norm -> proj_in -> norm1 -> q / k / v -> the self-attention phase ONCE for the two identical halves of
        a CFG batch?  Needs the caller's word that the garment branch is on for the first half only (``sa_pair_layout``, set where the
        pipeline builds ``sa_batch_mask``), a hybrid processor that opted in (``fused_pair_half``) with its garment tokens present, and
        the attention kernel's duplicated first-phase store for this shape."""
        blk = self.transformer_blocks[0]
        proc = blk.attn1.processor
        sa = cak.get("sa_hidden_states")
        Bh, H, W, Cc = x_half.shape
        return bool(cak.get("sa_pair_layout") and cak.get("sa_batch_mask") is not None and sa is not None
                    and getattr(proc, "fused_pair_half", False) and getattr(proc, "fused_residual", False)
                    and getattr(proc, "name", None) in sa and not ops.ATTN_FP8 and not ops.FUSED_OUT_PROJ
                    and ops.attention_dup_supported(blk.attn1.heads, H * W, Cc // blk.attn1.heads))

    def call_pair_half(self, x_half, ehs, cak):
        """``x_half`` [B/2, H, W, C]: the block input of the cond rows == that of the uncond rows.  -> the block output for all B rows.
        Everything up to and including the self-attention phase of attn1 runs on B/2 rows; the attention launch writes the hybrid result
        of the cond rows and their first phase as the uncond rows' result (bit-identical to the B-row form: same kernels on the same
        values); from the out-projection on the two halves differ and the block continues on B rows."""
        Bh, H, W, Cc = x_half.shape
        blk = self.transformer_blocks[0]
        # the two residuals of the block -- the transformer's own and attn1's -- are the same for both halves: ONE copy each, added periodically by the
        # projections (imd_conv_gemm_params.res_rows; ops.conv_gemm repeats them first wherever the launch cannot)
        h = self.proj_in(x_half, gn_in=(self.norm.weight, self.norm.bias, 1e-6, False, self.groups)).view(Bh, H * W, Cc)
        r = h
        h = blk.attn1(h, encoder_hidden_states=None, residual=r, layernorm=(blk.norm1, 1e-5), imd_pair_half=True, **cak)
        h = blk.after_attn1(h, ehs, cak)
        return self.proj_out(h.view(2 * Bh, H, W, Cc), res=x_half)


def _temb_stride(temb_all):
    """batch stride of the time-embedding vector: one row per image, or ONE row for all of them (a precomputed schedule: stride 0)"""
    return temb_all.shape[1] if temb_all.shape[0] > 1 else 0


class ResnetBlock:
    def __init__(self, sd, p, groups, device, temb_slices: list, dtype=bf16):
        self.norm1 = NormParams(sd, f"{p}.norm1", device)
        self.norm2 = NormParams(sd, f"{p}.norm2", device)
        self.conv1 = ConvOp(sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], device, dtype)
        self.conv2 = ConvOp(sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], device, dtype)
        self.shortcut = None
        if f"{p}.conv_shortcut.weight" in sd:
            self.shortcut = ConvOp(sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"], device, dtype)
        self.groups = groups
        # time_emb_proj is executed as part of ONE concatenated GEMM per forward
        self.temb_off = sum(w.shape[0] for w, _ in temb_slices)
        temb_slices.append((sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"]))

    def __call__(self, x, temb_all):
        if ops.FUSED_GN_CONV and x.shape[2] >= 32 and x.shape[1] >= 8:          # (A/B switch: the 64x64 / 32x32 maps, where the halo-patch kernel is the tuned choice)
            a, b = ops.group_norm_coeffs(x, self.norm1.weight, self.norm1.bias, groups=self.groups, eps=1e-5)
            h = self.conv1(x, rowvec=temb_all, rowvec_stride=_temb_stride(temb_all), rowvec_off=self.temb_off, gn_stats_groups=self.groups, gn=(a, b, True))
            a, b = ops.group_norm_coeffs(h, self.norm2.weight, self.norm2.bias, groups=self.groups, eps=1e-5)
            sc = x if self.shortcut is None else self.shortcut(x)
            return self.conv2(h, res=sc, gn_stats_groups=self.groups, gn=(a, b, True))
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, groups=self.groups, eps=1e-5, silu=True)
        # (gn_stats_groups: where the conv runs on the halo-patch kernel its epilogue also emits the GroupNorm statistics of its
        # output, and the next group_norm of that tensor -- norm2 here, the following block's norm after conv2 -- skips its own pass)
        # (gn_out: where conv1 is K-sliced -- the 16x16 / 8x8 levels -- its finish launch applies norm2 + SiLU itself and the raw conv1 output never exists)
        h = self.conv1(h, rowvec=temb_all, rowvec_stride=_temb_stride(temb_all), rowvec_off=self.temb_off, gn_stats_groups=self.groups,
                       gn_out=(self.norm2.weight, self.norm2.bias, 1e-5, True, self.groups))
        if not getattr(h, "_imd_gn_applied", False):
            h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, groups=self.groups, eps=1e-5, silu=True)
        sc = x if self.shortcut is None else self.shortcut(x)
        return self.conv2(h, res=sc, gn_stats_groups=self.groups)


class _Encoder(PretrainedMixin):
    """conv_in + time embedding + down blocks + mid block (shared by the UNet and the ControlNet)."""
    _config_keys = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "attention_head_dim",
                    "cross_attention_dim", "norm_num_groups", "sample_size")

    def _build_encoder(self, sd, cfg, device):
        boc = cfg["block_out_channels"]
        g, heads = cfg["norm_num_groups"], cfg["attention_head_dim"]
        self._temb_slices: list = []
        self.conv_in = ConvOp(sd["conv_in.weight"], sd["conv_in.bias"], device, self.dtype)
        self.time_lin1 = LinearOp(sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"], device, self.dtype)
        self.time_lin2 = LinearOp(sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"], device, self.dtype)
        self.down_blocks = []
        self._attn: Dict[str, Attention] = {}
        for i, ch in enumerate(boc):
            blk = SimpleNamespace(resnets=[], attentions=[], downsampler=None)
            for j in range(2):
                blk.resnets.append(ResnetBlock(sd, f"down_blocks.{i}.resnets.{j}", g, device, self._temb_slices, self.dtype))
                if cfg["down_attn"][i]:
                    blk.attentions.append(self._transformer(sd, f"down_blocks.{i}.attentions.{j}", ch, heads, g, device))
            if i != len(boc) - 1:
                blk.downsampler = ConvOp(sd[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                                         sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], device, self.dtype)
            self.down_blocks.append(blk)

    def _build_mid(self, sd, cfg, device):
        boc = cfg["block_out_channels"]
        g, heads = cfg["norm_num_groups"], cfg["attention_head_dim"]
        self.mid_block = SimpleNamespace(
            resnets=[ResnetBlock(sd, "mid_block.resnets.0", g, device, self._temb_slices, self.dtype)],
            attentions=[self._transformer(sd, "mid_block.attentions.0", boc[-1], heads, g, device)])
        self.mid_block.resnets.append(ResnetBlock(sd, "mid_block.resnets.1", g, device, self._temb_slices, self.dtype))

    def _transformer(self, sd, p, ch, heads, groups, device):
        t = Transformer2D(sd, p, ch, heads, groups, device, self.dtype)
        b = t.transformer_blocks[0]
        self._attn[f"{p}.transformer_blocks.0.attn1.processor"] = b.attn1
        self._attn[f"{p}.transformer_blocks.0.attn2.processor"] = b.attn2
        return t

    def _finish_temb(self, device):
        w = torch.cat([w for w, _ in self._temb_slices], dim=0)
        b = torch.cat([b for _, b in self._temb_slices], dim=0)
        self.temb_proj = LinearOp(w, b, device, self.dtype)
        del self._temb_slices

    # ---- processors (diffusers API) ----
    @property
    def attn_processors(self) -> Dict[str, object]:
        return {name: a.processor for name, a in self._attn.items()}

    def set_attn_processor(self, processor):
        if isinstance(processor, dict):
            missing = set(self._attn) - set(processor)
            if missing:
                raise ValueError(f"set_attn_processor: {len(missing)} processor names missing, e.g. {sorted(missing)[0]}")
            for name, a in self._attn.items():
                a.set_processor(processor[name])
        else:
            for a in self._attn.values():
                a.set_processor(processor)

    # ---- shared forward pieces ----
    def _time_embed(self, timestep, B, device):
        """-> [B, sum(Cout)] fp32, or [1, sum(Cout)] when one row serves the whole batch (a schedule precomputed by
        :meth:`precompute_time_embeddings`; the resnets then read it with a batch stride of 0)."""
        fixed = self.__dict__.get("_temb_fixed")
        if fixed is not None:
            return fixed
        tab = self.__dict__.get("_temb_table")
        if tab is not None and not torch.is_tensor(timestep):
            i = tab[1].get(float(timestep))
            if i is not None:
                return tab[0][i:i + 1]
        if not torch.is_tensor(timestep):
            t = torch.full((B,), float(timestep), dtype=torch.float32, device=device)
        else:
            t = timestep.to(device=device, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        return self._time_embed_rows(t)

    def _time_embed_rows(self, t):
        """t [R] fp32 timesteps -> [R, sum(Cout)] fp32: sinusoid -> linear_1 -> SiLU -> linear_2 (-> SiLU) -> every resnet's time_emb_proj as one GEMM
        (diffusers UNet2DConditionModel.forward: time_proj, time_embedding; ResnetBlock2D.time_emb_proj(nonlinearity(temb)))."""
        e = ops.f32_to_16(ops.timestep_embedding(t, self.time_lin1.in_features), self.dtype)
        e = ops.linear(e, self.time_lin1.weight, self.time_lin1.bias, act=ops.ACT_SILU)
        # every consumer applies SiLU first (ResnetBlock2D.time_emb_proj(nonlinearity(temb))): store silu(temb)
        e = ops.linear(e, self.time_lin2.weight, self.time_lin2.bias, act=ops.ACT_SILU)
        return ops.linear(e, self.temb_proj.weight, self.temb_proj.bias, out_f32=True)     # [R, sum(Cout)] fp32

    # The chain above depends on the timestep only -- not on the latent -- and the denoising loop knows its whole schedule up front (round 6): the
    # pipeline runs it ONCE over all timesteps of a call (R = 50 rows instead of 50 passes of one row: 7 launches, ~65 us, per UNet forward) and every
    # forward of the loop picks its row.  All images of a batch share the timestep, so the row is handed to the resnets as a [1, C] vector with batch
    # stride 0.  Engine-internal: a forward outside a pipeline call computes the chain as before.
    def precompute_time_embeddings(self, timesteps, device):
        ts = [float(t) for t in timesteps]
        table = self._time_embed_rows(torch.tensor(ts, dtype=torch.float32).to(device))
        self._temb_table = (table, {k: i for i, k in enumerate(ts)})
        return table

    def use_time_embedding(self, row):
        """``row`` [1, sum(Cout)] fp32 (a fixed buffer the caller refreshes per step: HIP-graph replay) or None."""
        self._temb_fixed = row

    def clear_time_embeddings(self):
        self._temb_table = None
        self._temb_fixed = None

    def _run_down(self, x, temb_all, ehs, cak, pair_skip=None, pair_attn_done=False):
        """``pair_skip``: the caller has already run conv_in and the first resnet on ONE half of a CFG batch whose halves are identical
        up to there (``x`` = that resnet's output repeated for both halves, ``pair_skip`` = the half-batch conv_in output);
        ``pair_attn_done``: ... and the first transformer as well (``Transformer2D.call_pair_half``; ``x`` = its output)."""
        skips = [x if pair_skip is None else pair_skip]
        first = pair_skip is not None
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                if first:
                    first = False                       # (already applied by the caller)
                    if blk.attentions and not pair_attn_done:
                        x = blk.attentions[j](x, ehs, cak)
                else:
                    x = r(x, temb_all)
                    if blk.attentions:
                        x = blk.attentions[j](x, ehs, cak)
                skips.append(x)
            if blk.downsampler is not None:
                x = blk.downsampler(x, stride=2, gn_stats_groups=self.cfg["norm_num_groups"])   # feeds the next level's norm1 (K-sliced: statistics from the finish launch)
                skips.append(x)
        return x, skips

    def _run_mid(self, x, temb_all, ehs, cak):
        x = self.mid_block.resnets[0](x, temb_all)
        x = self.mid_block.attentions[0](x, ehs, cak)
        return self.mid_block.resnets[1](x, temb_all)


def _default_processor():
    from .adapter.attention_processor import AttnProcessor2_0
    return AttnProcessor2_0()


def nchw_to_nhwc8(x: torch.Tensor, dtype=bf16) -> torch.Tensor:
    """[B, C<=8, H, W] any float dtype -> [B, H, W, 8] 16-bit (zero padded).  Boundary glue only."""
    B, Cc, H, W = x.shape
    out = torch.zeros(B, H, W, (Cc + 7) // 8 * 8, dtype=dtype, device=x.device)
    out[..., :Cc] = x.permute(0, 2, 3, 1)
    return out


class UNet2DConditionModel(_Encoder):
    def __init__(self, state_dict: Dict[str, torch.Tensor], config: Optional[dict] = None, device="cuda", dtype=bf16):
        cfg = dict(SD15_CONFIG, **(config or {}))
        if dtype not in ops.DTYPE_CODE:
            raise ValueError(f"dtype must be torch.bfloat16 or torch.float16, got {dtype}")
        want = unet_param_shapes(cfg)
        missing = [k for k in want if k not in state_dict]
        if missing:
            raise KeyError(f"UNet state dict is missing {len(missing)} keys, e.g. {missing[0]}")
        for k, shp in want.items():
            if tuple(state_dict[k].shape) != shp:
                raise ValueError(f"{k}: shape {tuple(state_dict[k].shape)} != expected {shp}")
        self.cfg = cfg
        self._ctor_config = config
        self.config = SimpleNamespace(**cfg)
        self.device = torch.device(device)
        self.dtype = dtype
        sd, dev = state_dict, self.device
        boc = cfg["block_out_channels"]
        g, heads = cfg["norm_num_groups"], cfg["attention_head_dim"]
        self._build_encoder(sd, cfg, dev)
        # registration order: down, up, mid (matches diffusers' named_children order)
        enc_attn = self._attn
        self._attn = {}
        self.up_blocks = []
        rev = list(reversed(boc))
        up_attn = list(reversed(cfg["down_attn"]))
        # resnets must be registered in execution order for the concatenated temb GEMM: build mid first
        # into a temporary, then ups; offsets are per-resnet so order only needs to be consistent.
        self._build_mid(sd, cfg, dev)
        mid_attn = self._attn
        self._attn = {}
        for i, ch in enumerate(rev):
            blk = SimpleNamespace(resnets=[], attentions=[], upsampler=None)
            for j in range(3):
                blk.resnets.append(ResnetBlock(sd, f"up_blocks.{i}.resnets.{j}", g, dev, self._temb_slices, self.dtype))
                if up_attn[i]:
                    blk.attentions.append(self._transformer(sd, f"up_blocks.{i}.attentions.{j}", ch, heads, g, dev))
            if i != len(boc) - 1:
                blk.upsampler = ConvOp(sd[f"up_blocks.{i}.upsamplers.0.conv.weight"],
                                       sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], dev, self.dtype)
            self.up_blocks.append(blk)
        up_attn_map = self._attn
        self._attn = {**enc_attn, **up_attn_map, **mid_attn}
        self._finish_temb(dev)
        self.conv_norm_out = NormParams(sd, "conv_norm_out", dev)
        self.conv_out = ConvOp(sd["conv_out.weight"], sd["conv_out.bias"], dev, self.dtype)
        self.set_attn_processor(_default_processor())

    @classmethod
    def random_init(cls, seed: int = 0, config: Optional[dict] = None, device="cuda", dtype=bf16):
        cfg = dict(SD15_CONFIG, **(config or {}))
        return cls(random_state_dict(unet_param_shapes(cfg), seed, device=device), cfg, device, dtype)

    # ------------------------------------------------------------------------------------
    def forward_nhwc(self, x: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                     cross_attention_kwargs: Optional[dict] = None,
                     down_block_additional_residuals: Optional[List[torch.Tensor]] = None,
                     mid_block_additional_residual: Optional[torch.Tensor] = None, cfg_pair: bool = False) -> torch.Tensor:
        """x [B, H, W, 8] bf16 (latent channels zero-padded to 8) -> eps [B, H*W, 4] fp32.
        ``cfg_pair``: the caller guarantees x[B/2:] == x[:B/2] (the CFG batch of the sampling loop: the reference feeds the SAME latent to
        its cond and uncond UNet calls, IMAGDressing_v1_pipeline.py:483-512).  conv_in and the first resnet see neither the text nor the
        garment, so their outputs are identical for the two halves: they run on one half and the result is repeated."""
        cak = dict(cross_attention_kwargs or {})
        B, H, W, _ = x.shape
        ehs = encoder_hidden_states
        temb_all = self._time_embed(timestep, B, x.device)
        if cfg_pair and B % 2 == 0 and ops.CFG_PAIR_DEDUP:
            h0 = self.conv_in(x[:B // 2], gn_stats_groups=self.cfg["norm_num_groups"])
            blk0 = self.down_blocks[0]
            r0 = blk0.resnets[0](h0, temb_all)
            # the first transformer's input is still identical for the two halves: norm / proj_in / norm1 / q-k-v / the self-attention
            # phase run once per image (Transformer2D.call_pair_half); the halves part ways at attn1's out-projection
            if ops.CFG_PAIR_ATTN and blk0.attentions and blk0.attentions[0].pair_half_ok(r0, cak):
                h = blk0.attentions[0].call_pair_half(r0, ehs, cak)
                h, skips = self._run_down(h, temb_all, ehs, cak, pair_skip=h0, pair_attn_done=True)
            else:
                h, skips = self._run_down(ops.repeat_batch(r0), temb_all, ehs, cak, pair_skip=h0)
        else:
            h = self.conv_in(x, gn_stats_groups=self.cfg["norm_num_groups"])          # (feeds the first resnet's norm1)
            h, skips = self._run_down(h, temb_all, ehs, cak)
        h = self._run_mid(h, temb_all, ehs, cak)
        if mid_block_additional_residual is not None:
            h = ops.add(h, mid_block_additional_residual)
        ctrl = down_block_additional_residuals
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                s = skips.pop()
                c = None if ctrl is None else ctrl[len(skips)]
                h = ops.concat_channels(h, s, c, gn_stats_groups=r.groups)          # cat([x, skip (+ ControlNet residual)]) + the statistics of the resnet's norm1
                h = r(h, temb_all)
                if blk.attentions:
                    h = blk.attentions[j](h, ehs, cak)
            if blk.upsampler is not None:
                h = blk.upsampler(h, ups=True)
        h = ops.group_norm(h, self.conv_norm_out.weight, self.conv_norm_out.bias,
                           groups=self.cfg["norm_num_groups"], eps=1e-5, silu=True)
        eps = self.conv_out(h, out_f32=True)
        return eps.view(B, H * W, self.cfg["out_channels"])

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict: bool = False, **unused):
        """diffusers-compatible entry: NCHW in, (NCHW,) out in the input dtype."""
        x = nchw_to_nhwc8(sample.to(self.device), self.dtype)
        B, _, H, W = sample.shape

        def res_nhwc(r):
            if r.dim() == 3:               # the reference indexes residuals by batch: down_block[1] (:662-666)
                r = r.unsqueeze(0)
            return r.to(device=self.device, dtype=self.dtype).permute(0, 2, 3, 1).contiguous()
        dres = None if down_block_additional_residuals is None else [res_nhwc(r) for r in down_block_additional_residuals]
        mres = None if mid_block_additional_residual is None else res_nhwc(mid_block_additional_residual)
        ehs = encoder_hidden_states.to(device=self.device, dtype=self.dtype).contiguous()
        eps = self.forward_nhwc(x, timestep, ehs, cross_attention_kwargs, dres, mres)
        out = eps.view(B, H, W, -1).permute(0, 3, 1, 2).to(sample.dtype)
        return (out,)

    __call__ = forward


class ControlNetModel(_Encoder):
    """SD1.5 ControlNet: encoder copy + conditioning embedding + 13 zero-convs.  Returns NHWC bf16
    residuals (consumed by ``UNet2DConditionModel.forward_nhwc``)."""

    def __init__(self, state_dict, config: Optional[dict] = None, device="cuda", dtype=bf16):
        cfg = dict(SD15_CONFIG, **(config or {}))
        want = controlnet_param_shapes(cfg)
        missing = [k for k in want if k not in state_dict]
        if missing:
            raise KeyError(f"ControlNet state dict is missing {len(missing)} keys, e.g. {missing[0]}")
        self.cfg = cfg
        self._ctor_config = config
        self.config = SimpleNamespace(global_pool_conditions=False, **cfg)
        self.device = torch.device(device)
        self.dtype = dtype
        sd, dev = state_dict, self.device
        self._build_encoder(sd, cfg, dev)
        self._build_mid(sd, cfg, dev)
        self._finish_temb(dev)
        e = "controlnet_cond_embedding"
        self.cond_convs = [ConvOp(sd[f"{e}.conv_in.weight"], sd[f"{e}.conv_in.bias"], dev, dtype)]
        self.cond_strides = [1]
        i = 0
        while f"{e}.blocks.{i}.weight" in sd:
            self.cond_convs.append(ConvOp(sd[f"{e}.blocks.{i}.weight"], sd[f"{e}.blocks.{i}.bias"], dev, dtype))
            self.cond_strides.append(2 if i % 2 == 1 else 1)
            i += 1
        self.cond_out = ConvOp(sd[f"{e}.conv_out.weight"], sd[f"{e}.conv_out.bias"], dev, dtype)
        self.zero_convs = []
        i = 0
        while f"controlnet_down_blocks.{i}.weight" in sd:
            self.zero_convs.append(ConvOp(sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"], dev, dtype))
            i += 1
        self.zero_mid = ConvOp(sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"], dev, dtype)
        self.set_attn_processor(_default_processor())
        self._cond_cache = None

    @classmethod
    def random_init(cls, seed: int = 1, config: Optional[dict] = None, device="cuda", dtype=bf16):
        cfg = dict(SD15_CONFIG, **(config or {}))
        return cls(random_state_dict(controlnet_param_shapes(cfg), seed, device=device, zero_convs=True), cfg, device, dtype)

    def cond_embedding(self, cond_nhwc8: torch.Tensor) -> torch.Tensor:
        """controlnet_cond [Bc, H, W, 8] bf16 -> [Bc, H/8, W/8, 320]; step-invariant, cached per image."""
        key = (cond_nhwc8.data_ptr(), cond_nhwc8._version, tuple(cond_nhwc8.shape))
        if self._cond_cache is not None and self._cond_cache[0] == key and self._cond_cache[1] is cond_nhwc8:
            return self._cond_cache[2]
        e = cond_nhwc8
        for conv, st in zip(self.cond_convs, self.cond_strides):
            e = conv(e, stride=st, act=ops.ACT_SILU)
        e = self.cond_out(e)
        self._cond_cache = (key, cond_nhwc8, e)
        return e

    def forward_nhwc(self, x, timestep, encoder_hidden_states, cond_nhwc8, conditioning_scale: float = 1.0):
        """x [B, H, W, 8]; cond [Bc, 8H, 8W, 8] with Bc == B or 1 -> (list of 12 NHWC residuals, mid)."""
        B = x.shape[0]
        temb_all = self._time_embed(timestep, B, x.device)
        emb = self.cond_embedding(cond_nhwc8)
        if emb.shape[0] != B:
            emb = emb.expand(B, -1, -1, -1).contiguous()
        h = self.conv_in(x, res=emb)
        h, skips = self._run_down(h, temb_all, encoder_hidden_states, {})
        h = self._run_mid(h, temb_all, encoder_hidden_states, {})
        down = []
        for s, zc in zip(skips, self.zero_convs):
            down.append(_scaled_conv1x1(zc, s, conditioning_scale))
        mid = _scaled_conv1x1(self.zero_mid, h, conditioning_scale)
        return down, mid

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                guess_mode=False, return_dict=False, **unused):
        """diffusers-compatible entry (NCHW in); residuals are returned NHWC-tagged bf16 tensors in
        NCHW *view* order so ``down[i][1]`` style indexing by batch keeps working."""
        x = nchw_to_nhwc8(sample.to(self.device), self.dtype)
        cond = nchw_to_nhwc8(controlnet_cond.to(self.device), self.dtype)
        ehs = encoder_hidden_states.to(device=self.device, dtype=self.dtype).contiguous()
        down, mid = self.forward_nhwc(x, timestep, ehs, cond, float(conditioning_scale))
        return [d.permute(0, 3, 1, 2) for d in down], mid.permute(0, 3, 1, 2)

    __call__ = forward


def _scaled_conv1x1(conv: ConvOp, x, scale: float):
    """zero-conv: (W x + b) * scale  ==  out_scale applied after the bias in the epilogue."""
    return ops.conv2d_nhwc(x, conv.weight, conv.bias, taps=1, out_scale=scale)
