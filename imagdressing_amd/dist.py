"""Multi-GPU: one process per GPU, images sharded over ranks, ONE collective per garment.

The reference has no inference parallelism (batch_size = 1 hard-coded,
/root/reference/dressing_sd/pipelines/IMAGDressing_v1_pipeline.py:389).  Images of a batch are
independent given the garment features, so the batch shards over ranks with no data-path collective
inside the loop.  The only exchange is the step-invariant garment feature set (16 post-LayerNorm
token matrices, 23.1 MB bf16 at 512x512): rank 0 runs the garment UNet and broadcasts ONE packed
buffer (RCCL ``ncclBroadcast`` over xGMI; backend "nccl" on ROCm is RCCL, "gloo" on CPU for tests).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_bounds(n: int, r: int = None, w: int = None) -> Tuple[int, int]:
    """Contiguous block partition of n items: the first n % w ranks get one extra."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_rows(t: torch.Tensor) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0])
    return t[lo:hi]


def pack_features(feats: Dict[str, torch.Tensor], names: List[str]) -> Tuple[torch.Tensor, List[Tuple[str, tuple]]]:
    """Flatten the named tensors (in ``names`` order) into one contiguous buffer + a layout table."""
    layout = [(n, tuple(feats[n].shape)) for n in names]
    flat = torch.cat([feats[n].reshape(-1) for n in names]) if names else torch.empty(0)
    return flat.contiguous(), layout


def unpack_features(flat: torch.Tensor, layout: List[Tuple[str, tuple]]) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for name, shape in layout:
        n = 1
        for s in shape:
            n *= s
        out[name] = flat[off:off + n].view(shape)
        off += n
    return out


def feature_layout(unet_like, ref_hw: Tuple[int, int]) -> List[Tuple[str, tuple]]:
    """Shapes of the garment features every rank can derive locally (no metadata exchange): one
    [1, M_l, C_l] entry per attention layer of the garment UNet, for a garment latent of ref_hw."""
    cfg = unet_like.cfg
    boc = cfg["block_out_channels"]
    h, w = ref_hw
    layout = []
    level_of = {}
    # a stride-2 3x3 conv with padding 1 maps n -> ceil(n / 2) (odd latent sizes round UP, level by level)
    hs, ws = [int(h)], [int(w)]
    for _ in range(len(boc) - 1):
        hs.append((hs[-1] + 1) // 2)
        ws.append((ws[-1] + 1) // 2)
    if hs[-1] * 2 ** (len(boc) - 1) != hs[0] or ws[-1] * 2 ** (len(boc) - 1) != ws[0]:
        # the up path doubles sizes back: only latents divisible by 2^(levels-1) make the skip shapes line up (as in
        # diffusers); say so on EVERY rank before anybody enters the collective
        raise ValueError(f"garment latent {h}x{w} is not divisible by {2 ** (len(boc) - 1)}: the UNet's skip connections would not line up")
    for i, ch in enumerate(boc):
        level_of[f"down_blocks.{i}"] = (ch, hs[i] * ws[i])
    rev = list(reversed(boc))
    for i, ch in enumerate(rev):
        lv = len(boc) - 1 - i
        level_of[f"up_blocks.{i}"] = (ch, hs[lv] * ws[lv])
    level_of["mid_block"] = (boc[-1], hs[-1] * ws[-1])
    for name in unet_like.attn_processors.keys():
        key = name.split(".attentions")[0]
        ch, tokens = level_of[key]
        layout.append((name, (1, tokens, ch)))
    return layout


def broadcast_packed(flat: torch.Tensor, src: int = 0) -> torch.Tensor:
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


@torch.no_grad()
def garment_features_broadcast(pipe, ref_latents: torch.Tensor, cloth_tokens: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Rank 0 computes the garment features; everyone receives them in one broadcast."""
    runet = pipe.reference_unet
    hw = (ref_latents.shape[-2], ref_latents.shape[-1])
    layout = [e for e in feature_layout(runet, hw) if e[0].endswith("attn1.processor")]   # only attn1 is consumed
    names = [n for n, _ in layout]
    total = sum(s[1] * s[2] for _, s in layout)
    # ONE packed broadcast: [features | status] (status LAST: the feature views keep the buffer's alignment).  The status element (0 = ok) travels with the payload, so a failure on rank 0
    # (layout mismatch, an exception in the garment pass) raises on EVERY rank instead of leaving the others to denoise NaNs or
    # to hang in the next collective after rank 0 has left.
    flat = torch.empty(total + 1, dtype=runet.dtype, device=pipe.device)
    err = None
    if rank() == 0:
        try:
            feats = pipe.garment_features(ref_latents, cloth_tokens)
            body, lay0 = pack_features({n: feats[n].to(runet.dtype) for n in names}, names)
            if lay0 != layout:
                raise RuntimeError(f"garment feature layout mismatch: derived {layout[:2]}..., garment UNet produced {lay0[:2]}...")
            flat[:total] = body
            flat[total] = 0.0
        except Exception as e:       # noqa: BLE001  (re-raised below, after the other ranks have been released)
            err = e
            try:                     # a device-side fault makes these two launches raise as well: the status then travels in a FRESH
                flat.zero_()         # host-built buffer (below) -- rank 0 must reach the broadcast whatever happened, or the others hang
                flat[total] = 1.0
            except Exception:        # noqa: BLE001
                flat = None
    if err is not None and flat is None:
        try:
            host = torch.zeros(total + 1, dtype=runet.dtype)
            host[total] = 1.0
            flat = host.to(pipe.device)
        except Exception:            # noqa: BLE001  (the device is gone: nothing can be sent; the process group's own timeout releases the others)
            raise err
    try:
        broadcast_packed(flat, 0)
    finally:
        if err is not None:
            raise err
    if rank() != 0 and float(flat[total].item()) != 0.0:
        raise RuntimeError("rank 0 failed while computing the garment features (status flag of the packed broadcast); see its traceback")
    flat = flat[:total]
    return unpack_features(flat, layout)
