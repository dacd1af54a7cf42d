"""Processor-name helpers shared by the fixture builders (no oracle / engine imports: usable from bench.py)."""


def hidden_size_of(name, boc):
    if name.startswith("mid_block"):
        return boc[-1]
    if name.startswith("up_blocks"):
        return list(reversed(boc))[int(name[len("up_blocks.")])]
    return boc[int(name[len("down_blocks.")])]


def attn_processor_names(cfg):
    """diffusers' ``unet.attn_processors`` key order for an SD1.5-shaped config (down blocks, up blocks, mid block)."""
    n_levels = len(cfg["block_out_channels"])
    lpb = cfg.get("layers_per_block", 2)
    names = []
    for i in range(n_levels - 1):               # CrossAttnDownBlock2D x (levels - 1), last down block has no attention
        for j in range(lpb):
            for a in ("attn1", "attn2"):
                names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for i in range(1, n_levels):                # first up block has no attention
        for j in range(lpb + 1):
            for a in ("attn1", "attn2"):
                names.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for a in ("attn1", "attn2"):
        names.append(f"mid_block.attentions.0.transformer_blocks.0.{a}.processor")
    return names


def attn1_names(cfg):
    return [n for n in attn_processor_names(cfg) if n.endswith("attn1.processor")]
