"""Import the reference's ``adapter/attention_processor.py`` and ``adapter/resampler.py``
VERBATIM from ``/root/reference`` (build container only -- that path does not exist on the GPU
box) behind a two-symbol ``diffusers`` stub (TEST INFRASTRUCTURE).

The reference module needs only ``diffusers.utils.USE_PEFT_BACKEND`` and
``diffusers.models.lora.LoRALinearLayer`` (attention_processor.py:6-7); ``diffusers`` itself
(pinned 0.24.0, requirements.txt:12) is not installed and cannot be (no network).  The stub
``LoRALinearLayer`` restates the 0.24.0 class (down/up Linear without bias, optional
``network_alpha / rank`` scaling, down ~ N(0, 1/rank), up = 0).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("IMAGDRESSING_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "adapter", "attention_processor.py"))


class LoRALinearLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False, device=device, dtype=dtype)
        self.up = nn.Linear(rank, out_features, bias=False, device=device, dtype=dtype)
        self.network_alpha = network_alpha
        self.rank = rank
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, hidden_states):
        orig = hidden_states.dtype
        dt = self.down.weight.dtype
        y = self.up(self.down(hidden_states.to(dt)))
        if self.network_alpha is not None:
            y = y * (self.network_alpha / self.rank)
        return y.to(orig)


def _install_stub():
    if "diffusers" in sys.modules and not getattr(sys.modules["diffusers"], "_imd_stub", False):
        return  # a real diffusers is importable: use it
    d = types.ModuleType("diffusers"); d._imd_stub = True
    u = types.ModuleType("diffusers.utils"); u.USE_PEFT_BACKEND = False
    m = types.ModuleType("diffusers.models")
    l = types.ModuleType("diffusers.models.lora"); l.LoRALinearLayer = LoRALinearLayer
    d.utils, d.models, m.lora = u, m, l
    sys.modules.update({"diffusers": d, "diffusers.utils": u, "diffusers.models": m, "diffusers.models.lora": l})


def _load(modname, relpath):
    path = os.path.join(REFERENCE_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference_adapter():
    """-> (attention_processor module, resampler module) of the reference, unmodified."""
    if not reference_available():
        raise FileNotFoundError(f"reference tree not found under {REFERENCE_ROOT}")
    _install_stub()
    ap = _load("_imd_ref_attention_processor", "adapter/attention_processor.py")
    rs = _load("_imd_ref_resampler", "adapter/resampler.py")
    return ap, rs
