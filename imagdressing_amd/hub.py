"""``from_pretrained`` for the MI355X engines: the model-construction lines of the reference's ``prepare()``
(/root/reference/inference_IMAGdressing.py:42-52, :90-92; ..._ipa_controlnetpose.py / ..._controlnetinpainting.py likewise)
keep their shape --

    unet = UNet2DConditionModel.from_pretrained(path, subfolder="unet").to(dtype=torch.float16, device=args.device)

-- with the class imported from ``imagdressing_amd`` instead of ``diffusers`` / ``transformers``.  Only LOCAL directories in
the Hugging Face layout are read (``<path>/<subfolder>/config.json`` + ``diffusion_pytorch_model.safetensors`` |
``model.safetensors`` | ``*.bin``); there is no hub client here.  ``from_pretrained`` returns a light handle holding the
CPU state dict; ``.to(dtype=, device=)`` is where the engine is built (weights are repacked once for the kernels and land in
HBM -- there is no CPU engine to move from), so the script's chained ``.to(...)`` costs nothing extra.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                "diffusion_pytorch_model.bin", "pytorch_model.bin")


def resolve_model_dir(name: str) -> str:
    """A local directory for ``name``: the path itself, else -- for a hub id such as "SG161222/Realistic_Vision_V4.0_noVAE", which
    the reference scripts hard-code (inference_IMAGdressing.py:42-52) -- ``$IMD_MODEL_ROOT/<org>/<name>``, else the newest snapshot
    of the local Hugging Face cache (``$HF_HOME`` / ``~/.cache/huggingface``).  Nothing is downloaded."""
    if os.path.isdir(name):
        return name
    roots = [os.environ.get("IMD_MODEL_ROOT")]
    for r in roots:
        if r and os.path.isdir(os.path.join(r, name)):
            return os.path.join(r, name)
    hf = os.environ.get("HF_HOME") or os.path.join(os.path.expanduser("~"), ".cache", "huggingface")
    snap = os.path.join(hf, "hub", "models--" + name.replace("/", "--"), "snapshots")
    if os.path.isdir(snap):
        cands = sorted((os.path.join(snap, d) for d in os.listdir(snap)), key=os.path.getmtime)
        if cands:
            return cands[-1]
    return name


def read_pretrained_dir(path: str, subfolder: Optional[str] = None):
    """-> (state dict on the CPU, config dict) of a local Hugging Face model directory."""
    path = resolve_model_dir(path)
    d = os.path.join(path, subfolder) if subfolder else path
    if not os.path.isdir(d):
        raise FileNotFoundError(f"from_pretrained: {d!r} is not a local directory (this build has no hub access; download the "
                                "model and pass its path, or put it under $IMD_MODEL_ROOT/<org>/<name>)")
    cfg = {}
    cj = os.path.join(d, "config.json")
    if os.path.isfile(cj):
        with open(cj) as f:
            cfg = json.load(f)
    for name in WEIGHT_FILES:
        f = os.path.join(d, name)
        if os.path.isfile(f):
            from .checkpoint import load_state_dict_file
            return load_state_dict_file(f), cfg
    raise FileNotFoundError(f"from_pretrained: none of {WEIGHT_FILES} found in {d!r}")


class PendingModel:
    """What ``Engine.from_pretrained`` returns: ``.to(dtype=, device=)`` (or any positional mix torch accepts) builds the
    engine.  ``.config`` is available before that (the reference reads ``unet.config.cross_attention_dim`` only after
    ``.to``, but nothing forbids the other order)."""

    def __init__(self, cls, state_dict: Dict[str, torch.Tensor], config: dict, engine_config: Optional[dict]):
        self._cls, self._sd, self._engine_config = cls, state_dict, engine_config
        from types import SimpleNamespace
        self.config = SimpleNamespace(**config)

    def to(self, *args, dtype=None, device=None, **unused):
        """``.to(dtype=, device=)`` builds the engine; ``.to(dtype)`` alone (or ``from_pretrained(torch_dtype=...)``,
        ..._controlnetpose.py:137-138) only records the element type -- the engine is built when the device arrives, or on first use."""
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, torch.device, int)):
                device = a
        if dtype is not None:
            self.__dict__["_dtype"] = dtype
        if device is None:
            return self
        eng = self.__dict__.get("_engine")              # (already built by a first use: same engine, `.to(device)` is then a no-op check)
        if eng is None:                                 # keep it: a caller that holds on to the HANDLE must reach the same engine later
            eng = self.__dict__["_engine"] = self._build(device)
            return eng
        return eng.to(device=device)

    def _build(self, device="cuda"):
        dtype = self.__dict__.get("_dtype") or torch.float16      # the reference's dtype (inference_IMAGdressing.py:42-52)
        if self._sd is None:
            raise RuntimeError(f"{self._cls.__name__}.from_pretrained(...) handle: the engine was already built (its weights were handed over); "
                               "use the engine `.to(device=...)` returned")
        eng = self._cls(self._sd, self._engine_config, device, dtype)
        self.__dict__["_sd"] = None
        return eng

    def __getattr__(self, name):
        """First use of the engine surface (``set_attn_processor``, ``load_state_dict``, ``attn_processors``, a call ...) on a handle that
        has its element type but never got a device: the engine is built NOW on the current HIP device ("on first use", as ``to``'s
        docstring says) and the handle forwards to it from then on.  Without any ``.to`` / ``torch_dtype`` the handle stays inert."""
        if name.startswith("_") or "_dtype" not in self.__dict__:        # (private probes such as hasattr(h, "_hf_hook") must not build an engine)
            raise AttributeError(f"{self._cls.__name__}.from_pretrained(...) handle has no attribute {name!r}: call "
                                 ".to(dtype=..., device=...) first (that is where the MI355X engine is built)")
        eng = self.__dict__.get("_engine")
        if eng is None:
            eng = self.__dict__["_engine"] = self._build("cuda")
        return getattr(eng, name)

    def __call__(self, *args, **kw):
        self.__getattr__("dtype")                       # builds on first use (or raises the "call .to(...) first" error)
        return self.__dict__["_engine"](*args, **kw)


class PretrainedMixin:
    """``from_pretrained`` / ``load_state_dict`` / ``state_dict`` / ``to`` / ``eval`` / ``requires_grad_`` for an engine whose
    constructor is ``cls(state_dict, config, device, dtype)``."""

    #: config.json keys that map onto the engine's config dict (everything else is the fixed SD1.5 architecture)
    _config_keys = ()

    @classmethod
    def _engine_config(cls, hf_config: dict) -> Optional[dict]:
        out = {}
        for k in cls._config_keys:
            if k in hf_config and hf_config[k] is not None:
                v = hf_config[k]
                out[k] = tuple(v) if isinstance(v, list) else v
        return out or None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None, **unused):
        sd, cfg = read_pretrained_dir(str(pretrained_model_name_or_path), subfolder)
        pend = PendingModel(cls, sd, cfg, cls._engine_config(cfg))
        return pend if torch_dtype is None else pend.to(dtype=torch_dtype)

    def to(self, *args, dtype=None, device=None, **unused):
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, (str, torch.device, int)):
                device = a
        if dtype is not None and dtype != self.dtype:
            raise ValueError(f"{type(self).__name__} was built as {self.dtype}; engines are not re-cast (weights are repacked "
                             f"for the kernels at build time) -- build it with dtype={dtype}")
        if device is not None and torch.device(device).type != "cuda":
            raise ValueError(f"{type(self).__name__} runs on MI355X only; cannot move to {device}")
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag: bool = False):
        return self

    def state_dict(self):
        """The engines keep weights REPACKED (NHWC filters, interleaved GEGLU rows, fused time-embedding projection, 16-bit),
        not in the diffusers layout, so there is nothing meaningful to hand back; the reference only takes and deletes this
        (inference_IMAGdressing.py:68,88)."""
        return {}

    def load_state_dict(self, state_dict, strict: bool = True):
        """Rebuild the engine from a diffusers-layout state dict (``ref_unet.load_state_dict(ref_unet_dict)``,
        inference_IMAGdressing.py:114); installed attention processors are kept.  A FRESH instance is built first and swapped
        in only on success, so a bad state dict (missing keys, wrong shapes) leaves this object exactly as it was.
        ``strict=False`` is refused: the engines repack every weight at build time and cannot run with a partial set."""
        if not strict:
            raise NotImplementedError(f"{type(self).__name__}.load_state_dict(strict=False): the engine needs the complete diffusers-layout "
                                      "state dict (weights are repacked for the kernels at build time)")
        procs = dict(self.attn_processors) if hasattr(self, "attn_processors") else None
        cfg = getattr(self, "_ctor_config", None)
        fresh = type(self)(state_dict, cfg, self.device, self.dtype)
        if procs is not None:
            fresh.set_attn_processor(procs)
        self.__dict__.clear()
        self.__dict__.update(fresh.__dict__)
        return torch.nn.modules.module._IncompatibleKeys([], [])
