"""Opt-in import shim: put ``<repo>/compat`` and ``<repo>`` at the FRONT of ``sys.path`` (or ``PYTHONPATH``) and the reference's
``inference_IMAGdressing*.py`` run UNCHANGED on the MI355X implementation --

    PYTHONPATH=/path/to/repo/compat:/path/to/repo  IMD_MODEL_ROOT=/models  python inference_IMAGdressing.py --cloth_path x.jpg

The scripts' ``from diffusers import UNet2DConditionModel, AutoencoderKL, DDIMScheduler, ControlNetModel``
(/root/reference/inference_IMAGdressing.py:6, ..._controlnetpose.py:8-9) resolve to the HIP engines, whose
``from_pretrained(...).to(dtype=, device=)`` / ``.config`` / ``.attn_processors`` / ``.set_attn_processor`` / ``.load_state_dict``
surface is what ``prepare()`` uses (:42-135).  This package is NOT diffusers: it exports only the names the reference scripts
import.  Hub ids ("SG161222/Realistic_Vision_V4.0_noVAE") are looked up under ``$IMD_MODEL_ROOT`` and the local Hugging Face
cache (imagdressing_amd/hub.py) -- there is no network client.

The scripts take the CLIP towers from ``transformers`` (:11).  ``transformers`` itself stays the installed library (tokenizer,
image processor); unless ``IMD_NATIVE_CLIP=0`` its two model classes ``CLIPTextModel`` / ``CLIPVisionModelWithProjection`` are
re-pointed at the MI355X engines (imagdressing_amd/clip.py, parity-tested against the library's own modules) when this shim is
imported, which the scripts do before they import ``transformers``' models.
"""
import os as _os

from imagdressing_amd.scheduler import DDIMScheduler, UniPCMultistepScheduler  # noqa: F401
from imagdressing_amd.unet import ControlNetModel, UNet2DConditionModel  # noqa: F401
from imagdressing_amd.vae import AutoencoderKL  # noqa: F401

__version__ = "0.24.0+imagdressing_amd"
__all__ = ["UNet2DConditionModel", "ControlNetModel", "AutoencoderKL", "DDIMScheduler", "UniPCMultistepScheduler"]


def _use_native_clip():
    try:
        import transformers as _tf
    except ImportError:          # the scripts would fail on their own import line; nothing to patch
        return False
    from imagdressing_amd.clip import CLIPTextModel, CLIPVisionModelWithProjection
    _tf.CLIPTextModel = CLIPTextModel
    _tf.CLIPVisionModelWithProjection = CLIPVisionModelWithProjection
    return True


NATIVE_CLIP = _use_native_clip() if _os.environ.get("IMD_NATIVE_CLIP", "1") != "0" else False
