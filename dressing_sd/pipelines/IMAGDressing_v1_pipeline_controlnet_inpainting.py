"""Shim: same module path as the reference's ``dressing_sd/pipelines/IMAGDressing_v1_pipeline_controlnet_inpainting.py``."""
from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline_controlnet_inpainting import IMAGDressing_v1  # noqa: F401
