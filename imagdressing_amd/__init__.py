"""imagdressing_amd -- MI355X-native (gfx950) implementation of the IMAGDressing-v1 denoising hot
path behind the reference's own plugin surface (diffusers AttnProcessor protocol +
``dressing_sd/pipelines`` pipeline classes).  Device math: hand-written HIP in ``csrc/`` behind the
C ABI of ``include/imagdressing_hip.h``; this package is the Python host side."""
__version__ = "0.1.0"
