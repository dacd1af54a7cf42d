"""Shim: same module path as the reference's ``adapter/resampler.py``."""
from imagdressing_amd.adapter.resampler import (FacePerceiverResampler, FeedForward, PerceiverAttention, PerceiverResampler,  # noqa: F401
                                                 ProjPlusModel, Resampler, masked_mean, reshape_tensor)
