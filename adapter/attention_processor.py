"""Shim: same module path as the reference's ``adapter/attention_processor.py``."""
from imagdressing_amd.adapter.attention_processor import *  # noqa: F401,F403
from imagdressing_amd.adapter.attention_processor import (AttnProcessor2_0, BaseSAttnProcessor2_0, CacheAttnProcessor2_0,  # noqa: F401
                                                            CAttnProcessor2_0, IPAttnProcessor2_0, LoRAIPAttnProcessor2_0,
                                                            LoRALinearLayer, LoraRefSAttnProcessor2_0, RefCAttnProcessor2_0,
                                                            RefLoraSAttnProcessor2_0, RefSAttnProcessor2_0, SAttnProcessor2_0)
