"""``from diffusers.models.lora import LoRALinearLayer`` (/root/reference/adapter/attention_processor.py:7)."""
from imagdressing_amd.adapter.attention_processor import LoRALinearLayer  # noqa: F401
