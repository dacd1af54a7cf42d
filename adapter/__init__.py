"""Drop-in import path: ``from adapter.attention_processor import RefSAttnProcessor2_0`` etc. resolve to the
MI355X implementation (imagdressing_amd.adapter), so the reference's inference_IMAGdressing*.py import lines
keep working unchanged when this repository root is on sys.path ahead of the reference checkout."""
