"""Namespace of the one name the reference scripts import from here (``diffusers.pipelines.stable_diffusion``)."""
