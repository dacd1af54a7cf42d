"""``from diffusers.pipelines.stable_diffusion import StableDiffusionSafetyChecker`` (/root/reference/inference_IMAGdressing.py:9):
the scripts pass this CLASS (never an instance) as ``safety_checker=`` (:133) and the pipelines never call it."""


class StableDiffusionSafetyChecker:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the IMAGDressing pipelines never instantiate the safety checker (the reference passes the class itself)")
