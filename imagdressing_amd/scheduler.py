"""DDIM scheduler with the surface of diffusers==0.24.0 ``DDIMScheduler`` that the reference uses
(constructed at /root/reference/inference_IMAGdressing.py:119-127; ``set_timesteps``
IMAGDressing_v1_pipeline.py:386, ``scale_model_input`` :486, ``step`` :530, ``add_noise``
..._pipeline_controlnet_inpainting.py:496).  The per-step arithmetic runs in the fused HIP
``ddim_cfg_step`` kernel; this class owns the schedule (host-side scalars, fp32 like diffusers)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 timestep_spacing="leading", **unused):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule!r}")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by IMAGDressing (inference_IMAGdressing.py:124)")
        if prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only epsilon prediction with leading spacing (the reference's inference config)")
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.config = dict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)        # kept on the host: the loop reads them as Python ints

    def scale_model_input(self, sample, timestep=None):
        return sample

    # ---- host-side coefficients for the fused kernel ----
    def alpha(self, t: int) -> float:
        return float(self.alphas_cumprod[int(t)])

    def alpha_prev(self, t: int) -> float:
        prev = int(t) - self.num_train_timesteps // self.num_inference_steps
        return float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)

    # ---- diffusers-compatible tensor API (NCHW in / out), thin over the same kernel ----
    def step(self, model_output, timestep, sample, eta: float = 0.0, return_dict: bool = False, **unused):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by IMAGDressing")
        B, Cc, H, W = sample.shape
        z = sample.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
        e = model_output.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
        ops.ddim_cfg_step(z, torch.cat([e, e]), None, guidance=1.0, a_t=self.alpha(timestep), a_prev=self.alpha_prev(timestep))
        out = z.view(B, H, W, Cc).permute(0, 3, 1, 2).to(sample.dtype)
        return (out,)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alpha(int(torch.as_tensor(timesteps).reshape(-1)[0]))
        return a ** 0.5 * original_samples + (1 - a) ** 0.5 * noise
