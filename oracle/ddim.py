"""Restatement of diffusers==0.24.0 ``DDIMScheduler`` for the configuration every reference
entry script builds (TEST INFRASTRUCTURE; **parity unpinned**, third-party arithmetic).

Config anchor: ``/root/reference/inference_IMAGdressing.py:119-127``
(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, "scaled_linear",
clip_sample=False, set_alpha_to_one=False, steps_offset=1; timestep_spacing defaults to
"leading", prediction_type "epsilon").  Call sites: ``set_timesteps``
IMAGDressing_v1_pipeline.py:386, ``scale_model_input`` :486 (identity for DDIM), ``step`` :530,
``add_noise`` ..._controlnet_inpainting.py:496.
"""
from __future__ import annotations

import numpy as np
import torch


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.T = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]      # set_alpha_to_one=False
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.order = 1
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.T // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)
        return self.timesteps

    def scale_model_input(self, x, t=None):
        return x

    def step(self, eps, t, x, eta=0.0, variance_noise=None):
        """``DDIMScheduler.step`` (diffusers 0.24 scheduling_ddim.py: formula (12) of the DDIM paper).  eta > 0 adds
        std_dev_t * variance_noise with std_dev_t = eta * sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) and shrinks the
        direction term to sqrt(1 - a_prev - std_dev_t^2) * eps; the noise is passed in (the library draws it from a generator)."""
        t = int(t)
        prev_t = t - self.T // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        std = eta * ((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
        if eta > 0:
            assert variance_noise is not None
            prev = prev + std * variance_noise
        return prev

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[int(t)]
        return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise
