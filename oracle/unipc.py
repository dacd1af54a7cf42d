"""Restatement of diffusers==0.24.0 ``UniPCMultistepScheduler`` (TEST INFRASTRUCTURE; **parity unpinned**: third-party
arithmetic, un-vendored, no vectors in the reference -- /root/reference/app.py:28 imports it, the supplementary PDF p.1
names UniPC-50 as the paper's sampler).

Written in the LIBRARY'S OWN FORM -- ``convert_model_output`` / ``multistep_uni_p_bh_update`` /
``multistep_uni_c_bh_update`` / ``step`` operating on tensors, with the ``rks`` / ``D1s`` / ``R`` / ``b`` / ``rhos``
construction of Zhao et al. 2023 (UniPC, "bh" variants, data-prediction) -- in float64, so that it is an independent code
path from ``imagdressing_amd/scheduler.py``, which expands every update into a flat list of (coefficient, tensor) terms on
the host and applies it with one fused launch.  Defaults are the library's: solver_order 2, predict_x0, "bh2",
lower_order_final, no disabled correctors, "linspace" timestep spacing, final sigma = that of training timestep 0."""
from __future__ import annotations

import numpy as np
import torch


class UniPCOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2, solver_type="bh2",
                 lower_order_final=True, disable_corrector=()):
        self.T = num_train_timesteps
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2     # scaled_linear
        self.alphas_cumprod = np.cumprod(1.0 - betas)
        self.solver_order, self.solver_type = solver_order, solver_type
        self.lower_order_final, self.disable_corrector = lower_order_final, tuple(disable_corrector)
        self.init_noise_sigma = 1.0
        self.order = solver_order

    def set_timesteps(self, n):
        ac = self.alphas_cumprod
        ts = np.linspace(0, self.T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)        # "linspace"
        sig = ((1 - ac) / ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [((1 - ac[0]) / ac[0]) ** 0.5]])
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = n
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = 1
        return self.timesteps

    def scale_model_input(self, x, t=None):
        return x

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha_t, sigma * alpha_t

    def convert_model_output(self, eps, sample):            # epsilon prediction, predict_x0
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[self.step_index])
        return (sample - sigma_t * eps) / alpha_t

    def _rhos_inputs(self, h, rks, order):
        hh = -h
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        factorial_i = 1
        B_h = hh if self.solver_type == "bh1" else np.expm1(hh)
        R, b = [], []
        for i in range(1, order + 1):
            R.append(np.power(np.array(rks), i - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / factorial_i
        return h_phi_1, B_h, np.stack(R), np.array(b)

    def multistep_uni_p_bh_update(self, sample, order):
        m0 = self.model_outputs[-1]
        x = sample
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[self.step_index + 1])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[self.step_index])
        lambda_t, lambda_s0 = np.log(alpha_t) - np.log(sigma_t), np.log(alpha_s0) - np.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            a_si, s_si = self._alpha_sigma(self.sigmas[self.step_index - i])
            rk = ((np.log(a_si) - np.log(s_si)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        h_phi_1, B_h, R, b = self._rhos_inputs(h, rks, order)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        if D1s:
            rhos_p = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
            pred_res = sum(r * d for r, d in zip(rhos_p, D1s))
        else:
            pred_res = 0.0
        return x_t_ - alpha_t * B_h * pred_res

    def multistep_uni_c_bh_update(self, this_model_output, last_sample, order):
        m0 = self.model_outputs[-1]
        x, model_t = last_sample, this_model_output
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[self.step_index])
        alpha_s0, sigma_s0 = self._alpha_sigma(self.sigmas[self.step_index - 1])
        lambda_t, lambda_s0 = np.log(alpha_t) - np.log(sigma_t), np.log(alpha_s0) - np.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            mi = self.model_outputs[-(i + 1)]
            a_si, s_si = self._alpha_sigma(self.sigmas[self.step_index - (i + 1)])
            rk = ((np.log(a_si) - np.log(s_si)) - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        h_phi_1, B_h, R, b = self._rhos_inputs(h, rks, order)
        rhos_c = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = sum(r * d for r, d in zip(rhos_c[:-1], D1s)) if D1s else 0.0
        D1_t = model_t - m0
        return x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)

    def step(self, eps, t, sample):
        """one call per entry of ``timesteps`` in order; eps / sample: torch tensors; the history is kept in float64, the
        returned sample has the dtype of ``sample`` (so the fp32 oracle UNet can consume it)"""
        out_dtype = sample.dtype
        eps, sample = eps.double(), sample.double()
        use_corrector = (self.step_index > 0 and (self.step_index - 1) not in self.disable_corrector
                         and self.last_sample is not None)
        m = self.convert_model_output(eps, sample)
        if use_corrector:
            sample = self.multistep_uni_c_bh_update(m, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [m]
        this_order = min(self.solver_order, self.num_inference_steps - self.step_index) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self.multistep_uni_p_bh_update(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(out_dtype)

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[int(t)]
        return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise
