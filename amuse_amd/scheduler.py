"""Host-side scheduler tables for the in-kernel update (host/scheduler code stays Python on PyTorch).

Mirrors how the reference drives diffusers 0.17.1 (not vendored; amuse.yml:143):
  * DDIMScheduler(num_train_timesteps, beta_start, beta_end, beta_schedule="scaled_linear",
    set_alpha_to_one, steps_offset) - infer_ldm.py:116-123; clip_sample is NOT passed, so diffusers'
    default clip_sample=True applies; eta from configs/diff_latent_v2.json:57-66
  * DDPMScheduler(..., variance_type="fixed_small", clip_sample=False) - ldm.py:41-49,
    configs/diff_latent_v2.json:48-56 (the reference only calls add_noise on it; the 1000-step
    ancestral sampler of BASELINE configs 2/3 follows diffusers' DDPMScheduler.step)
All coefficient arithmetic is done on float32 torch tensors in the order diffusers uses, so the table
entries carry diffusers' rounding.  Row layout: include/amuse_hip.h (amuse_schedule).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

DEFAULT_SCHED_CFG = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule="scaled_linear"):
    if beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise NotImplementedError(f"{beta_schedule} is not implemented")
    return torch.cumprod(1.0 - betas, dim=0)


def alphas_cumprod(**cfg) -> np.ndarray:
    """float32 alphas_cumprod of the training-side DDPMScheduler (ldm.py:41-49), for add_noise."""
    c = dict(DEFAULT_SCHED_CFG)
    c.update(cfg)
    return _alphas_cumprod(**c).numpy().copy()


def timestep_freqs() -> np.ndarray:
    """exp(-ln(1e4) * k / 128), k < 128, evaluated with torch like embeddings.py:262-267."""
    exponent = -math.log(10000) * torch.arange(0, 128, dtype=torch.float32) / 128
    return torch.exp(exponent).numpy().copy()


@dataclass
class ScheduleTable:
    kind: str
    timesteps: np.ndarray   # int32 [T]
    coef: np.ndarray        # float32 [T, 8]: sb, sa, c0, cx, ce, sigma, clip, 0
    init_noise_sigma: float = 1.0

    @property
    def n_steps(self) -> int:
        return int(len(self.timesteps))

    def needs_noise(self) -> np.ndarray:
        return self.coef[:, 5] != 0


def ddim_table(num_inference_steps=50, steps_offset=1, set_alpha_to_one=False, eta=0.0, clip_sample=True,
               clip_sample_range=1.0, **cfg) -> ScheduleTable:
    cfg = {**DEFAULT_SCHED_CFG, **cfg}
    n_train = cfg["num_train_timesteps"]
    ac = _alphas_cumprod(**cfg)
    final_ac = torch.tensor(1.0) if set_alpha_to_one else ac[0]
    ratio = n_train // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + steps_offset
    if ts.max() >= n_train:
        raise ValueError(f"DDIM timesteps reach {ts.max()} >= num_train_timesteps (alphas_cumprod index out of range)")
    rows = []
    for t in ts:
        prev = int(t) - ratio
        a_t, a_p = ac[int(t)], (ac[prev] if prev >= 0 else final_ac)
        b_t = 1 - a_t
        var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
        std = eta * var ** 0.5
        rows.append([float(b_t ** 0.5), float(a_t ** 0.5), float(a_p ** 0.5), 0.0,
                     float((1 - a_p - std ** 2) ** 0.5), float(std), clip_sample_range if clip_sample else 0.0, 0.0])
    return ScheduleTable("ddim", ts.astype(np.int32), np.asarray(rows, dtype=np.float32))


def ddpm_table(num_inference_steps=None, **cfg) -> ScheduleTable:
    cfg = {**DEFAULT_SCHED_CFG, **cfg}
    n_train = cfg["num_train_timesteps"]
    n_inf = num_inference_steps or n_train
    ac = _alphas_cumprod(**cfg)
    ratio = n_train // n_inf
    ts = (np.arange(0, n_inf) * ratio).round()[::-1].copy().astype(np.int64)
    one = torch.tensor(1.0)
    rows = []
    for t in ts:
        prev = int(t) - ratio
        a_t, a_p = ac[int(t)], (ac[prev] if prev >= 0 else one)
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        c0 = (a_p ** 0.5 * cur_b) / b_t
        cx = cur_a ** 0.5 * b_p / b_t
        sigma = 0.0
        if t > 0:
            sigma = float(torch.clamp((1 - a_p) / (1 - a_t) * cur_b, min=1e-20) ** 0.5)
        rows.append([float(b_t ** 0.5), float(a_t ** 0.5), float(c0), float(cx), 0.0, sigma, 0.0, 0.0])
    return ScheduleTable("ddpm", ts.astype(np.int32), np.asarray(rows, dtype=np.float32))


def from_ldm_cfg(ldm_cfg: dict, kind: str = "ddim", num_inference_steps=None) -> ScheduleTable:
    """Build from configs/diff_latent_v2.json: "scheduler" (DDIM, inference) / "noisy_scheduler" (DDPM)."""
    if kind == "ddim":
        s = ldm_cfg["scheduler"]
        return ddim_table(num_inference_steps or s["num_inference_timesteps"], s["steps_offset"], s["set_alpha_to_one"],
                          s["eta"], num_train_timesteps=s["num_train_timesteps"], beta_start=s["beta_start"],
                          beta_end=s["beta_end"], beta_schedule=s["beta_schedule"])
    s = ldm_cfg["noisy_scheduler"]
    assert s["variance_type"] == "fixed_small" and not s["clip_sample"] and s["prediction_type"] == "epsilon"
    return ddpm_table(num_inference_steps, num_train_timesteps=s["num_train_timesteps"], beta_start=s["beta_start"],
                      beta_end=s["beta_end"], beta_schedule=s["beta_schedule"])
