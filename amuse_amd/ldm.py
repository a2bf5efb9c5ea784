"""Host-side mirror of the training-time twin of the hot path, models/latent_diffusion/ldm.py (class
LatentDiffusionModel): the same method names, argument meaning and return layout, compute on libamuse_hip.so.

  diffusion_forward(z, ld_audio_con, ld_audio_emo, ld_audio_sty, ...)   ldm.py:71-115  noise + per-sample timesteps,
                                                                        add_noise, denoiser -> {"noise", "noise_pred", ...}
  diffusion_backward(ld_audio_con, ld_audio_emo, ld_audio_sty, ld_audio_mfcc, bsz)
                                                                        ldm.py:117-153 the in-loop DDIM-50 sampler

Forward values only: the reference runs these under autograd with dropout live (the module is in train mode,
scripts/trainer.py); this mirror has eval semantics and no backward pass, so it serves evaluation of the epsilon loss
and the periodic in-training sampling, not optimisation (config 4 is optional)."""
from __future__ import annotations

from typing import Optional

import torch

from .infer_ldm import PretrainedLPDM_v1


class LatentDiffusionModel:
    def __init__(self, lpdm: PretrainedLPDM_v1, predict_epsilon: bool = True, lambda_prior: float = 0.0):
        self.lpdm = lpdm
        self.engine = lpdm.engine
        self.device = lpdm.device
        self.predict_epsilon = predict_epsilon
        self.lambda_prior = lambda_prior
        self.num_train_timesteps = int(getattr(lpdm, "ldm_cfg", {}).get("noisy_scheduler", {}).get("num_train_timesteps", 1000))
        self.generator = torch.Generator().manual_seed(lpdm.seed)

    def diffusion_forward(self, z, ld_audio_con, ld_audio_emo, ld_audio_sty, plot_latent=False, emo_label=None,
                          plot_path=None, attr=None, lengths=None, ld_audio_mfcc=None, noise=None, timesteps=None):
        """z (1, B, 128) as MotionPrior.encode returns it.  `noise` / `timesteps` are extensions: the reference draws
        them from the device RNG (torch.randn_like, torch.randint), here they default to a host generator."""
        if ld_audio_mfcc is not None:
            raise NotImplementedError("LPDM: Baseline audio AE not implemented yet")
        z = z.permute(1, 0, 2)                                           # (B, 1, 128)
        bsz = z.shape[0]
        if noise is None:
            noise = torch.randn(z.shape, generator=self.generator)
        if timesteps is None:
            timesteps = torch.randint(0, self.num_train_timesteps, (bsz,), generator=self.generator)
        timesteps = torch.as_tensor(timesteps).long()
        out = self.engine.diffusion_forward(z[:, 0], torch.as_tensor(noise).reshape(bsz, 128), timesteps.tolist(),
                                            ld_audio_con, ld_audio_emo, ld_audio_sty, self.lpdm.precision)
        noise = torch.as_tensor(noise).reshape(bsz, 1, 128).to(self.device)
        noise_pred = out["noise_pred"][:, None]
        if self.lambda_prior != 0:
            noise_pred, noise_pred_prior = torch.chunk(noise_pred, 2, dim=0)
            noise, noise_prior = torch.chunk(noise, 2, dim=0)
        else:
            noise_pred_prior, noise_prior = 0, 0
        n_set = {"noise": noise, "noise_prior": noise_prior, "noise_pred": noise_pred, "noise_pred_prior": noise_pred_prior}
        if not self.predict_epsilon:
            n_set["pred"] = noise_pred
            n_set["latent"] = z
        return n_set

    def diffusion_backward(self, ld_audio_con, ld_audio_emo, ld_audio_sty, ld_audio_mfcc, bsz, x_init: Optional[torch.Tensor] = None):
        """-> latents (1, B, 128) after the DDIM sampler of the inference path (ldm.py:117-153)."""
        if ld_audio_mfcc is not None:
            raise NotImplementedError("LPDM: Baseline audio AE not implemented yet")
        prev = self.lpdm.sampler
        self.lpdm.set_sampler("ddim")
        try:
            c0 = self.lpdm._clip_counter
            lat = self.engine.sample(ld_audio_con, ld_audio_emo, ld_audio_sty, self.lpdm.precision, seed=self.lpdm.seed,
                                     clip_index0=c0, x_init=x_init)
            self.lpdm._clip_counter += bsz
        finally:
            if prev != "ddim":
                self.lpdm.set_sampler(prev)   # the shared lpdm keeps the sampler its owner chose
        return lat[None]
