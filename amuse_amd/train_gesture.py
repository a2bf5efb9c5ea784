"""train_gesture (BASELINE config 4): the data-parallel training step of the latent-prior-diffusion model, reference
scripts/trainer.py:335-498 (`trainer.train_prior_latdiff_forward_backward_v2`) with models/latent_diffusion/ldm.py:71-153
and models/latent_diffusion/utils/latent_losses.py:8-151.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Per iteration and rank
(trainer.py:356-466):
  motion (B,300,168) axis-angle + translation -> 333 features (axis-angle -> matrix -> 6D)                 :362-368
  motion_z, dist_m = prior.encode(features);  feats_rst = prior.decode(motion_z)                           :381-382
  with no_grad: inferred_motion_z = prior.encode(features)        (a second rsample)                       :409-410
  n_set = ldm.diffusion_forward(inferred_motion_z, con, emo, sty) (noise, per-sample timesteps, add_noise)  :411-412
  with no_grad: noise2z = ldm.diffusion_backward(...)  (DDIM-50);  noise2feats = prior.decode(noise2z)      :415-417
  loss = SmoothL1(feats_rst, feats) + 1e-4 KL(dist_m || N(0,1)) + MSE(noise_pred, noise) + SmoothL1(noise2feats, feats)
         (stage "vae_diffusion", train_lpdm v0, configs/diff_latent_v2.json "losses"; recons_joints is off for SMPL-X data,
         trainer.py:174; the vertex-displacement terms need the SMPL-X body model assets and are not built)  :451
  zero_grad; loss.backward(); [DP: ONE all-reduce of the flat fp32 gradient bucket]; AdamW(lr 1e-4).step()  :452-456

What runs where.  The two networks under autograd are plain torch modules (amuse_amd/nn_modules.py - the reference's
networks key for key).  The no-gradient half of the iteration - the in-loop DDIM-50 sampler (51 denoiser passes in the
reference) and the decode of its result - is the inference hot path, so it runs on the HIP kernels (amuse_sample +
amuse_vae_decode) on the weights of the current iteration (amuse_update_weights_device re-packs them on the GPU; `sampler_refresh` > 1
re-packs every n-th iteration only).  Two stated differences there: the HIP sampler has eval semantics (the reference
leaves dropout live in its inner sampler because the modules are in train mode, trainer.py:357-358) and draws its initial
latent from the counter-based generator; the term it feeds, gen_feature, carries no gradient (it is computed under no_grad
in the reference too) - it is logged and added to `total`, nothing else.

Gradient exchange.  All parameters' gradients live in ONE flat fp32 buffer (6,835,661 elements, 27.3 MB; every p.grad is
a view into it), so the data-parallel step is a single all-reduce(SUM) of that buffer followed by a scale by 1 / world.
xGMI is point-to-point (7 links per GPU): one 27 MB collective per iteration is the bucket size that keeps every link busy
with large messages; there is nothing to overlap it with (it needs the complete backward pass).
"""
from __future__ import annotations

import contextlib
import os
import time
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scheduler as sch
from .nn_modules import Denoiser, MotionPrior, load_numpy_state, numpy_state

SEQ_LEN = 300
LOSS_CFG = {"train_lpdm": {"version": "v0"}, "stage": "vae_diffusion", "LAMBDA_PRIOR": 0.0, "LAMBDA_GEN": 1.0,
            "LAMBDA_LATENT": 1.0, "LAMBDA_KL": 1e-4, "LAMBDA_REC": 1.0, "LAMBDA_JOINT": 1.0, "use_recons_joints": False,
            "predict_epsilon": True, "vtex_displacement": False}   # configs/diff_latent_v2.json "losses" + trainer.py:174-175


# ------------------------------------------------------------------ rotations (pytorch3d.transforms, vendored copy
# models/diffusion/utils/rotation_conversions.py:425-478, 41-71, 536-551)
def axis_angle_to_rotation_6d(aa: torch.Tensor) -> torch.Tensor:
    ang = torch.linalg.vector_norm(aa, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    s = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), aa * s], dim=-1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    row0 = torch.stack([1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r)], -1)
    row1 = torch.stack([two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r)], -1)
    return torch.cat([row0, row1], dim=-1)


def motion_to_feats(ld_motion: torch.Tensor) -> torch.Tensor:
    """(B, S, 168) = 55 x 3 axis-angle ++ 3 translation -> (B, S, 333) = 55 x 6D ++ translation   (trainer.py:362-368)."""
    poses, trans = ld_motion[..., :-3], ld_motion[..., -3:]
    rot6 = axis_angle_to_rotation_6d(poses.reshape(*poses.shape[:-1], 55, 3)).reshape(*poses.shape[:-1], 330)
    return torch.cat([rot6, trans], dim=-1)


# ------------------------------------------------------------------ losses (latent_losses.py:8-151, without torchmetrics)
class LatentPriorLosses:
    """Stage / version / lambdas of the loss config select the terms exactly as latent_losses.py:36-98 does; update()
    returns the weighted total of one iteration and accumulates the un-weighted terms, compute() averages them."""

    def __init__(self, cfg: Optional[dict] = None, device="cpu"):
        c = dict(LOSS_CFG)
        c.update((cfg or {}).get("losses", cfg or {}))
        if c.get("vtex_displacement"):
            raise NotImplementedError("vertex-displacement losses need the SMPL-X body models (trainer.py:91-104); not built")
        self.cfg, self.d = c, device
        self.stage, self.version = c["stage"], c["train_lpdm"]["version"]
        losses = []
        if self.stage in ("diffusion", "vae_diffusion"):
            losses.append("inst_loss")
        if self.stage in ("vae", "vae_diffusion"):
            losses += ["recons_feature", "recons_joints", "kl_motion"]
            if self.version == "v0":
                losses += ["gen_feature", "gen_joints"]
            elif self.version == "v1":
                losses.append("latent_feature")
            else:
                raise ValueError(f"train_lpdm_version {self.version} not supported, choose: v0 or v1")
        if self.stage not in ("vae", "diffusion", "vae_diffusion"):
            raise ValueError(f"Stage {self.stage} not supported")
        losses.append("total")
        self.losses = losses
        self._params = {"inst_loss": 1.0, "kl_motion": c["LAMBDA_KL"], "recons_feature": c["LAMBDA_REC"],
                        "recons_joints": c["LAMBDA_JOINT"], "gen_feature": c["LAMBDA_GEN"], "gen_joints": c["LAMBDA_JOINT"],
                        "latent_feature": c["LAMBDA_LATENT"]}
        self.reset()

    def reset(self):
        if getattr(self, "sums", None) is not None:      # in place: a captured training step (GestureTrainer.enable_graph) accumulates into THESE tensors
            for v in self.sums.values():
                v.zero_()
        else:
            self.sums = {k: torch.zeros((), device=self.d) for k in self.losses}
        self.count = 0

    def _update_loss(self, name, outputs, inputs):
        if name == "inst_loss":
            val = F.mse_loss(outputs, inputs)
        elif name == "kl_motion":
            val = torch.distributions.kl_divergence(outputs, inputs).mean()
        else:
            val = F.smooth_l1_loss(outputs, inputs)
        self.sums[name] += val.detach()
        return self._params[name] * val

    def update(self, rs_set, audio_ablation=None):
        total = 0.0
        if self.stage in ("vae", "vae_diffusion"):
            total = total + self._update_loss("recons_feature", rs_set["m_rst"], rs_set["m_ref"])
            if self.cfg["use_recons_joints"]:
                total = total + self._update_loss("recons_joints", rs_set["joints_rst"], rs_set["joints_ref"])
            if self.cfg["LAMBDA_KL"] != 0.0:
                total = total + self._update_loss("kl_motion", rs_set["dist_m"], rs_set["dist_ref"])
        if self.stage in ("diffusion", "vae_diffusion"):
            if not self.cfg["predict_epsilon"]:
                raise NotImplementedError("x-prediction (predict_epsilon False) is not configured by the reference")
            total = total + self._update_loss("inst_loss", rs_set["noise_pred"], rs_set["noise"])
        if self.stage == "vae_diffusion":
            if self.version == "v0":
                if rs_set.get("gen_m_rst") is not None:
                    total = total + self._update_loss("gen_feature", rs_set["gen_m_rst"], rs_set["m_ref"])
            else:
                total = total + self._update_loss("latent_feature", rs_set["lat_rm"], rs_set["lat_m"])
        self.sums["total"] += total.detach()
        self.count += 1
        return total

    def compute(self):
        return {k: (v / max(self.count, 1)) for k, v in self.sums.items()}


# ------------------------------------------------------------------ the ldm wrapper under autograd (ldm.py:16-153)
class LatentDiffusionTrainModule(nn.Module):
    """`allmodels["latent_diffusion"]` for training: owns the Denoiser (state-dict prefix `denoiser.`, which is what
    infer_ldm.py:91-104 strips) and the DDPM noise schedule of add_noise."""

    def __init__(self, ldm_cfg: Optional[dict] = None, dropout: float = 0.1):
        super().__init__()
        self.denoiser = Denoiser(dropout=dropout)
        ns = (ldm_cfg or {}).get("noisy_scheduler", {})
        keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule")
        self.register_buffer("alphas_cumprod", torch.from_numpy(sch.alphas_cumprod(**{k: ns[k] for k in keys if k in ns})),
                             persistent=False)
        self.num_train_timesteps = int(self.alphas_cumprod.numel())

    def diffusion_forward(self, z, ld_audio_con, ld_audio_emo, ld_audio_sty, lengths=None, ld_audio_mfcc=None, noise=None,
                          timesteps=None):
        """z (1, B, 128) -> {"noise", "noise_pred", "noise_prior", "noise_pred_prior"}   (ldm.py:71-115); `noise` /
        `timesteps` may be given (tests, data-parallel equivalence checks), else torch.randn_like / torch.randint."""
        if ld_audio_mfcc is not None:
            raise NotImplementedError("LPDM: Baseline audio AE not implemented yet")
        z = z.permute(1, 0, 2)
        bsz = z.shape[0]
        noise = torch.randn_like(z) if noise is None else noise.to(z)
        if timesteps is None:
            timesteps = torch.randint(0, self.num_train_timesteps, (bsz,), device=z.device)
        timesteps = timesteps.long().to(z.device)
        ac = self.alphas_cumprod.to(z.device)[timesteps]
        noisy = ac.sqrt()[:, None, None] * z.clone() + (1 - ac).sqrt()[:, None, None] * noise   # DDPMScheduler.add_noise
        noise_pred = self.denoiser(noisy, timesteps, ld_audio_con, ld_audio_emo, ld_audio_sty, lengths=lengths)[0]
        return {"noise": noise, "noise_prior": 0, "noise_pred": noise_pred, "noise_pred_prior": 0}


def synthetic_batch(bsz: int, seed: int, device="cpu") -> Dict[str, torch.Tensor]:
    """ld_motion (B,300,168) axis-angle + translation ~ N(0, 0.1), conditions ~ N(0, 1); the keys of
    latdiff_long_collate_fn_v1 (dm/dataload.py:287-308) that the iteration reads."""
    g = torch.Generator().manual_seed(seed)
    return {"ld_motion": (0.1 * torch.randn(bsz, SEQ_LEN, 168, generator=g)).to(device),
            "ld_audio_con": torch.randn(bsz, 256, generator=g).to(device), "ld_audio_emo": torch.randn(bsz, 256, generator=g).to(device),
            "ld_audio_sty": torch.randn(bsz, 256, generator=g).to(device), "ld_attr": [("scott", "male")] * bsz}


class GestureTrainer:
    """The LPDM half of scripts/trainer.py's `trainer` (tag "LPDM"): models {"prior", "ldm"}, LatentPriorLosses, AdamW over
    both parameter lists (trainer.py:176-182), the iteration of :335-466 and the checkpoint writer of :468-496."""

    def __init__(self, prior: MotionPrior, ldm: LatentDiffusionTrainModule, device, lr: float = 1e-4, loss_cfg: Optional[dict] = None,
                 inner_sampler: Optional[Callable] = None, process_group=None, world: int = 1, kind: Optional[str] = None,
                 grads_mode: str = "steal", sampler_stream: bool = True, optimizer: str = "flat", denoiser_stream: bool = True):
        self.model = {"prior": prior.to(device), "ldm": ldm.to(device)}
        # torch.distributions.Normal validates its arguments with blocking device -> host reads (6 per iteration: the host then waits for the previous
        # iteration's backward + optimizer step before it dispatches anything of the next, tools/probes/train_host/sync_points.py).  Off on the GPU
        # unless AMUSE_TRAIN_VALIDATE=1; a non-finite or non-positive std still surfaces in the kl_motion loss the trainer logs.
        if torch.device(device).type == "cuda" and os.environ.get("AMUSE_TRAIN_VALIDATE", "0") != "1":
            prior.validate_args = False
        self.device = torch.device(device)
        self.lpdm_losses = LatentPriorLosses(loss_cfg, self.device)
        self.inner_sampler = inner_sampler      # (con, emo, sty, bsz) -> noise2feats (B,300,333) or None
        self.world, self.pg = world, process_group
        self.kind = kind                        # ablation variant of the LMDB id (trainer.py:393-399): full / emotion / identity
        # the parameter lists of trainer.py:181 (prior, then ldm), each in state-dict order - the order of the flat images the
        # HIP library takes (weights.prior_param_spec / denoiser_param_spec).  Every parameter is a view into ONE flat fp32
        # buffer laid out that way, so the in-loop sampler's re-pack reads its two networks as slices of it (no per-iteration
        # torch.cat of 427 tensors), and every p.grad is a view into a second buffer of the same layout: the gradient bucket.
        from . import weights as wts
        pn, dn = dict(prior.named_parameters()), dict(ldm.denoiser.named_parameters())
        self.params: List[nn.Parameter] = [pn[k] for k in wts.prior_param_spec()] + [dn[k] for k in wts.denoiser_param_spec()]
        assert {id(p) for p in self.params} == {id(p) for m in (prior, ldm) for p in m.parameters()} and len(self.params) == len(pn) + len(dn)
        n = sum(p.numel() for p in self.params)
        self.n_prior = sum(p.numel() for p in pn.values())
        self.flat_param = torch.empty(n, device=self.device, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in self.params:
                v = self.flat_param[off:off + p.numel()].view_as(p)
                v.copy_(p)
                p.data = v
                off += p.numel()
        self.flat_grad = torch.zeros(n, device=self.device, dtype=torch.float32)
        off, self.views = 0, []
        for p in self.params:
            self.views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            p.grad = self.views[-1]
            off += p.numel()
        # parameters the iteration never reaches (denoiser.mem_pos.pe: the trans_enc path uses query_pos only) get no
        # gradient in the reference (grad stays None after zero_grad(set_to_none=True)), so its AdamW never touches them -
        # not even with weight decay.  They stay in the bucket (zeros) but out of the optimizer.
        unused = {id(ldm.denoiser.mem_pos.pe)}
        # ... and so do the projections of a condition token the ablation variant drops (trainer.py:393-399: `emotion` /
        # `baseline` feed no style token, `identity` no emotion token): their gradients stay None in the reference
        dropped = {"emotion": ("emb_proj_sty",), "baseline": ("emb_proj_sty",), "identity": ("emb_proj_emo",)}.get(kind or "", ())
        unused |= {id(p) for n, p in ldm.denoiser.named_parameters() if n.split(".")[0] in dropped}
        # the multi-tensor ("fused") AdamW of torch on the GPU: the same update in a handful of launches instead of ~10 per
        # parameter group of the default foreach path (2.7 ms of device time and 4.8 ms of host time per iteration, section 4.6)
        # ... and on the GPU that update is ONE launch per contiguous run of optimizer parameters in the flat buffers (train_ops.FlatAdamW, csrc/k_train.hip);
        # optimizer = "fused" / "foreach" select torch's own implementations (A/B: tools/gpu_train_variants.py)
        opt_kind = optimizer if self.device.type == "cuda" else "foreach"
        opt_params = [p for p in self.params if id(p) not in unused]
        if opt_kind == "flat":
            from .train_ops import FlatAdamW
            layout, off = [], 0
            for p in self.params:
                layout.append((p, off, p.numel()))
                off += p.numel()
            self.lpdm_opt = FlatAdamW(opt_params, self.flat_param, self.flat_grad, layout, lr=lr)
        else:
            self.lpdm_opt = torch.optim.AdamW(lr=lr, params=opt_params, **({"fused": True} if opt_kind == "fused" else {}))
        # how gradients reach the bucket (grads_mode): "steal" (default) - autograd hands every parameter a fresh gradient, one multi-tensor copy packs
        # them; "sink" - the library's layer calls write their parameter gradients straight into the bucket (train_ops.sink_begin: no AccumulateGrad work, no
        # copy), the few eager parameters accumulate into its zeroed views; "views" - every gradient accumulates into the zeroed views.  All three fill the
        # bucket with the same bits (tests/test_gpu_train_ops.py); on the boxes measured the step is device-bound either way (profiles/r04_train_grad_sink_ab.txt)
        assert grads_mode in ("steal", "sink", "views"), grads_mode
        self.grads_mode = grads_mode
        self.sampler_stream = sampler_stream       # the no-gradient half on a stream of its own beside the forward pass (False: in line; tests)
        self.steal = self.grads_mode == "steal"
        # (eager fallback layers only - AMUSE_TRAIN_FUSED=0 / CPU: the library's own GEMMs need no BLAS)  the step's ~500 fp32 GEMMs are small: rocBLAS's
        # choices run them in 9 ms of device time per iteration where hipBLASLt's heuristics take 12.5, at a third of the host
        # time per call (a process-wide torch setting)
        if self.device.type == "cuda":
            torch.backends.cuda.preferred_blas_library("cublas")
        self._ar_events: list = []
        self._side_stream = None
        self.denoiser_stream = denoiser_stream     # the Denoiser's forward / backward chain on a stream of its own beside the prior's (False: in line; tests)
        self._den_stream = None

    def n_grad_elements(self) -> int:
        return int(self.flat_grad.numel())

    # ------------------------------------------------------------------ one iteration (trainer.py:356-456)
    def forward_losses(self, batch, noise=None, timesteps=None, eps_enc=None, eps_inf=None):
        prior, ldm = self.model["prior"], self.model["ldm"]
        motion = motion_to_feats(batch["ld_motion"].to(self.device, torch.float32))
        lengths = [SEQ_LEN] * motion.shape[0]
        con, emo, sty = batch["ld_audio_con"], batch.get("ld_audio_emo"), batch.get("ld_audio_sty")
        if self.kind in ("emotion", "baseline"):
            sty = None
        elif self.kind == "identity":
            emo = None
        con = con.to(self.device)
        emo = emo.to(self.device) if emo is not None else None
        sty = sty.to(self.device) if sty is not None else None
        # The no-gradient half (in-loop DDIM-50 + decode on the HIP kernels: ~1.7 ms of a latency-bound kernel on 16 of the 256 CUs) depends on the weights and the
        # conditions only: on the GPU it runs on a stream of its own beside the networks' forward pass and is joined in front of the losses.
        gen, side = None, None
        if self.inner_sampler is not None:
            if self.device.type == "cuda" and self.sampler_stream and not getattr(self.inner_sampler, "serial", False):
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(self.device)
                side = self._side_stream
                side.wait_stream(torch.cuda.current_stream(self.device))     # the weights of the last optimizer step, the conditions
                with torch.cuda.stream(side), torch.no_grad():
                    gen = self.inner_sampler(con, emo, sty, motion.shape[0])
        # The Denoiser's chain (the no-gradient encode that feeds it, its forward pass and - autograd runs a node's backward pass on its forward pass's stream - its
        # backward pass) shares nothing with the prior's encode -> decode chain but the inputs: 160-row layers, ~1.4 ms of launches that each leave most of the chip
        # idle.  On the GPU it is issued on a stream of its own (scratch lane 1 of the library, train_ops.register_lane) beside the prior's 9,600-row kernels.
        den = None
        if self.device.type == "cuda" and self.denoiser_stream and _train_ops_enabled():
            from . import train_ops
            if self._den_stream is None:
                self._den_stream = torch.cuda.Stream(self.device)
            den = self._den_stream
            if den.cuda_stream == torch.cuda.current_stream(self.device).cuda_stream:
                den = None                            # (the pool handed this trainer the stream it is running on: nothing to fork onto)
            else:
                train_ops.register_lane(den, 1)       # (every iteration: the one lane-1 stream of the process is this trainer's)
        if den is not None:
            den.wait_stream(torch.cuda.current_stream(self.device))          # motion, the conditions, the weights
        motion_z, dist_m = prior.encode(motion, lengths)
        if eps_enc is not None:                       # explicit rsample draw (tests): z = mu + std * eps
            motion_z = dist_m.loc + dist_m.scale * eps_enc.to(self.device)
        feats_rst = prior.decode(motion_z, lengths)
        dist_ref = torch.distributions.Normal(torch.zeros_like(dist_m.loc), torch.ones_like(dist_m.scale), validate_args=prior.validate_args)
        with (torch.cuda.stream(den) if den is not None else contextlib.nullcontext()):
            with torch.no_grad():
                inferred_z, dist_i = prior.encode(motion, lengths)
                if eps_inf is not None:
                    inferred_z = dist_i.loc + dist_i.scale * eps_inf.to(self.device)
            n_set = ldm.diffusion_forward(inferred_z, con, emo, sty, lengths=lengths, noise=noise, timesteps=timesteps)
        if den is not None:                           # join in front of the losses (the backward pass forks and joins by itself: autograd's stream semantics)
            torch.cuda.current_stream(self.device).wait_stream(den)
            for t in (n_set["noise_pred"], n_set["noise"]):
                t.record_stream(torch.cuda.current_stream(self.device))
        if side is not None:                          # join: the losses read `gen`, and the optimizer step must not overtake the sampler's reads of the weights
            torch.cuda.current_stream(self.device).wait_stream(side)
            gen.record_stream(torch.cuda.current_stream(self.device))
        elif self.inner_sampler is not None:          # inverse diffusion (no gradient): the HIP sampler + decode
            with torch.no_grad():
                gen = self.inner_sampler(con, emo, sty, motion.shape[0])
        rs_set = {"m_ref": motion, "m_rst": feats_rst, "dist_m": dist_m, "dist_ref": dist_ref, "noise_pred": n_set["noise_pred"],
                  "noise": n_set["noise"], "gen_m_rst": gen, "attr": batch.get("ld_attr")}
        return self.lpdm_losses.update(rs_set)

    def train_step(self, batch, **explicit) -> torch.Tensor:
        for m in self.model.values():
            m.train()
        torch.set_grad_enabled(True)
        if self._graph is not None and not explicit and self._graph_takes(batch):
            return self._graphed_step(batch)
        if self._graph is not None:                 # an eager step between replays: the device-side step count is the optimizer's truth while a graph exists
            self.lpdm_opt.sync_step()
        loss = self.forward_losses(batch, **explicit)
        self.backward_into_bucket(loss)
        self.allreduce_gradients()
        self.lpdm_opt.step()
        if self._graph is not None:
            self.lpdm_opt.push_step()
        return loss.detach()

    # ------------------------------------------------------------------ the iteration as two HIP graphs (GPU)
    _graph = None

    def enable_graph(self, batch) -> bool:
        """Capture the iteration at `batch`'s shapes as TWO HIP graphs - forward + losses + backward into the bucket (+ the side-stream sampler), and the optimizer
        step - with the gradient all-reduce between them as an ordinary call; later train_step(batch) calls of those shapes copy the batch into the captured
        buffers and replay (other shapes / explicit draws run eagerly).  What makes a replay a NEW iteration: torch's graph-safe generator for the step's draws
        (noise, timesteps, the two rsamples, the sampler's initial latents), the library's device-side dropout epoch (amuse_train_epoch_advance at the end of the
        second graph) and AdamW's device-side step count (FlatAdamW.step_dev).  Needs the gradient-sink path (the library writes gradients straight into the bucket:
        tools/probes/train_host/graph_capture_probe.py - replays bitwise the eager step) and the flat optimizer.  Call it after a few eager steps (workspaces exist).
        Returns False - and leaves the trainer eager - where the step is not capturable (CPU, another optimizer, a train-mode inner sampler)."""
        from . import _lib as _libmod
        from . import train_ops
        if (self.device.type != "cuda" or not isinstance(self.lpdm_opt, train_ops.FlatAdamW) or getattr(self.inner_sampler, "serial", False) or not train_ops.enabled()
                or getattr(self.inner_sampler, "refresh", 1) != 1):
            return False
        for m in self.model.values():
            m.train()
        torch.set_grad_enabled(True)
        self.grads_mode, self.steal = "sink", False
        if self.inner_sampler is not None:
            self.inner_sampler.graph_safe = True     # initial latents from torch's generator (advances per replay) instead of the host-side clip counter
        keys = [k for k in ("ld_motion", "ld_audio_con", "ld_audio_emo", "ld_audio_sty") if batch.get(k) is not None]
        static = {k: batch[k].to(self.device).clone() for k in keys}
        static["ld_attr"] = batch.get("ld_attr")
        lib = train_ops._st(self.device)["lib"]
        self.lpdm_opt.push_step()
        count0 = self.lpdm_losses.count
        torch.cuda.synchronize(self.device)
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g1):
                loss = self.forward_losses(static)
                self.backward_into_bucket(loss)
                out = loss.detach()
            self.allreduce_gradients()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self.lpdm_opt.step_dev()
                with train_ops._on(self.device):
                    _libmod.check(lib.amuse_train_epoch_advance(1, torch.cuda.current_stream(self.device).cuda_stream))
        except Exception:
            self._graph = None
            raise
        self.lpdm_losses.count = count0              # (a capture records the launches, it does not run them: no iteration happened)
        self._graph = {"g1": g1, "g2": g2, "static": static, "keys": keys, "loss": out}
        return True

    def _graph_takes(self, batch) -> bool:
        st = self._graph["static"]
        return all((batch.get(k) is not None) == (k in self._graph["keys"]) and (batch.get(k) is None or tuple(batch[k].shape) == tuple(st[k].shape))
                   for k in ("ld_motion", "ld_audio_con", "ld_audio_emo", "ld_audio_sty"))

    def _graphed_step(self, batch) -> torch.Tensor:
        g = self._graph
        for k in g["keys"]:
            g["static"][k].copy_(batch[k], non_blocking=True)
        g["g1"].replay()
        self.allreduce_gradients()
        g["g2"].replay()
        self.lpdm_losses.count += 1
        if self.inner_sampler is not None:
            self.inner_sampler.calls += 1
        return g["loss"]

    def backward_into_bucket(self, loss):
        """zero_grad + backward with every gradient ending up in the flat bucket.  With p.grad unset autograd hands each
        parameter its gradient tensor as it is (no 426 read-modify-write adds into zeroed views); one multi-tensor copy then
        packs them, and p.grad points into the bucket again - what allreduce_gradients and the optimizer read."""
        if not self.steal:
            self.flat_grad.zero_()                    # the views stay attached to the bucket; backward accumulates into them
            if self.grads_mode == "sink":
                from . import train_ops
                train_ops.sink_begin(self.flat_param, self.flat_grad)
                try:
                    loss.backward()
                finally:
                    train_ops.sink_end()
                return
            loss.backward()
            return
        for p in self.params:
            p.grad = None
        loss.backward()
        dst = [v for p, v in zip(self.params, self.views) if p.grad is not None]
        src = [p.grad for p in self.params if p.grad is not None]
        torch._foreach_copy_(dst, src)                # parameters backward never reached keep the zeros of their slice
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce_gradients(self):
        if self.world <= 1:
            return
        import torch.distributed as dist
        timed = self.device.type == "cuda"
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)   # the iteration's ONE collective
        self.flat_grad.mul_(1.0 / self.world)
        if timed:
            e1.record()
            self._ar_events.append((e0, e1))
            del self._ar_events[:-256]

    def pop_allreduce_ms(self) -> List[float]:
        """Durations (ms) of the all-reduces since the last call, from HIP events.  Call it AFTER a timed region: it waits for the
        recorded events, and a wait inside the loop would serialise host dispatch and device work of a host-bound step."""
        ev, self._ar_events = self._ar_events, []
        out = []
        for e0, e1 in ev:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        return out

    # ------------------------------------------------------------------ checkpoints (trainer.py:468-496)
    def save_checkpoint(self, model_path, epoch: int, loss_dict: Optional[dict] = None):
        """prior_model_NoOpt_..._e<N>.pt {"epoch", "model_state_dict"} and latdiff_model_wOpt_..._e<N>.pt {"epoch",
        "model_state_dict", "optimizer_state_dict"} with the reference's loss-encoded file names (the readers pick "best"
        by the `total` field, infer_ldm.py:76-81)."""
        ld = {k: float(v) for k, v in (loss_dict or self.lpdm_losses.compute()).items()}
        for k in ("recons_feature", "recons_joints", "kl_motion", "gen_feature", "gen_joints", "inst_loss", "total"):
            ld.setdefault(k, 0.0)
        ld.setdefault("rec_vtex_displacement", 0.0)
        ld.setdefault("gen_vtex_displacement", 0.0)
        tail = "recF{:.4f}_recJ{:.4f}_kl{:.4f}_genF{:.4f}_genJ{:.4f}_instL{:.4f}_vtexR{:.4f}_vtexG{:.4f}_total{:.4f}_e{}.pt".format(
            ld["recons_feature"], ld["recons_joints"], ld["kl_motion"], ld["gen_feature"], ld["gen_joints"], ld["inst_loss"],
            ld["rec_vtex_displacement"], ld["gen_vtex_displacement"], ld["total"], epoch + 1)
        model_path = Path(model_path)
        model_path.mkdir(parents=True, exist_ok=True)
        p1, p2 = model_path / ("prior_model_NoOpt_" + tail), model_path / ("latdiff_model_wOpt_" + tail)
        own = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}   # not views of the 27 MB flat buffer
        torch.save({"epoch": epoch, "model_state_dict": own(self.model["prior"])}, p1)
        torch.save({"epoch": epoch, "model_state_dict": own(self.model["ldm"]),
                    "optimizer_state_dict": self.lpdm_opt.state_dict()}, p2)
        return p1, p2

    # ------------------------------------------------------------------ the epoch loop (trainer.py:353-466)
    def train_prior_latdiff_forward_backward_v2(self, train_loader: Iterable, epochs: int, model_path=None, model_save_freq: int = 200,
                                                rank: int = 0, on_iteration: Optional[Callable] = None):
        iter_start_time = time.time()
        for epoch in range(epochs):
            sampler = getattr(train_loader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)   # a DistributedSampler reshuffles per epoch only when told (the reference's DataLoader(shuffle=True) does by itself)
            for batch in train_loader:
                self.train_step(batch)
                self._steps_done = getattr(self, "_steps_done", 0) + 1
                if getattr(self, "use_graph", False) and self._graph is None and self._steps_done == 3:
                    try:                               # from the fourth iteration on: two graph replays per step instead of ~1,300 launches (enable_graph)
                        ok = self.enable_graph(batch)
                    except Exception as e:  # noqa: BLE001 - training goes on eagerly
                        ok = False
                        print(f"[LPDM-T] HIP-graph capture of the iteration failed ({type(e).__name__}: {str(e)[:160]}): staying eager", flush=True)
                    self.use_graph = ok
                    if rank == 0:
                        print(f"[LPDM-T] iteration {'captured as two HIP graphs around the gradient all-reduce' if ok else 'runs eagerly'}", flush=True)
                if on_iteration is not None:
                    on_iteration(self)
            loss_dict = self.lpdm_losses.compute()
            self.lpdm_losses.reset()
            if rank == 0:
                print(f"[LPDM-T] Epoch: [{epoch + 1}/{epochs}] t: {time.time() - iter_start_time:.4f} s, rec_feat: "
                      f"{float(loss_dict['recons_feature']):.8f}, kl: {float(loss_dict['kl_motion']):.8f}, inst_loss: "
                      f"{float(loss_dict['inst_loss']):.8f}, gen_feature: {float(loss_dict['gen_feature']):.8f}, total: "
                      f"{float(loss_dict['total']):.8f}", flush=True)
                if model_path is not None and (epoch + 1) % model_save_freq == 0:
                    print(f"[LPDM-T] Saving model at epoch {epoch + 1}", flush=True)
                    self.save_checkpoint(model_path, epoch, loss_dict)
        if rank == 0:
            print("[LPDM] Training finished, total time elapsed: %4.4f mins" % ((time.time() - iter_start_time) / 60.0))


class HipInnerSampler:
    """ldm.diffusion_backward (DDIM-50, ldm.py:117-153) + prior.decode of its result on the HIP kernels, on the trainer's
    CURRENT weights.  refresh = n: amuse_update_weights every n-th call (1 = every iteration, the reference's semantics)."""

    def __init__(self, trainer_models: Dict[str, nn.Module], device, precision: str = "bf16", refresh: int = 1, seed: int = 2024,
                 ldm_cfg: Optional[dict] = None, flat: Optional[tuple] = None, rank: int = 0, world: int = 1):
        from .engine import HipEngine
        self.models, self.precision, self.refresh, self.seed = trainer_models, precision, max(1, refresh), seed
        self.flat = flat                        # (prior, denoiser) flat fp32 images that ARE the parameters (GestureTrainer.flat_param)
        self.engine = HipEngine(self._den_state(), self._prior_state(), device)
        self.engine.set_schedule(sch.from_ldm_cfg(ldm_cfg, "ddim") if ldm_cfg and "scheduler" in ldm_cfg else sch.ddim_table())
        self.calls, self.clip_counter, self.sync_ms = 0, 0, []
        self.rank, self.world = rank, world
        # the in-loop decode shares the GPU with the networks' forward pass (side stream): the per-clip kernel keeps it on `bsz` CUs (0.51 ms on 32 of 256) where the staged
        # launches AUTO would take below 64 clips spread over all of them (0.42 ms of 19 chip-wide launches in the forward pass's way)
        self.engine.set_decode_path("fused")
        self.what = {"bf16": 2, "fp32x": 8, "fp16": 16}.get(precision, 1)   # AMUSE_UPD_* mask of the streams this sampler runs
        self.on_device = True      # re-pack on the GPU straight from the trainer's flat parameter buffer (False: the host path of amuse_update_weights; tests)

    def _den_state(self):
        return {k: v.detach().cpu().numpy() for k, v in self.models["ldm"].denoiser.state_dict().items()}

    def _prior_state(self):
        return {k: v.detach().cpu().numpy() for k, v in self.models["prior"].state_dict().items()}

    def __call__(self, con, emo, sty, bsz):
        if self.calls % self.refresh == 0 and self.calls > 0:
            t0 = time.perf_counter()
            if self.on_device:
                # the weights never leave the GPU: the parameters ARE the flat images (or one torch.cat per network), then a gather kernel per packed image
                if self.flat is not None:
                    pri, den = self.flat
                else:
                    from .engine import flatten_on_device
                    from . import weights as wts
                    den = flatten_on_device(self.models["ldm"].denoiser.state_dict(), wts.denoiser_param_spec())
                    pri = flatten_on_device(self.models["prior"].state_dict(), wts.prior_param_spec())
                self.engine.update_weights_device(den, pri, what=self.what)
            else:
                self.engine.update_weights(self._den_state(), self._prior_state(), what=self.what)
            self.sync_ms.append((time.perf_counter() - t0) * 1e3)
        self.calls += 1
        if getattr(self, "graph_safe", False):
            # inside a captured training step (GestureTrainer.enable_graph) the host-side clip counter would be frozen: the initial latents come from torch's
            # generator on the device, whose state the graph advances per replay
            lat = self.engine.sample(con, emo, sty, self.precision, seed=self.seed, x_init=torch.randn(bsz, 128, device=self.engine.device))
            return self.engine.vae_decode(lat, None, self.precision, return_feats=True)["feats"]
        # initial latents are keyed by a global clip index: rank r draws clips [counter + r * bsz, counter + (r + 1) * bsz)
        lat = self.engine.sample(con, emo, sty, self.precision, seed=self.seed, clip_index0=self.clip_counter + self.rank * bsz)
        self.clip_counter += bsz * self.world
        return self.engine.vae_decode(lat, None, self.precision, return_feats=True)["feats"]


def numpy_denoiser_state(ldm) -> dict:
    return {k: v.detach().cpu().numpy() for k, v in ldm.denoiser.state_dict().items()}


class TrainModeInnerSampler:
    """OPT-IN (`--inner-sampler train`, AMUSE_TRAIN_INNER=train): the no-gradient half of an iteration with the reference's own semantics.  The reference calls
    `ldm.diffusion_backward` (DDIM-50, ldm.py:117-153) and `prior.decode` inside its training loop while both networks are in train() mode
    (trainer.py:357-358,413-415): every dropout of the Denoiser's nine encoder layers - attention weights, both residual branches, the FFN activation - and of
    the prior's decoder is LIVE in all 50 denoising steps and in the decode.  HipInnerSampler (the default) runs the persistent eval-mode sampler kernel
    instead: ~10 x faster and, being deterministic, the better estimate of what the networks generate - a stated difference.  This class is the reference's
    form: the trainer's own modules as they are (train mode -> dropout through the library's counter-based masks, train_ops / k_train.hip; eval mode -> none),
    50 x Denoiser.forward + the scheduler row update of amuse_amd/scheduler.py (the arithmetic the sampler kernels apply, tests/test_pins_cpu.py), one decode.
    Runs on the trainer's stream (the layer calls share the rocBLAS handle with the forward pass).  Initial latents: the library's counter-based normals on the
    GPU (the HIP sampler's draw for the same clips), torch.randn from a seeded generator on the CPU."""
    serial = True      # forward_losses: no side stream

    def __init__(self, trainer_models: Dict[str, nn.Module], device, seed: int = 2024, ldm_cfg: Optional[dict] = None, rank: int = 0, world: int = 1,
                 engine=None):
        self.models, self.device, self.seed, self.rank, self.world = trainer_models, torch.device(device), seed, rank, world
        self.table = sch.from_ldm_cfg(ldm_cfg, "ddim") if ldm_cfg and "scheduler" in ldm_cfg else sch.ddim_table()
        self.coef = torch.as_tensor(self.table.coef)            # (T, 8) host copy: the loop reads python floats, no device sync
        self.engine = engine                                    # a HipEngine to draw the initial latents from (GPU), or None
        self.calls, self.clip_counter, self.sync_ms = 0, 0, []

    def initial_latents(self, bsz: int) -> torch.Tensor:
        c0 = self.clip_counter + self.rank * bsz
        self.clip_counter += bsz * self.world
        if self.engine is not None:
            return self.engine.counter_normal(self.seed, c0, bsz, 0, 0)
        g = torch.Generator().manual_seed((self.seed * 1000003 + c0) & 0x7FFFFFFF)
        return torch.randn(bsz, 128, generator=g).to(self.device)

    @staticmethod
    def scheduler_update(row, x, eps, z=None):
        """One schedule row [sb, sa, c0, cx, ce, sigma, clip, 0] (include/amuse_hip.h amuse_schedule) in the kernels' operation order."""
        sb, sa, c0, cx, ce, sg, clipv = (float(v) for v in row[:7])
        x0 = (x - sb * eps) * (1.0 / sa)
        if clipv > 0:
            x0 = x0.clamp(-clipv, clipv)
        nx = c0 * x0
        if cx != 0:
            nx = nx + cx * x
        if ce != 0:
            nx = nx + ce * eps
        if sg != 0:
            nx = nx + sg * z
        return nx

    @torch.no_grad()
    def __call__(self, con, emo, sty, bsz, x_init: Optional[torch.Tensor] = None, return_latents: bool = False):
        den, prior = self.models["ldm"].denoiser, self.models["prior"]
        x = (self.initial_latents(bsz) if x_init is None else x_init.to(self.device)) * self.table.init_noise_sigma
        self.calls += 1
        for i, t in enumerate(self.table.timesteps):
            eps = den(x[:, None], int(t), con, emo, sty)[0][:, 0]
            z = torch.randn_like(x) if float(self.coef[i, 5]) != 0 else None          # (eta > 0 only)
            x = self.scheduler_update(self.coef[i], x, eps, z)
        feats = prior.decode(x[None], [SEQ_LEN] * bsz)
        return (feats, x) if return_latents else feats


def _train_ops_enabled() -> bool:
    from . import train_ops
    return train_ops.enabled()


def ablation_kind(lmdb_id: Optional[str]) -> Optional[str]:
    """The ablation variant the reference derives from the LMDB cache id (trainer.py:396-401): full | emotion | identity | baseline."""
    if not lmdb_id:
        return None
    parts = Path(lmdb_id).name.split("_")        # (Path.name: a trailing slash does not leave an empty last component)
    kind = parts[-3] if len(parts) >= 3 else ""
    if kind == "feat" and len(parts) >= 5:
        kind = parts[-5]
    assert kind in ("full", "emotion", "identity", "baseline"), f"Invalid lmdb_id: {lmdb_id}"
    return kind


def build_trainer(device, rank: int = 0, world: int = 1, process_group=None, seed: int = 0, use_hip_sampler: bool = True,
                  dropout: float = 0.1, sampler_refresh: int = 1, ldm_cfg: Optional[dict] = None, lr: float = 1e-4,
                  kind: Optional[str] = None, inner: Optional[str] = None, grads_mode: str = "steal", sampler_stream: bool = True,
                  optimizer: str = "flat", denoiser_stream: bool = True) -> GestureTrainer:
    """Random-init prior + ldm (the deterministic weights of amuse_amd/weights.py, identical on every rank - what DDP's
    initial broadcast gives the reference's DataParallel-less single-GPU run) and the trainer around them.
    lr = TRAIN_PARAM.latent_diffusion.lr_base (trainer.py:181-184); ldm_cfg = configs/<arch>.json merged with diff_o.yaml (its
    "losses" section weighs the terms, its schedulers drive add_noise and the in-loop sampler); kind = the LMDB id's ablation
    variant (ablation_kind): the condition the variant never feeds keeps its projection out of the optimizer."""
    from . import weights as wts
    prior = load_numpy_state(MotionPrior(dropout=dropout), wts.make_prior_weights(seed))
    ldm = LatentDiffusionTrainModule(ldm_cfg, dropout=dropout)
    load_numpy_state(ldm.denoiser, wts.make_denoiser_weights(seed))
    assert kind in (None, "full", "emotion", "identity", "baseline"), f"Invalid ablation kind: {kind}"
    loss_cfg = None
    if (ldm_cfg or {}).get("losses") is not None:   # trainer.py:175-177: SMPL-X data switches the joints terms off; the vertex terms are not built
        loss_cfg = dict(ldm_cfg["losses"], use_recons_joints=False, vtex_displacement=False)
    tr = GestureTrainer(prior, ldm, device, lr=lr, loss_cfg=loss_cfg, inner_sampler=None, process_group=process_group,
                        world=world, kind=None if kind == "full" else kind, grads_mode=grads_mode, sampler_stream=sampler_stream, optimizer=optimizer, denoiser_stream=denoiser_stream)
    inner = inner or os.environ.get("AMUSE_TRAIN_INNER", "eval")
    if inner not in ("eval", "train"):
        raise ValueError(f"inner sampler {inner!r}: 'eval' (the persistent HIP sampler kernel, default) or 'train' (the reference's train-mode semantics, dropout live)")
    if inner == "train":      # TrainModeInnerSampler: the modules themselves (any device); on the GPU the initial latents are the HIP sampler's draws
        eng = None
        if torch.device(device).type == "cuda":
            from .engine import HipEngine
            eng = HipEngine(numpy_denoiser_state(ldm), numpy_state(prior), device)   # (only amuse_counter_normal is used: the HIP sampler's initial latents)
        tr.inner_sampler = TrainModeInnerSampler(tr.model, device, ldm_cfg=ldm_cfg, rank=rank, world=world, engine=eng)
    elif use_hip_sampler:
        if torch.device(device).type != "cuda":
            raise RuntimeError("the in-loop sampler of train_gesture runs on the HIP kernels: no CPU path (pass use_hip_sampler=False "
                               "to train without the no-gradient gen_feature term)")
        tr.inner_sampler = HipInnerSampler(tr.model, device, refresh=sampler_refresh, ldm_cfg=ldm_cfg,
                                           flat=(tr.flat_param[:tr.n_prior], tr.flat_param[tr.n_prior:]), rank=rank, world=world)
    return tr


def bench_main(args):
    """`python bench.py --config train [--gpus N]`: iterations/s of the data-parallel step at batch 32 per GPU (weak
    scaling: the reference's batch_size 32 is per process), all-reduce ms, HIP weight re-pack ms.  Prints one JSON line."""
    import json
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # (bench.py starts the ranks itself when it was not launched by torch.distributed.run)
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py --config train needs an MI355X")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(1234 + rank)
    tr = build_trainer(dev, rank, world)
    bsz = int(getattr(args, "train_batch", 32) or 32)     # BASELINE config 4: 32 per process (the reference's batch_size); --train-batch: a larger one as a further point
    batches = [synthetic_batch(bsz, 100 * rank + i, dev) for i in range(4)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tr.train_step(batches[i % 4])
    # the iteration as two HIP graphs around the all-reduce (GestureTrainer.enable_graph): the host's ~1,300 launches per step become two replays
    graphed, graph_note = False, "--no-graph"
    if not getattr(args, "no_graph", False):
        try:
            graphed = tr.enable_graph(batches[0])
            graph_note = "captured" if graphed else "not capturable in this configuration: eager"
        except Exception as e:  # noqa: BLE001 - the bench falls back to the eager step and says so
            graphed, graph_note = False, f"capture failed ({type(e).__name__}: {str(e)[:200]}): eager"
        for i in range(3):
            tr.train_step(batches[i % 4])
    barrier()
    tr.pop_allreduce_ms()
    t0 = time.perf_counter()
    for i in range(args.steps):
        tr.train_step(batches[i % 4])
    barrier()
    elapsed = time.perf_counter() - t0
    ar = tr.pop_allreduce_ms()                     # event times, read after the timed region (no host sync inside it)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        its = args.steps / elapsed
        ld = {k: round(float(v), 6) for k, v in tr.lpdm_losses.compute().items()}
        sync = tr.inner_sampler.sync_ms if tr.inner_sampler is not None else []
        print(json.dumps({
            "metric": f"train_gesture iterations/sec (data-parallel step, batch {bsz} per GPU)", "value": round(its, 3), "unit": "it/s",
            "n_gpus": world, "world_size_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"train_gesture (configs/diff_latent_v2.json): prior encode/decode + epsilon loss under autograd "
                                   f"(fp32; transformer layers = one autograd.Function each on the library's layer entry points: HIP glue, fp32 attention and fp32-MFMA GEMM kernels - no vendor BLAS{'' if __import__('amuse_amd.train_ops', fromlist=['x']).enabled() else ' - SWITCHED OFF: eager torch'}), in-loop DDIM-50 sampler + decode on the HIP kernels (bf16), AdamW(1e-4), one flat "
                                   f"all-reduce of {tr.n_grad_elements():,} fp32 gradients; batch {bsz} per GPU, {bsz * world} global; "
                                   f"vertex-displacement loss off (needs SMPL-X assets)",
                       "batch_per_gpu": bsz, "grad_elements": tr.n_grad_elements(),
                       "hip_graph": graphed, "hip_graph_note": graph_note + (" - forward + losses + backward (+ the side-stream sampler) and the optimizer step replayed as two HIP graphs "
                                                                          "around the all-reduce; fresh draws per replay: torch's graph-safe generator, the library's device-side dropout "
                                                                          "epoch and AdamW step count" if graphed else ""),
                       "streams": ("three: the prior's encode -> decode chain, the Denoiser's chain (no-gradient encode, forward, backward; library scratch lane 1), the "
                                   "no-gradient sampler + decode - forked and joined inside the iteration (parallel branches of the graph)") if tr._den_stream is not None
                                  else "the networks' chain + the no-gradient sampler's side stream",
                       "gemm": "own",
                       "gemm_detail": "hand-written HIP (fp32 MFMA) for every GEMM of the networks: the tall projections and input gradients (k_train_gemm_tall), the chunked "
                                      "weight-gradient reductions (k_train_wgrad), and the generic kernel (k_train_gemm_any) for the 333-wide embedding / output layers, the 32-row "
                                      "condition / memory projections and the Denoiser's 160-row layers; attention, LayerNorm / dropout / GELU / bias gradients, AdamW hand-written too; "
                                      "the library links and loads no vendor BLAS (torch's own ops remain only in the loss arithmetic)",
                       "inner_sampler": ("train: the reference's train-mode loop, dropout live (TrainModeInnerSampler)" if getattr(tr.inner_sampler, "serial", False)
                                         else "eval: the persistent HIP sampler kernel (dropout off - the reference's loop runs in train mode; opt in with AMUSE_TRAIN_INNER=train)")},
            "samples_per_s": round(its * bsz * world, 1),
            "allreduce_ms": round(float(np.median(ar)), 3) if ar else None,
            "hip_weight_repack_ms": round(float(np.median(sync)), 3) if sync else None,
            "losses": ld}), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    """`python -m amuse_amd.train_gesture [--gpus N]`: one process per GPU.  Under torch.distributed.run this process is a rank
    (RANK / LOCAL_RANK / WORLD_SIZE); typed directly with --gpus N > 1 it starts the ranks itself (amuse_amd/launch.py)."""
    import argparse
    import sys
    ap = argparse.ArgumentParser(description="train_gesture: the LPDM iteration on the BEAT latent-diffusion cache (--cache, amuse_amd/dataload.py; "
                                             "needs `lmdb` + a pyarrow with `deserialize`) or on synthetic batches of the same shape")
    ap.add_argument("--cache", default=None, help="the LMDB cache directory of dm/dataload.py:113-129 (default: synthetic data)")
    ap.add_argument("--batch", type=int, default=32, help="per process, like the reference's batch_size")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--iters-per-epoch", type=int, default=8)
    ap.add_argument("--out", default=None)
    ap.add_argument("--device", default=None, help="default: cuda:<LOCAL_RANK>")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--seed", type=int, default=2024, help="TRAIN_PARAM.seed (configs/base_new.json)")
    ap.add_argument("--save-freq", type=int, default=1, help="checkpoint every N epochs (TRAIN_PARAM.latent_diffusion.model_save_freq: 200)")
    ap.add_argument("--lr", type=float, default=1e-4, help="TRAIN_PARAM.latent_diffusion.lr_base (AdamW, trainer.py:181-184)")
    ap.add_argument("--kind", default=None, choices=["full", "emotion", "identity", "baseline"],
                    help="ablation variant (default: derived from the --cache id like trainer.py:396-401; synthetic data: full)")
    ap.add_argument("--inner-sampler", default=None, choices=["eval", "train"],
                    help="the no-gradient DDIM-50 + decode of every iteration: eval (default) = the persistent HIP sampler kernel, dropout off; train = the "
                         "reference's semantics (ldm.py:117-153 under model.train(): every dropout live) through the trainer's own modules, ~10 x slower")
    ap.add_argument("--no-graph", action="store_true", help="keep the iteration eager (default on the GPU: captured as two HIP graphs after three eager iterations)")
    ap.add_argument("--ldm-cfg", default=None, help="JSON file: configs/<arch>.json merged with diff_o.yaml (losses, schedulers); default: the shipped values")
    args = ap.parse_args(argv)
    from . import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        return launch.run_ranks("amuse_amd.train_gesture", list(sys.argv[1:] if argv is None else argv), args.gpus, module=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.device and args.device.startswith("cuda:"):
        raise SystemExit(f"--device {args.device} with {world} ranks would put every rank on one GPU: drop the index (rank r uses cuda:<LOCAL_RANK>)")
    device = torch.device(f"cuda:{local_rank}" if (args.device in (None, "cuda") and torch.cuda.is_available()) else (args.device or "cpu"))
    pg = None
    if device.type == "cuda":
        torch.cuda.set_device(device)       # also for ONE rank on cuda:N, N > 0: the library's handles and workspaces key on the current device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if device.type == "cuda":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        pg = dist.group.WORLD
    from .main import fixseed
    fixseed(args.seed + rank)            # scripts/main.py fixseed(TRAIN_PARAM.seed); per-rank offset: ranks draw different noise / timesteps
    ldm_cfg = None
    if args.ldm_cfg:
        import json
        ldm_cfg = json.load(open(args.ldm_cfg))
    kind = args.kind or ablation_kind(args.cache)
    tr = build_trainer(device, rank, world, process_group=pg, use_hip_sampler=device.type == "cuda", ldm_cfg=ldm_cfg, lr=args.lr, kind=kind,
                       inner=args.inner_sampler)
    tr.use_graph = device.type == "cuda" and not args.no_graph
    if rank == 0:
        lc = tr.lpdm_losses.cfg
        print(f"[LPDM-T] lr {args.lr:g}, ablation kind {kind or 'full'}, loss weights " +
              ", ".join(f"{k} {lc[k]}" for k in ("LAMBDA_REC", "LAMBDA_KL", "LAMBDA_LATENT", "LAMBDA_GEN")), flush=True)
        if tr.inner_sampler is None:
            print("[LPDM-T] inner sampler: none (CPU run: the no-gradient gen_feature term is off)", flush=True)
        elif getattr(tr.inner_sampler, "serial", False):
            print("[LPDM-T] inner sampler: train - the reference's semantics (ldm.py:117-153 and prior.decode under model.train(): every dropout live), on the trainer's own modules", flush=True)
        else:
            print("[LPDM-T] inner sampler: eval - DDIM-50 + decode on the persistent HIP sampler kernel with dropout OFF.  DEVIATION from the reference, whose loop runs under "
                  "model.train() with every dropout live (scripts/trainer.py:357-358,413-415): the gen_feature loss term sees un-dropped samples.  "
                  "--inner-sampler train (or AMUSE_TRAIN_INNER=train) selects the reference's semantics, ~10 x slower.", flush=True)
    if args.cache:
        from .dataload import LatentDiffusionCache, make_loader
        loader = make_loader(LatentDiffusionCache.open(args.cache), args.batch, rank=rank, world=world, seed=args.seed)
    else:
        loader = [synthetic_batch(args.batch, 1000 * rank + i, device) for i in range(args.iters_per_epoch)]
    tr.train_prior_latdiff_forward_backward_v2(loader, args.epochs, args.out, model_save_freq=args.save_freq, rank=rank)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
