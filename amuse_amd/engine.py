"""Thin torch-facing wrapper over the C ABI (include/amuse_hip.h).  PyTorch supplies device memory and
streams only; all compute is in libamuse_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import weights as wts
from .scheduler import ScheduleTable, timestep_freqs

# "fp32x": the fast parity mode of the sampling loop (split-fp16 MFMA operands, include/amuse_hip.h AMUSE_PREC_F32X)
_N_DEN = sum(int(np.prod(s)) for s in wts.denoiser_param_spec().values())   # AMUSE_DENOISER_PARAMS
_N_PRI = sum(int(np.prod(s)) for s in wts.prior_param_spec().values())      # AMUSE_PRIOR_PARAMS
# "fp16": the throughput mode on fp16 operands (the bf16 kernels' speed, an eighth of their drift; AMUSE_PREC_F16)
PREC = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16, "fp32x": _lib.PREC_F32X, "f32x": _lib.PREC_F32X,
        "fp16": _lib.PREC_F16, "f16": _lib.PREC_F16}
QUAT = {"p3d": _lib.QUAT_P3D, "legacy": _lib.QUAT_LEGACY}


def flatten_on_device(sd: Dict[str, torch.Tensor], spec) -> torch.Tensor:
    """The same layout from a module's CUDA state dict, without leaving the device (one torch.cat)."""
    parts = []
    for k, shape in spec.items():
        v = sd[k]
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
        parts.append(v.detach().reshape(-1).to(torch.float32))
    return torch.cat(parts)


def flatten_state_dict(sd: Dict[str, np.ndarray], spec) -> np.ndarray:
    """Concatenate tensors in state-dict order (the layout amuse_create expects)."""
    parts = []
    for k, shape in spec.items():
        if k not in sd:
            raise KeyError(f"key {k} not found in state dict")
        v = sd[k]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
        parts.append(np.ascontiguousarray(v, dtype=np.float32).ravel())
    return np.concatenate(parts)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipEngine:
    """One amuse_ctx on one GPU.  `arch` / `diffusion_only` = the Denoiser variant of configs/diff_latent_v2.json "arch_denoiser"
    (denoiser.py:61,92-131): the shipped ("trans_enc", False) by default; the pose-space variants (diffusion_only) may be built
    without MotionPrior weights (prior_sd None) - they never decode."""

    def __init__(self, denoiser_sd: Dict[str, np.ndarray], prior_sd: Optional[Dict[str, np.ndarray]], device="cuda:0",
                 arch: str = "trans_enc", diffusion_only: bool = False):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AmuseHipError("amuse_amd runs on an MI355X (torch device 'cuda:N'); there is no CPU path")
        self.arch, self.diffusion_only = arch, bool(diffusion_only)
        self.den_spec = wts.denoiser_param_spec(arch, diffusion_only)
        self.state_shape = (300, 333) if diffusion_only else (128,)
        den = flatten_state_dict(denoiser_sd, self.den_spec)
        pri = flatten_state_dict(prior_sd, wts.prior_param_spec()) if prior_sd is not None else None
        fp = C.POINTER(C.c_float)
        torch.cuda.init()
        self.ctx = self.lib.amuse_create_arch(self.device.index or 0, wts.arch_id(arch, diffusion_only), den.ctypes.data_as(fp), den.size,
                                              pri.ctypes.data_as(fp) if pri is not None else None, 0 if pri is None else pri.size)
        if not self.ctx:
            raise _lib.AmuseHipError(f"amuse_create failed: {self.lib.amuse_last_error().decode()}")
        assert self.lib.amuse_state_dim(self.ctx) == int(np.prod(self.state_shape))
        self.schedule: Optional[ScheduleTable] = None
        self.noisy_cfg: Dict[str, object] = {}   # DDPMScheduler config of add_noise (set_noisy_scheduler)

    def update_weights(self, denoiser_sd=None, prior_sd=None, what: int = _lib.UPD_ALL):
        """amuse_update_weights: new state dicts (or pre-flattened float32 arrays in state-dict order) into this context;
        `what` = AMUSE_UPD_* mask (1 fp32 streams, 2 bf16 streams, 8 fp32x streams, 16 fp16 streams, 4 prior-encoder streams too).  The current schedule
        is re-applied after a denoiser update."""
        fp = C.POINTER(C.c_float)

        def flat(sd, spec):
            if sd is None:
                return None
            a = sd if isinstance(sd, np.ndarray) else flatten_state_dict(sd, spec)
            return np.ascontiguousarray(a, dtype=np.float32)
        den, pri = flat(denoiser_sd, self.den_spec), flat(prior_sd, wts.prior_param_spec())
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_update_weights(self.ctx, den.ctypes.data_as(fp) if den is not None else None,
                                                     0 if den is None else den.size,
                                                     pri.ctypes.data_as(fp) if pri is not None else None,
                                                     0 if pri is None else pri.size, int(what), self._stream()))
        if den is not None and self.schedule is not None:
            self.set_schedule(self.schedule)

    def update_weights_device(self, denoiser_flat: Optional[torch.Tensor] = None, prior_flat: Optional[torch.Tensor] = None, what: int = _lib.UPD_ALL):
        """amuse_update_weights_device: the same from flat float32 CUDA tensors in state-dict order (flatten_on_device) - a gather
        kernel per packed image, stream-ordered, no host round trip; the library rebuilds the schedule's time-token table itself."""
        def ptr(t, n):
            if t is None:
                return None
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
                raise ValueError(f"expected a contiguous float32 CUDA tensor of {n} elements")
            return C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_update_weights_device(self.ctx, ptr(denoiser_flat, _N_DEN), ptr(prior_flat, _N_PRI),
                                                            int(what), self._stream()))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.amuse_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, shape=None) -> Optional[torch.Tensor]:
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _cond(self, con, emo, sty):
        """The three condition embeddings as (B,256) device tensors; a wrong rank / feature count raises here like
        the reference's Linear(256, 128) would, instead of sending the kernels out of bounds."""
        con = self._dev(con)
        if con.dim() != 2 or con.shape[1] != 256:
            raise ValueError(f"z_con must be (B, 256), got {tuple(con.shape)}")
        B = con.shape[0]
        emo = self._dev(emo, (B, 256)) if emo is not None else None
        sty = self._dev(sty, (B, 256)) if sty is not None else None
        return con, emo, sty, B

    def set_noisy_scheduler(self, **cfg):
        """Config of the training-side DDPMScheduler (ldm.py:41-49, configs/diff_latent_v2.json "noisy_scheduler") that
        diffusion_forward's add_noise coefficients come from; defaults = scheduler.DEFAULT_SCHED_CFG."""
        self.noisy_cfg = dict(cfg)

    def set_clips_per_group(self, g: int):
        _lib.check(self.lib.amuse_set_clips_per_group(self.ctx, int(g)))

    def set_decode_path(self, path: str = "auto"):
        """decode kernels of the bf16 / fp16 / fp32x modes: "auto" (from 64 clips up the fused per-clip kernel, in fp32x the no-split-K row
        kernel), "staged", "fused", "clip" (the fp32x mode's per-clip decoder; "fused" in the other modes) (amuse_hip.h amuse_set_decode_path)."""
        _lib.check(self.lib.amuse_set_decode_path(self.ctx, {"auto": 0, "staged": 1, "fused": 2, "clip": 3}[path]))

    def set_ablation(self, mask: int = 0):
        """amuse_debug_set_ablation: 1 = the fused kernels run without their S ~ 300 self-attention (timing only: bench.py)."""
        _lib.check(self.lib.amuse_debug_set_ablation(self.ctx, int(mask)))

    def set_schedule(self, table: ScheduleTable):
        ts = np.ascontiguousarray(table.timesteps, dtype=np.int32)
        cf = np.ascontiguousarray(table.coef, dtype=np.float32)
        fr = np.ascontiguousarray(timestep_freqs(), dtype=np.float32)
        s = _lib.Schedule(len(ts), ts.ctypes.data_as(C.POINTER(C.c_int)), cf.ctypes.data_as(C.POINTER(C.c_float)),
                          fr.ctypes.data_as(C.POINTER(C.c_float)))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_set_schedule(self.ctx, C.byref(s), self._stream()))
        self.schedule = table

    def sample(self, con, emo, sty, precision="fp32", seed=0, clip_index0=0, x_init=None, step_noise=None,
               return_traj=False):
        con, emo, sty, B = self._cond(con, emo, sty)
        ss = self.state_shape
        x_init = self._dev(x_init, (B, *ss)) if x_init is not None else None
        T = self.schedule.n_steps
        step_noise = self._dev(step_noise, (T, B, *ss)) if step_noise is not None else None
        lat = torch.empty(B, *ss, device=self.device, dtype=torch.float32)
        traj = torch.empty(T, B, *ss, device=self.device, dtype=torch.float32) if return_traj else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_sample(self.ctx, _ptr(con), _ptr(emo), _ptr(sty), B, PREC[precision], seed,
                                             clip_index0, _ptr(x_init), _ptr(step_noise), _ptr(lat), _ptr(traj),
                                             self._stream()))
        return (lat, traj) if return_traj else lat

    def denoise_step(self, x_t, timestep: int, con, emo, sty, precision="fp32", taps=False, lengths: Optional[Sequence[int]] = None):
        """One Denoiser.forward.  lengths (pose-space variants only): eps rows of frames >= lengths[b] are zeroed (denoiser.py:187,199)."""
        con, emo, sty, B = self._cond(con, emo, sty)
        x_t = self._dev(x_t, (B, *self.state_shape))
        eps = torch.empty(B, *self.state_shape, device=self.device, dtype=torch.float32)
        tap = torch.zeros(11, 16, 128, device=self.device, dtype=torch.float32) if taps else None
        if lengths is not None:
            if not self.diffusion_only:
                raise ValueError("lengths only reach the output of the diffusion_only variants")
            la = np.ascontiguousarray(lengths, dtype=np.int32)
            if la.shape != (B,):
                raise ValueError("lengths must have one entry per clip")
            with torch.cuda.device(self.device):
                _lib.check(self.lib.amuse_denoise_step_pose(self.ctx, _ptr(x_t), int(timestep), _ptr(con), _ptr(emo), _ptr(sty),
                                                            la.ctypes.data_as(C.POINTER(C.c_int)), B, PREC[precision], _ptr(eps), self._stream()))
            return eps
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_denoise_step(self.ctx, _ptr(x_t), int(timestep), _ptr(con), _ptr(emo), _ptr(sty),
                                                   B, PREC[precision], _ptr(eps), _ptr(tap), self._stream()))
        return (eps, tap) if taps else eps

    def diffusion_forward(self, z0, noise, timesteps: Sequence[int], con, emo, sty, precision="fp32"):
        """LatentDiffusionModel.diffusion_forward (ldm.py:71-97), eval semantics: -> {"noisy_latents", "noise_pred"}."""
        from .scheduler import alphas_cumprod
        z0 = self._dev(z0)
        B = z0.shape[0]
        ss = self.state_shape
        if tuple(z0.shape[1:]) != tuple(ss):
            raise ValueError(f"z0 must be (B, {', '.join(map(str, ss))}), got {tuple(z0.shape)}")
        noise = self._dev(noise, (B, *ss))
        con, emo, sty, Bc = self._cond(con, emo, sty)
        if Bc != B:
            raise ValueError(f"z_con has {Bc} rows, z0 has {B}")
        ts = np.ascontiguousarray(timesteps, dtype=np.int32)
        if ts.shape != (B,):
            raise ValueError("timesteps must have one entry per clip")
        ac = alphas_cumprod(**self.noisy_cfg)
        if ts.min() < 0 or ts.max() >= len(ac):
            raise ValueError(f"timesteps must lie in 0..{len(ac) - 1}")
        sa = np.ascontiguousarray(np.sqrt(ac[ts]), dtype=np.float32)
        sb = np.ascontiguousarray(np.sqrt(np.float32(1.0) - ac[ts]), dtype=np.float32)
        out = {"noisy_latents": torch.empty(B, *ss, device=self.device), "noise_pred": torch.empty(B, *ss, device=self.device)}
        fpp = C.POINTER(C.c_float)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_diffusion_forward(self.ctx, _ptr(z0), _ptr(noise), ts.ctypes.data_as(C.POINTER(C.c_int)),
                                                        sa.ctypes.data_as(fpp), sb.ctypes.data_as(fpp), _ptr(con), _ptr(emo),
                                                        _ptr(sty), B, PREC[precision], _ptr(out["noisy_latents"]),
                                                        _ptr(out["noise_pred"]), self._stream()))
        return out

    def vae_decode(self, z, lengths: Optional[Sequence[int]] = None, precision="fp32", quat_mode="p3d",
                   return_feats=False, return_taps=False):
        """return_taps (fused bf16 kernel only, tests): out["taps"] = (11, 300, 128) - clip 0's residual stream after decoder
        blocks 0..8, after decoder.norm and (slot 10) behind block 0's norm1 (amuse_debug_set_decode_tap)."""
        z = self._dev(z)
        B = z.shape[0]
        taps = torch.zeros(11, 300, 128, device=self.device, dtype=torch.float32) if return_taps else None
        feats = torch.empty(B, 300, 333, device=self.device, dtype=torch.float32) if return_feats else None
        poses = torch.empty(B, 300, 55, 3, device=self.device, dtype=torch.float32)
        trans = torch.empty(B, 300, 3, device=self.device, dtype=torch.float32)
        lp = None
        if lengths is not None:
            la = np.ascontiguousarray(lengths, dtype=np.int32)
            if la.shape != (B,):
                raise ValueError("lengths must have one entry per clip")
            lp = la.ctypes.data_as(C.POINTER(C.c_int))
        with torch.cuda.device(self.device):
            if return_taps:
                _lib.check(self.lib.amuse_debug_set_decode_tap(self.ctx, _ptr(taps)))
            try:
                _lib.check(self.lib.amuse_vae_decode(self.ctx, _ptr(z), lp, B, PREC[precision], QUAT[quat_mode],
                                                     _ptr(feats), _ptr(poses), _ptr(trans), self._stream()))
            finally:
                if return_taps:
                    _lib.check(self.lib.amuse_debug_set_decode_tap(self.ctx, None))
        out = {"poses": poses, "trans": trans}
        if return_taps:
            out["taps"] = taps
        if return_feats:
            out["feats"] = feats
        return out

    def vae_encode(self, feats, lengths: Optional[Sequence[int]] = None, precision="fp32", eps=None):
        """MotionPrior.encode (vae.py:154-214): feats (B,300,333) -> {"mu", "std", "latent"} each (B,128);
        latent = mu + std * eps (Normal.rsample with the caller's draw; eps None -> latent = mu)."""
        feats = self._dev(feats, None)
        if feats.dim() != 3 or tuple(feats.shape[1:]) != (300, 333):
            raise ValueError(f"feats must be (B, 300, 333), got {tuple(feats.shape)}")
        B = feats.shape[0]
        eps = self._dev(eps, (B, 128)) if eps is not None else None
        out = {k: torch.empty(B, 128, device=self.device, dtype=torch.float32) for k in ("mu", "std", "latent")}
        lp = None
        if lengths is not None:
            la = np.ascontiguousarray(lengths, dtype=np.int32)
            if la.shape != (B,):
                raise ValueError("lengths must have one entry per clip")
            lp = la.ctypes.data_as(C.POINTER(C.c_int))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_vae_encode(self.ctx, _ptr(feats), lp, B, PREC[precision], _ptr(eps),
                                                 _ptr(out["mu"]), _ptr(out["std"]), _ptr(out["latent"]), self._stream()))
        return out

    def feats_to_smplx(self, feats, quat_mode="p3d"):
        """(B,300,333) features (6D rotations | translation) -> {"poses": (B,300,55,3), "trans": (B,300,3)} (infer_ldm.py:168-173)."""
        feats = self._dev(feats)
        B = feats.shape[0]
        if tuple(feats.shape) != (B, 300, 333):
            raise ValueError(f"feats must be (B, 300, 333), got {tuple(feats.shape)}")
        out = {"poses": torch.empty(B, 300, 55, 3, device=self.device, dtype=torch.float32),
               "trans": torch.empty(B, 300, 3, device=self.device, dtype=torch.float32)}
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_feats_to_smplx(self.ctx, _ptr(feats), B, QUAT[quat_mode], _ptr(out["poses"]), _ptr(out["trans"]), self._stream()))
        return out

    def smplx_to_feats(self, poses, trans):
        """(B,300,55,3) axis-angle + (B,300,3) translation -> (B,300,333) prior features (infer_ldm.py:459-464)."""
        poses = self._dev(poses)
        B = poses.shape[0]
        if tuple(poses.shape) != (B, 300, 55, 3):
            raise ValueError(f"poses must be (B, 300, 55, 3), got {tuple(poses.shape)}")
        trans = self._dev(trans, (B, 300, 3))
        feats = torch.empty(B, 300, 333, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_smplx_to_feats(self.ctx, _ptr(poses), _ptr(trans), B, _ptr(feats), self._stream()))
        return feats

    def diffusion_backward(self, con, emo, sty, precision="fp32", quat_mode="p3d", seed=0, clip_index0=0, x_init=None,
                           step_noise=None, out=None):
        con, emo, sty, B = self._cond(con, emo, sty)
        ss = self.state_shape
        x_init = self._dev(x_init, (B, *ss)) if x_init is not None else None
        T = self.schedule.n_steps
        step_noise = self._dev(step_noise, (T, B, *ss)) if step_noise is not None else None
        if out is None:
            out = {"latents": torch.empty(B, *ss, device=self.device, dtype=torch.float32),
                   "poses": torch.empty(B, 300, 55, 3, device=self.device, dtype=torch.float32),
                   "trans": torch.empty(B, 300, 3, device=self.device, dtype=torch.float32)}
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_diffusion_backward(self.ctx, _ptr(con), _ptr(emo), _ptr(sty), B, PREC[precision],
                                                         QUAT[quat_mode], seed, clip_index0, _ptr(x_init),
                                                         _ptr(step_noise), _ptr(out["latents"]), _ptr(out["poses"]),
                                                         _ptr(out["trans"]), self._stream()))
        return out

    def profile_sample(self, con, emo, sty, precision="bf16", prof_step=1):
        """768 s_memtime stamps of one denoising step of workgroup 0 (amuse_profile_sample): [4 waves][192] from the
        4-wave kernels (fp32), [8 waves][96] flattened into the same buffer from the 8-wave bf16 kernel."""
        con, emo, sty, B = self._cond(con, emo, sty)
        st = torch.zeros(4, 192, device=self.device, dtype=torch.int64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_profile_sample(self.ctx, _ptr(con), _ptr(emo), _ptr(sty), B, PREC[precision],
                                                     int(prof_step), _ptr(st), self._stream()))
        torch.cuda.synchronize(self.device)
        return st.cpu().numpy()

    def counter_normal(self, seed, clip_index0, B, step, rng_stream):
        o = torch.empty(B, *self.state_shape, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_counter_normal(self.ctx, seed, clip_index0, B, step, rng_stream, _ptr(o),
                                                     self._stream()))
        return o
