"""Host-side mirror of the reference's hot-path driver, models/latent_diffusion/infer_ldm.py
(class PretrainedLPDM_v1): same method names, argument meaning, return layout and error behaviour for the
path this library accelerates, with the compute routed through libamuse_hip.so.

What is mirrored                                   reference
  setup(...) -> ldm_epoch                           infer_ldm.py:30-128  (config + checkpoint pick/load + scheduler)
  diffusion_backward(bsz, z_con, z_emo, z_sty)      infer_ldm.py:130-178 (the hot loop, VAE decode, 6D -> axis-angle)
  process_loader(data_dict)                         infer_ldm.py:225-414 (latent swapping for the edit tasks; the
                                                    per-take latents must already be in the dict - see below)
  _loader_helper_v1(motion, audio)                  infer_ldm.py:416-493 (motion half on the HIP path: 300-frame takes,
                                                    axis-angle -> 6D, MotionPrior.encode + rsample; the audio half
                                                    goes through the injected `audio_encoder`)
  process_single_seq(wave)                          infer_ldm.py:180-193 (kaldi fbank + 3 x AST on the HIP path,
                                                    amuse_amd/audio.py, when an AST checkpoint is present; an injected
                                                    `audio_encoder` callable takes precedence)
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import checkpoint as ckpt
from . import scheduler as sch
from .audio import AudioEngine
from .engine import HipEngine


# dm/utils/ldm_evals.py:79-87: the two BEAT takes per emotion the edit tasks draw on
TAKES = {"neutral": ["0_9_9", "0_10_10"], "happy": ["0_65_65", "0_66_66"], "angry": ["0_73_73", "0_74_74"],
         "sad": ["0_81_81", "0_82_82"], "contempt": ["0_87_87", "0_88_88"], "surprise": ["0_95_95", "0_96_96"],
         "fear": ["0_103_103", "0_104_104"], "disgust": ["0_111_111", "0_112_112"]}


def mapinfo2takes(info, trainer=False):
    """infer_ldm.py:519-528."""
    if not trainer:
        info = info.split("_")[1]
    for emo in ("happy", "sad", "angry", "contempt", "disgust", "surprise", "fear"):
        if emo in info:
            return TAKES[emo]
    raise Exception("Unknown emotion: ", info)


def denoiser_variant(arch_denoiser: Optional[dict]):
    """(arch, diffusion_only) of configs/<arch>.json "arch_denoiser" (denoiser.py:20-61), refusing what the kernels are not
    specialised for.  None (a configuration without the section) = the shipped one."""
    if arch_denoiser is None:
        return "trans_enc", False
    a = arch_denoiser
    arch, pose = a.get("arch", "trans_enc"), bool(a.get("diffusion_only", False))
    if arch not in ("trans_enc", "trans_dec"):
        raise ValueError(f"Not supported architechure{arch}!")          # denoiser.py:133
    want = {"ff_size": 512, "num_layers": 9, "num_heads": 4, "normalize_before": False, "activation": "gelu",
            "position_embedding": "learned", "cond_dim": 256, "freq_shift": 0, "pe_type": "mld", "flip_sin_to_cos": True}
    bad = {k: a[k] for k, v in want.items() if k in a and a[k] != v}
    if "latent_dim" in a and a["latent_dim"][-1] != 128:
        bad["latent_dim"] = a["latent_dim"]
    if arch == "trans_enc" and not a.get("ablation_skip_connection", True):
        bad["ablation_skip_connection"] = False      # nn.TransformerEncoder without skips: not built
    if arch == "trans_dec" and a.get("return_intermediate_dec", False):
        bad["return_intermediate_dec"] = True
    if bad:
        raise NotImplementedError(f"the HIP kernels are specialised for configs/diff_latent_v2.json's denoiser; unsupported: {bad}")
    return arch, pose


class PretrainedLPDM_v1:
    def __init__(self, base_prior=None, base_con_ae=None, base_emo_ae=None, base_audio_ae=None,
                 audio_encoder: Optional[Callable] = None):
        self.base_vae = base_prior      # kept for signature compatibility (scripts/main.py:217); unused
        self.arch, self.diffusion_only = "trans_enc", False   # Denoiser variant (setup reads configs/<arch>.json "arch_denoiser")
        self.audio_encoder = audio_encoder
        self.audio_engine: Optional[AudioEngine] = None
        self.engine: Optional[HipEngine] = None
        self.precision = "fp32"         # "fp32" = parity mode, "fp32x" = the same bars on the fp16 MFMA (2.5 x faster), "bf16" / "fp16" = throughput modes
        self.sampler = "ddim"           # "ddim" (the reference's entry point) or "ddpm" (BASELINE configs 2/3)
        self.quat_mode = "p3d"
        self.seed = 2024                # configs/base_new.json TRAIN_PARAM.seed, scripts/main.py:78
        self._clip_counter = 0          # advances like the reference's device RNG does between calls

    # ------------------------------------------------------------------ construction
    def setup(self, config, device, processed, backup_cfg, EXEC_ON_CLUSTER, baseline=False, verbose=False,
              diffonly=False):
        self.config, self.device, self.processed = config, device, Path(processed)
        self.baseline, self.diffonly = baseline, diffonly
        ld = config["TRAIN_PARAM"]["latent_diffusion"]
        self.smplx_rep = ld["smplx_rep"]
        if self.smplx_rep != "6D" or not ld["smplx_data"] or ld["skip_trans"] or ld["train_upper_body"]:
            raise NotImplementedError("the HIP path covers smplx_data + smplx_rep 6D (333 features) only")
        for k in ("style_transfer", "emotion_control", "content_control", "style_Xemo_transfer"):
            setattr(self, k, config["TRAIN_PARAM"]["test"][k]["use"])
        self.seq_len = config["DATA_PARAM"]["Bvh"]["train_pose_framelen"]
        assert self.seq_len == 300, "the decoder kernels are specialised for 300-frame clips (dm/dm.py:91)"
        self.seed = config["TRAIN_PARAM"].get("seed", 2024)
        saved = "saved-models" if not EXEC_ON_CLUSTER else "saved-models-new"
        ep_prior, ep_ldm = ld["pretrained_prior_lpdm_e"], ld["pretrained_ldm_lpdm_e"]
        assert ep_prior == ep_ldm, "Epochs for prior and ldm should be same"
        root = self.processed.parents[1]
        if config.get("_ldm_cfg_override") is not None:      # amuse_amd.main: <arch>.json merged with diff_o.yaml in memory
            self.ldm_cfg = config["_ldm_cfg_override"]
        else:
            with open(root / f"configs/{ld['arch']}.json", "r") as f:
                self.ldm_cfg = json.load(f)
        if backup_cfg is not None:
            raise NotImplementedError("Backup for LPDM not implemented yet!")
        self.arch, self.diffusion_only = denoiser_variant(self.ldm_cfg.get("arch_denoiser"))
        model_dir = root / saved / ld["pretrained_lpdm"]
        lat = ckpt.pick_checkpoint(model_dir, "latdiff", ep_ldm)
        ldm_epoch = ckpt.epoch_of(lat)
        pri = ckpt.pick_checkpoint(model_dir, "prior", ldm_epoch if ep_prior == "best" else ep_prior)
        print("[LDM] <===== Chosen LDM model based on total loss: ", lat, " =====>")
        print("[LATDIFF] <===== Chosen VAE model based on total loss: ", pri, " =====>")
        self._build(ckpt.load_denoiser_checkpoint(lat, self.arch, self.diffusion_only), ckpt.load_prior_checkpoint(pri), device)
        # the audio encoders (infer_ldm.py:111-114 -> Pretrained_AST_EVP.get_model, infer_pretrained_ast_evp.py:12-18):
        # <root>/saved-models/<TRAIN_PARAM[TRAIN_PARAM.tag].pretrained_ast>/*.pt - always under "saved-models", also on
        # the cluster layout.  The reference fails in iterdir() when the directory is missing; an injected
        # `audio_encoder` (precomputed embeddings) is the only case where that is not an error here.
        wd = config["TRAIN_PARAM"].get("wav_dtw_mfcc", {})
        tag = config["TRAIN_PARAM"].get("tag", "latent_diffusion")
        ast_name = config["TRAIN_PARAM"].get(tag, {}).get("pretrained_ast")
        ast_dir = root / "saved-models" / ast_name if ast_name is not None else None
        if ast_dir is not None and ast_dir.is_dir():
            audio_ablation = wd.get("ablation")
            assert audio_ablation is not None, f"[LPDM EVAL] Audio ablation flag: {audio_ablation}"
            best = ckpt.pick_ast_checkpoint(ast_dir, audio_ablation)
            print("[LATDIFF] (2/3) <===== Chosen AST model: ", best, " , loading state dict... =====>")
            sds = ckpt.load_ast_checkpoint(best)
            self.set_audio_encoders(sds["con"], sds["emo"], sds["sty"], wd.get("dataset_mean", -9.173025),
                                    wd.get("dataset_std", 5.062332), wd.get("frame_based_feats", True))
        elif self.audio_encoder is None:
            raise FileNotFoundError(f"[LATDIFF] AST checkpoint directory {ast_dir} not found (TRAIN_PARAM.{tag}.pretrained_ast); "
                                    f"construct PretrainedLPDM_v1(audio_encoder=...) to run from precomputed embeddings")
        return ldm_epoch

    def set_audio_encoders(self, con_sd, emo_sd, sty_sd, norm_mean=-9.173025, norm_std=5.062332, frame_based_feats=True):
        """Build the HIP audio front-end from three ASTModel state dicts (AST_EVP.{con,emo,sty}_enc)."""
        self.audio_engine = AudioEngine(con_sd, emo_sd, sty_sd, self.device, norm_mean, norm_std, frame_based_feats)

    @classmethod
    def from_state_dicts(cls, denoiser_sd: Dict[str, np.ndarray], prior_sd: Optional[Dict[str, np.ndarray]],
                         ldm_cfg: Optional[dict] = None, device="cuda:0", seed: int = 2024, arch: str = "trans_enc",
                         diffusion_only: bool = False):
        """arch / diffusion_only: the Denoiser variant the state dict belongs to (configs/diff_latent_v2.json "arch_denoiser");
        prior_sd may be None for a diffusion_only denoiser, which never decodes."""
        self = cls()
        self.arch, self.diffusion_only = arch, bool(diffusion_only)
        self.ldm_cfg = ldm_cfg or {"scheduler": dict(sch.DEFAULT_SCHED_CFG, set_alpha_to_one=False, steps_offset=1,
                                                     num_inference_timesteps=50, eta=0.0),
                                   "noisy_scheduler": dict(sch.DEFAULT_SCHED_CFG, variance_type="fixed_small",
                                                           clip_sample=False, prediction_type="epsilon")}
        self.seq_len, self.seed, self.smplx_rep, self.diffonly = 300, seed, "6D", bool(diffusion_only)
        for k in ("style_transfer", "emotion_control", "content_control", "style_Xemo_transfer"):
            setattr(self, k, False)
        self._build(denoiser_sd, prior_sd, device)
        return self

    def _build(self, denoiser_sd, prior_sd, device):
        self.device = torch.device(device)
        self.engine = HipEngine(denoiser_sd, prior_sd, self.device, arch=self.arch, diffusion_only=self.diffusion_only)
        ns = self.ldm_cfg.get("noisy_scheduler")
        if ns is not None:   # add_noise coefficients of diffusion_forward follow the loaded DDPMScheduler config (ldm.py:41-49)
            self.engine.set_noisy_scheduler(num_train_timesteps=ns["num_train_timesteps"], beta_start=ns["beta_start"],
                                            beta_end=ns["beta_end"], beta_schedule=ns["beta_schedule"])
        self.num_inference_timesteps = self.ldm_cfg["scheduler"]["num_inference_timesteps"]
        self.eta = self.ldm_cfg["scheduler"]["eta"]
        self.latent_dim = [300, 333] if self.diffusion_only else [1, 128]   # shape of the sampled state (infer_ldm.py:137-141)
        self._tables = {}
        self.set_sampler(self.sampler)

    def set_sampler(self, kind: str, num_inference_steps: Optional[int] = None):
        """"ddim": DDIMScheduler exactly as built at infer_ldm.py:116-123; "ddpm": the ancestral sampler of the
        training-side DDPMScheduler config (ldm.py:41-49), 1000 steps unless strided."""
        key = (kind, num_inference_steps)
        if key not in self._tables:
            self._tables[key] = sch.from_ldm_cfg(self.ldm_cfg, kind, num_inference_steps)
        self.sampler = kind
        self.engine.set_schedule(self._tables[key])

    # ------------------------------------------------------------------ the hot path
    def diffusion_backward(self, bsz, z_con, z_emo, z_sty, x_init=None, step_noise=None, clip_index0=None,
                           return_latents=False):
        """-> {"poses": (B,300,55,3) f32, "trans": (B,300,3) f32} on self.device (infer_ldm.py:130-178) - exactly the
        reference's two keys unless return_latents asks for the final latents too.
        z_emo / z_sty may be None (the token is dropped, denoiser.py:159-171).  Extra keyword arguments
        (explicit noise, global clip index for sharded batches) are extensions; the reference draws the
        initial latent from the device RNG."""
        if self.diffonly and not self.diffusion_only:
            raise  # noqa: PLE0704 - mirrors the bare `raise` at infer_ldm.py:177: a latent denoiser with the decode switched off
        # A diffusion_only denoiser samples the [300][333] feature sequence itself (denoiser.py:64-66,177-187); the reference's own loop
        # cannot run it (its initial noise keeps the latent shape, infer_ldm.py:137-141, and the decode-free branch is a bare
        # `raise`), so this is where the mirror goes beyond it: the sampled features through the 6D -> axis-angle tail of :168-173.
        assert z_con.shape[0] == bsz, f"bsz {bsz} != z_con batch {z_con.shape[0]}"
        c0 = self._clip_counter if clip_index0 is None else clip_index0
        out = self.engine.diffusion_backward(z_con, z_emo, z_sty, self.precision, self.quat_mode, self.seed, c0,
                                             x_init, step_noise)
        if clip_index0 is None:
            self._clip_counter += bsz
        res = {"poses": out["poses"], "trans": out["trans"]}
        if return_latents:
            res["latents"] = out["latents"]
        return res

    def process_single_seq(self, sliced_chunk, framerate=16000 // 2, baseline=False):
        """(con, emo, sty), each (1,256) (infer_ldm.py:180-193): kaldi fbank -> pad / normalise -> 3 x AST on the HIP
        path.  `sliced_chunk` is the (C, n) float waveform the reference hands to kaldi.fbank (channel 0 is used)."""
        if self.audio_encoder is not None:
            con, emo, sty = self.audio_encoder(sliced_chunk)
            return con.reshape(1, -1), emo.reshape(1, -1), sty.reshape(1, -1)
        if self.audio_engine is None:
            raise NotImplementedError("no audio encoders loaded: setup() found no AST checkpoint; call "
                                      "set_audio_encoders(...), pass audio_encoder=..., or feed precomputed embeddings")
        return self.audio_engine.process_single_seq(sliced_chunk, framerate, baseline)

    def process_seq_list(self, chunks, framerate=16000 // 2, baseline=False):
        """process_single_seq for a list of waveforms -> [(con, emo, sty), ...], each (1, 256).  The reference embeds audio by
        audio (trainer.py:516-523); with the HIP front-end loaded the waveforms - of whatever lengths - go through the three
        encoders as ONE batch (0.98 instead of 2.1 ms per clip from 8 clips up; row k bitwise what the single call returns)."""
        if self.audio_engine is None or self.audio_encoder is not None or len(chunks) < 2:
            return [self.process_single_seq(c, framerate, baseline) for c in chunks]
        con, emo, sty = self.audio_engine.features_ragged(chunks)
        return [(con[k:k + 1], emo[k:k + 1], sty[k:k + 1]) for k in range(len(chunks))]

    def motion_to_latent(self, motion, sample: bool = True, clip_index0: Optional[int] = None):
        """The motion half of _loader_helper_v1 (infer_ldm.py:453-465): `motion` (frames, 168) = 55 x 3 SMPL-X
        axis-angle + 3 translation per frame is cut into whole 300-frame takes, converted to the 333 prior features
        (axis-angle -> matrix -> 6D) and pushed through MotionPrior.encode; returns z_motion (takes, 128) =
        Normal(mu, std).rsample() (PretrainedVAE.get_latent, infer_pretrained_vae.py:51-56).  The draw comes from
        the build's counter-based generator (seed, global take index, step 0, stream 2) instead of the device RNG;
        sample=False returns mu."""
        motion = torch.as_tensor(motion)
        if motion.dim() != 2 or motion.shape[1] != 168:
            raise ValueError(f"motion must be (frames, 168), got {tuple(motion.shape)}")
        takes = motion.shape[0] // self.seq_len
        if takes < 1:
            raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) at infer_ldm.py:456
        m = motion[: takes * self.seq_len].reshape(takes, self.seq_len, 168).to(self.device, torch.float32)
        feats = self.engine.smplx_to_feats(m[..., :165].reshape(takes, self.seq_len, 55, 3), m[..., 165:])
        eps = None
        if sample:
            c0 = self._clip_counter if clip_index0 is None else clip_index0
            eps = self.engine.counter_normal(self.seed, c0, takes, 0, 2)
            if clip_index0 is None:
                self._clip_counter += takes
        return self.engine.vae_encode(feats, None, self.precision, eps=eps)["latent"]

    def _loader_helper_v1(self, motion, audio):
        """-> {"z_motion", "z_con", "z_emo", "z_sty"} (infer_ldm.py:416-493).  `audio` is whatever the injected
        audio_encoder accepts; it returns (con, emo, sty), each (takes_audio, 256) or None."""
        if self.audio_encoder is not None:
            con, emo, sty = self.audio_encoder(audio)
        elif self.audio_engine is not None:
            # infer_ldm.py:418-438: audio (C, n) -> n // 160000 chunks.  NB the reference slices audio[:, k:k+160000]
            # - chunk k starts at SAMPLE k, not at k * 160000 - and that is what is reproduced here.
            a = torch.as_tensor(audio)
            chunks = [a[0, k:k + 160000] for k in range(a.shape[1] // 160000)]
            if not chunks:
                raise RuntimeError("stack expects a non-empty TensorList")   # torch.stack([]) at infer_ldm.py:436
            con, emo, sty = self.audio_engine.features(torch.stack(chunks))
        else:
            raise NotImplementedError("no audio encoders loaded: call set_audio_encoders(...), pass audio_encoder=..., "
                                      "or call motion_to_latent for the motion half")
        z = self.motion_to_latent(motion)
        n = z.shape[0]
        return {"z_motion": z, "z_con": con[:n], "z_emo": emo[:n] if emo is not None else None,
                "z_sty": sty[:n] if sty is not None else None}

    def _take_latents(self, entry: dict):
        """What the reference does per take inside process_loader (infer_ldm.py:300-305, 359-364, 395-399):
        `_loader_helper_v1(ld_motion, ld_waveform)` -> ld_z / ld_z_con / ld_z_emo / ld_z_sty.  Takes that already carry
        their latents (precomputed embeddings, no raw data) are left as they are."""
        if "ld_motion" not in entry or "ld_waveform" not in entry:
            missing = [k for k in ("ld_z_con", "ld_z_emo", "ld_z_sty") if k not in entry]
            if missing:
                raise KeyError(f"take has neither (ld_motion, ld_waveform) nor the latents {missing}")
            return
        motion = torch.from_numpy(np.asarray(entry["ld_motion"])) if not isinstance(entry["ld_motion"], torch.Tensor) else entry["ld_motion"]
        z = self._loader_helper_v1(motion, entry["ld_waveform"])
        entry["ld_z"], entry["ld_z_con"], entry["ld_z_emo"], entry["ld_z_sty"] = z["z_motion"], z["z_con"], z["z_emo"], z["z_sty"]

    def process_loader(self, data_dict):
        """Latents + latent swapping of the edit tasks (infer_ldm.py:225-414): style_Xemo_transfer, style_transfer,
        emotion_control, with the same dictionary keys.  skip_trans / train_upper_body variants are refused in setup
        like every non-6D configuration."""
        loader_data = dict()
        if self.style_Xemo_transfer:
            info, data = data_dict["style_Xemo_transfer_info"], data_dict["style_Xemo_transfer"]
            if "," in info:
                raise NotImplementedError("Multiple style transfer not implemented yet")
            # "[lu-lawrence]_[angry-happy]_*lu_angry_0_73_73*lu_happy_0_65_65*lawrence_angry_0_73_73*lawrence_happy_0_65_65*"
            a1, a2 = info.split("_")[0][1:-1].split("-")
            t1, t2, t3, t4 = ("_".join(info.split("*")[k].split("_")[2:]) for k in (1, 2, 3, 4))
            assert all([t1 == t3, t2 == t4]), "Takes are not the same for style transfer!"
            assert data[a1][t1]["ld_emo_label"] == data[a2][t1]["ld_emo_label"], \
                f"Emotion labels are not the same for style transfer! {data[a1][t1]['ld_emo_label']} != {data[a2][t1]['ld_emo_label']}"
            assert data[a1][t2]["ld_emo_label"] == data[a2][t2]["ld_emo_label"], \
                f"Emotion labels are not the same for style transfer! {data[a1][t2]['ld_emo_label']} != {data[a2][t2]['ld_emo_label']}"
            for actor, take in ((a1, t1), (a2, t3), (a1, t2), (a2, t4)):
                self._take_latents(data[actor][take])
            for (xa, xt), (ya, yt) in (((a1, t1), (a2, t4)), ((a2, t3), (a1, t2)), ((a1, t2), (a2, t3)), ((a2, t4), (a1, t1))):
                data[xa][xt][f"ld_z_emo_{ya}_{yt}"] = data[ya][yt]["ld_z_emo"]
                data[xa][xt][f"ld_z_sty_{ya}_{yt}"] = data[ya][yt]["ld_z_sty"]
            data["takes"] = f"{t1}*{t2}*{t3}*{t4}"
            loader_data["style_Xemo_transfer"] = data
        if self.style_transfer:
            data, info = data_dict["style_transfer"], data_dict["style_transfer_info"]
            if "," in info:
                raise NotImplementedError("Multiple style transfer not implemented yet")
            a1, a2 = info.split("_")[0][1:-1].split("-")                       # "[ayana-scott]_[fear]"
            t1, t2 = mapinfo2takes(info)
            assert data[a1][t1]["ld_emo_label"] == data[a2][t1]["ld_emo_label"] == \
                   data[a1][t2]["ld_emo_label"] == data[a2][t2]["ld_emo_label"], \
                   "Emotion labels are not the same for style transfer!"
            for actor, take in ((a1, t1), (a2, t1), (a1, t2), (a2, t2)):
                self._take_latents(data[actor][take])
            for take in (t1, t2):
                # NB the reference stores the partner's EMO latent under "sty" and vice versa (infer_ldm.py:371-381)
                data[a1][take][f"ld_z_sty_{a2}"] = data[a2][take]["ld_z_emo"]
                data[a1][take][f"ld_z_emo_{a2}"] = data[a2][take]["ld_z_sty"]
                data[a2][take][f"ld_z_sty_{a1}"] = data[a1][take]["ld_z_emo"]
                data[a2][take][f"ld_z_emo_{a1}"] = data[a1][take]["ld_z_sty"]
            loader_data["style_transfer"] = data
        if self.emotion_control:
            data, info = data_dict["emotion_control"], data_dict["emotion_control_info"]
            if "," in info:
                raise NotImplementedError("Emotion control with multiple actors or multiple content emotions not implemented yet")
            for actor in data.keys():
                for take in data[actor].keys():
                    self._take_latents(data[actor][take])
            for actor in data.keys():
                for take in data[actor].keys():
                    for other in data[actor].keys():
                        if other != take:
                            data[actor][take][f"ld_z_emo_{other}"] = data[actor][other]["ld_z_emo"]
            loader_data["emotion_control"] = data
        return loader_data
