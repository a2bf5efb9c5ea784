"""Host-side mirror of the EVAL half of the reference's scripts/trainer.py (class `trainer`, tag "LPDM_infer"): the
callers of the hot path.  Same method names, configuration keys, job order, info strings and output directory naming;
the sampling itself is ONE `diffusion_backward` launch per group of jobs instead of one per job.

  _infer_prior_latdiff_from_audio_v1     trainer.py:500-543   infer_gesture: every *.wav of custom_audios
  eval_prior_latdiff_forward_backward_v1 trainer.py:545-1098  edit tasks on the dataset dict of process_loader
      style_Xemo_transfer jobs           trainer.py:559-631   8 jobs
      style_transfer jobs                trainer.py:705-772   8 jobs
      emotion_control jobs               trainer.py:839-901   takes x (own + swapped emotions) = 64 jobs (:919)
      demo_emotion_control               trainer.py:1037-1098 edit_gesture: *_source.wav / *_target.wav -> 2 jobs

What a "job" is: one `self.model.diffusion_backward(bsz, z_con, z_emo, z_sty)` call of the reference.  The reference runs
them one after another, each drawing fresh noise from the device RNG.  Here the jobs of a task are concatenated along the
clip axis and sampled in one launch; job j's clips get the global clip indices the model's clip counter would have
handed out had the calls been made one by one, so the batched result is the sequential one (bitwise up to 128 clips per
launch: one clip per workgroup tile in both cases).  Rendering (Blender / ffmpeg, wav export) is out of scope: outputs
stop at the `*_motion_smplx.npz` files `CaMNVisualizer.animate_ldm_sample_v1/v2` write first (npz_writer.write_sample).

More than one GPU (`main.py --fn infer_gesture | edit_gesture --gpus N`; the reference refuses multi-GPU inference,
models/audio/infer_pretrained_ast_evp.py:45): one process per GPU, every rank builds the SAME job list and takes a contiguous
range of it (shard.job_range; cuts on tile boundaries of the whole launch, clips keep their global indices, tile size and decode
kernels are pinned to what the single-process launch would pick -> the NPZ bytes are the single-process run's).  A rank embeds only
the WAVs its own jobs read and writes only its own NPZs, under the names the single-process run would have given them (the random
file tags are drawn for every job on every rank).  No collective - except the one real exchange of the path: when an edit batch
crosses MANY sources with MANY targets (emotion_control_list.all_pairs), each rank embeds its share of the WAVs and the 3 x 256
floats per WAV are all-gathered (torch.distributed: RCCL on the GPUs, gloo on the CPU).
"""
from __future__ import annotations

import os
import random
import string
import time
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from .infer_ldm import TAKES, mapinfo2takes
from .npz_writer import pack_feats, write_sample

# models/diffusion/viz/visualizer.py:62-67 - the order matters: the FIRST subject whose name occurs in the info string
SUBJECTS = ["wayne", "scott", "solomon", "lawrence", "stewart", "nidal", "zhao", "lu", "zhang", "carlos", "jorge", "itoi",
            "daiki", "jaime", "li", "carla", "sophie", "catherine", "miranda", "kieks", "ayana", "luqi", "hailing", "kexin",
            "goto", "reamey", "yingqing", "tiffnay", "hanieh", "katya"]


def subject_of(info: str) -> str:
    """visualizer.py:303: subject = [x for x in self.subjects if x in attr][0]."""
    return [x for x in SUBJECTS if x in info][0]


def load_wav(path) -> torch.Tensor:
    """torchaudio.load semantics (trainer.py:519): (channels, samples) float32 in [-1, 1), native sample rate - the
    reference does not resample, it feeds whatever rate the file has to a 16 kHz fbank."""
    from scipy.io import wavfile
    _, data = wavfile.read(str(path))
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T))


def _dist_ready() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _emotion_of_take(take_emo: str) -> str:
    """trainer.py:873,877: [k for k, v in train_takes_dict.items() if f"0_{take_emo}_{take_emo}" in v][0]."""
    return [k for k, v in TAKES.items() if f"0_{take_emo}_{take_emo}" in v][0]


def _job(actor, take, z_con, z_emo, z_sty, bsz, info, swap_info=None, audio=None, src_motion=None, **extra) -> dict:
    bsz = int(bsz)
    j = {"actor": actor, "take": take, "bsz": bsz, "z_con": z_con[:bsz], "z_emo": None if z_emo is None else z_emo[:bsz],
         "z_sty": None if z_sty is None else z_sty[:bsz], "info": info, "swap_info": swap_info,
         "audio": None if audio is None else audio[0:bsz * 10000], "src_motion": src_motion}
    j["no_emo"], j["no_sty"], j["remote"] = z_emo is None, z_sty is None, False
    j.update(extra)
    return j


def _remote_job(actor, take, bsz, info, swap_info=None, no_emo=False, no_sty=False, **extra) -> dict:
    """A job another rank samples: what the job LIST needs of it (clip count, token set, names) without its tensors."""
    j = {"actor": actor, "take": take, "bsz": int(bsz), "z_con": None, "z_emo": None, "z_sty": None, "info": info,
         "swap_info": swap_info, "audio": None, "src_motion": None, "no_emo": no_emo, "no_sty": no_sty, "remote": True}
    j.update(extra)
    return j


def job_runs(jobs: List[dict]):
    """The launches of run_jobs: [(token-set key, [job indices])], one per contiguous run of jobs with the same set of condition tokens."""
    groups: Dict[tuple, List[int]] = {}
    for j, job in enumerate(jobs):
        # (the tensors decide; the stored flags stand in for them in jobs built without tensors - remote jobs, plan triples)
        key = (job["no_emo"], job["no_sty"]) if job.get("remote", "z_con" not in job) else (job["z_emo"] is None, job["z_sty"] is None)
        groups.setdefault(key, []).append(j)
    runs = []
    for key, idx in groups.items():
        start = 0
        for k in range(1, len(idx) + 1):
            if k == len(idx) or idx[k] != idx[k - 1] + 1:
                runs.append((key, idx[start:k]))
                start = k
    return runs


def local_jobs(jobs_or_specs, rank: int, world: int) -> List[bool]:
    """Which jobs of the list this rank samples (run by run, shard.job_range on the run's tile size).  Entries: job dicts, or
    (bsz, no_emo, no_sty) triples when the tensors do not exist yet (the plan comes before the audio front-end)."""
    from . import shard
    specs = [j if isinstance(j, dict) else {"bsz": j[0], "no_emo": j[1], "no_sty": j[2]} for j in jobs_or_specs]
    mine = [world == 1] * len(specs)
    if world == 1:
        return mine
    for (no_emo, no_sty), idx in job_runs(specs):
        bszs = [specs[j]["bsz"] for j in idx]
        g = shard.job_plan(sum(bszs), 5 - no_emo - no_sty)["clips_per_group"]
        ja, jb = shard.job_range(bszs, rank, world, align=g)
        for k in range(ja, jb):
            mine[idx[k]] = True
    return mine


def emotion_control_jobs(data: Dict, take_element: str = "first") -> List[dict]:
    """trainer.py:839-901.  For every take of every actor and every `ld_z_emo*` key of that take (its own emotion first,
    then the swapped-in emotions process_loader added, infer_ldm.py:403-410): bsz = min over the latents, truncate, one
    diffusion_backward.  NB the reference truncates z_con / z_sty in place inside the key loop, so a short emotion
    latent shortens every later job of the same take - reproduced."""
    jobs = []
    for actor in data.keys():
        for take in data[actor].keys():
            e = data[actor][take]
            z, z_con, z_sty = e["ld_z"], e["ld_z_con"], e["ld_z_sty"]
            attr, tt = e["ld_attr"], take.split("_")[-1]
            for key in [k for k in e.keys() if "ld_z_emo" in k]:
                z_emo = e[key]
                bsz = min(z.shape[0], z_con.shape[0], z_emo.shape[0], z_sty.shape[0])
                z_con, z_emo, z_sty = z_con[:bsz], z_emo[:bsz], z_sty[:bsz]
                if key == "ld_z_emo":
                    emo = f"original {_emotion_of_take(take.split('_')[-1])}"
                else:
                    emo = f"swap emo {_emotion_of_take(key.split('_')[-1])} in element {take_element}"
                info = f"{attr[0]} {attr[1]} {attr[2]} {attr[4]} yrs {tt} {emo}"
                jobs.append(_job(actor, take, z_con, z_emo, z_sty, bsz, info, "", e.get("ld_wav"), e.get("ld_motion"),
                                 z_emo_key=key))
    return jobs


def _transfer_jobs(eval_set, a1_attr, a2_attr, label, emotion, min_over_z: bool):
    jobs = []
    for actor, take, z, audio, z_con, z_emo, z_sty, swap, src_motion in eval_set:
        act_attr = a1_attr if actor in a1_attr else a2_attr
        tt = take.split("_")[-1]
        rst_info = f"{label} - {act_attr} yrs {tt} {emotion[1:-1]}"
        if swap:
            swap_attr = a1_attr if swap in a1_attr else a2_attr
            swap_info = f"Swapped - {swap_attr} yrs {tt} {emotion[1:-1]}"
            shapes = [ii.shape[0] for ii in (z_con, z_emo, z_sty) if ii is not None]
            bsz = min(shapes + [z.shape[0]]) if min_over_z else min(shapes)
        else:
            swap_info = "Not swapped, original"
            bsz = z.shape[0]
        jobs.append(_job(actor, take, z_con, z_emo, z_sty, bsz, rst_info, swap_info, audio, src_motion))
    return jobs


def style_transfer_jobs(data: Dict, actors: str, emotion: str) -> List[dict]:
    """trainer.py:705-772: actors "[lu-lawrence]", emotion "[angry]" -> 2 takes x (2 originals + 2 swapped) = 8 jobs."""
    a1, a2 = actors[1:-1].split("-")[0], actors[1:-1].split("-")[1]
    t1, t2 = mapinfo2takes(emotion, True)
    at = lambda a: f"{a[0]} {a[1]} {a[2]} {a[4]}"
    a1_attr, a2_attr = at(data[a1][t1]["ld_attr"]), at(data[a2][t2]["ld_attr"])
    row = lambda a, t, ek, sk, swap: (a, t, data[a][t]["ld_z"], data[a][t].get("ld_wav"), data[a][t]["ld_z_con"],
                                      data[a][t][ek], data[a][t][sk], swap, data[a][t].get("ld_motion"))
    es = []
    for t in (t1, t2):
        es += [row(a1, t, "ld_z_emo", "ld_z_sty", ""), row(a2, t, "ld_z_emo", "ld_z_sty", ""),
               row(a1, t, f"ld_z_emo_{a2}", f"ld_z_sty_{a2}", f"{a2}"), row(a2, t, f"ld_z_emo_{a1}", f"ld_z_sty_{a1}", f"{a1}")]
    return _transfer_jobs(es, a1_attr, a2_attr, "Style Transfer", emotion, min_over_z=True)


def style_Xemo_transfer_jobs(data: Dict, actors: str, emotion: str) -> List[dict]:
    """trainer.py:559-631: actors "[scott-lu]", emotion "[happy-angry]" -> 8 jobs (takes from data["takes"])."""
    a1, a2 = actors[1:-1].split("-")[0], actors[1:-1].split("-")[1]
    t1, t2, t3, t4 = data["takes"].split("*")[:4]
    assert t1 == t3 and t2 == t4, f"[LDM EVAL] Takes: {data['takes']} not aligned"
    at = lambda a: f"{a[0]} {a[1]} {a[2]} {a[4]}"
    a1_attr, a2_attr = at(data[a1][t1]["ld_attr"]), at(data[a2][t2]["ld_attr"])

    def row(a, label_take, t, ek="ld_z_emo", sk="ld_z_sty", swap=""):
        e = data[a][t]
        return (a, label_take, e["ld_z"], e.get("ld_wav"), e["ld_z_con"], e[ek], e[sk], swap, e.get("ld_motion"))
    es = [row(a1, t1, t1), row(a2, t1, t3),
          row(a1, t1, t1, f"ld_z_emo_{a2}_{t4}", f"ld_z_sty_{a2}_{t4}", f"{a1}_{t1}_to_{a2}_{t4}"),
          row(a2, t1, t3, f"ld_z_emo_{a1}_{t2}", f"ld_z_sty_{a1}_{t2}", f"{a2}_{t3}_to_{a1}_{t2}"),
          row(a1, t2, t2), row(a2, t2, t4),
          row(a1, t2, t2, f"ld_z_emo_{a2}_{t3}", f"ld_z_sty_{a2}_{t3}", f"{a1}_{t2}_to_{a2}_{t3}"),
          row(a2, t2, t4, f"ld_z_emo_{a1}_{t1}", f"ld_z_sty_{a1}_{t1}", f"{a2}_{t4}_to_{a1}_{t1}")]
    return _transfer_jobs(es, a1_attr, a2_attr, "Style X Emo Transfer", emotion, min_over_z=False)


def run_jobs(model, jobs: List[dict], return_latents: bool = False, batched: bool = True, rank: int = 0, world: int = 1) -> List[Optional[dict]]:
    """Sample every job; returns, per job and in job order, the reference's `rst` entry
    {"feats": (bsz,300,168), "audio", "info"[, "swap_info"]} (trainer.py:884-890) [+ "latents"].
    batched=True: ONE diffusion_backward per group of jobs that share the set of condition tokens (a missing z_emo /
    z_sty drops a token, denoiser.py:159-171, which changes the kernel's tile shape); clip indices are assigned in job
    order exactly as the sequential calls would have drawn them.  batched=False: the reference's call pattern.
    What "the same as the sequential calls" means: the same noise (counter-based, keyed by the global clip index) and the same
    network - bitwise in the fp32 mode up to 128 clips per launch and in the fp32x / bf16 / fp16 modes below 64 clips per launch,
    to rounding otherwise: a launch of more than 128 clips packs several clips per tile (the softmax / PV summation order
    follows a clip's slot, amuse_hip.h amuse_set_clips_per_group), and a launch of 64 clips or more decodes on another kernel
    than the sequential one-clip calls take - bf16 / fp16: the fused per-clip decoder instead of the staged kernels; fp32x: the
    row stages without split-K (k_vae_rows8x) instead of k_vae_rows<f16x2> - the same operands, another summation order.
    A caller that needs the sequential bits pins both: engine.set_clips_per_group(1), set_decode_path("staged").
    world > 1 (one process per GPU, every rank calls this with the SAME list): each launch of the batched form is cut into `world`
    contiguous job ranges (shard.job_range) and this rank samples its range only, with the tile size and decode kernels of the WHOLE
    launch pinned - its results are bitwise the single-process ones; the entries of the other ranks' jobs are None.  Jobs marked
    `remote` (built without tensors, _remote_job) must fall into other ranks' ranges."""
    from . import shard
    out: List[Optional[dict]] = [None] * len(jobs)
    dev = model.device
    c0 = model._clip_counter
    offs = np.concatenate([[0], np.cumsum([j["bsz"] for j in jobs])]).astype(int)

    def finish(j, poses, trans, lat):
        r = {"feats": pack_feats(poses, trans), "audio": jobs[j]["audio"], "info": jobs[j]["info"]}
        if jobs[j].get("swap_info") is not None:
            r["swap_info"] = jobs[j]["swap_info"]
        if return_latents:
            r["latents"] = lat
        out[j] = r

    if not batched:
        mine = local_jobs(jobs, rank, world)
        for j, job in enumerate(jobs):
            if not mine[j]:
                continue
            o = model.diffusion_backward(job["bsz"], job["z_con"], job["z_emo"], job["z_sty"], clip_index0=int(c0 + offs[j]),
                                         return_latents=return_latents)
            finish(j, o["poses"], o["trans"], o.get("latents"))
        model._clip_counter = int(c0 + offs[-1])
        return out
    cat = lambda ts: None if ts[0] is None else torch.cat([torch.as_tensor(t).to(dev, torch.float32) for t in ts])
    eng = getattr(model, "engine", None)
    for (no_emo, no_sty), idx in job_runs(jobs):     # contiguous runs of jobs keep contiguous global clip indices: one launch per run
        bszs = [jobs[j]["bsz"] for j in idx]
        ja, jb = 0, len(idx)
        if world > 1:
            g = shard.job_plan(sum(bszs), 5 - no_emo - no_sty)["clips_per_group"]
            ja, jb = shard.job_range(bszs, rank, world, align=g)
            if jb == ja:
                continue
            if eng is not None:      # the WHOLE launch's tiling and decode kernels, not this shard's
                eng.set_clips_per_group(g)
                eng.set_decode_path(shard.job_plan(sum(bszs))["decode_path"])
        sel = idx[ja:jb]
        assert not any(jobs[j]["remote"] for j in sel), "a job built without its tensors fell into this rank's range"
        con, emo, sty = (cat([jobs[j][k] for j in sel]) for k in ("z_con", "z_emo", "z_sty"))
        pos = np.concatenate([[0], np.cumsum([jobs[j]["bsz"] for j in sel])]).astype(int)
        try:
            o = model.diffusion_backward(int(pos[-1]), con, emo, sty, clip_index0=int(c0 + offs[sel[0]]), return_latents=return_latents)
        finally:
            if world > 1 and eng is not None:
                eng.set_clips_per_group(0)
                eng.set_decode_path("auto")
        for k, j in enumerate(sel):
            s0, s1 = pos[k], pos[k + 1]
            finish(j, o["poses"][s0:s1], o["trans"][s0:s1], o["latents"][s0:s1] if return_latents else None)
    model._clip_counter = int(c0 + offs[-1])
    return out


class trainer:
    """The `trainer(config, device, train_loader=eval_loader, model_path=..., tag="LPDM_infer", model=pretrained_lpdm, ...)`
    object scripts/main.py:218-222 builds, reduced to what the eval entry points read (trainer.py:38-170)."""

    def __init__(self, config, device, train_loader=None, val_loader=None, model_path=None, tag="LPDM_infer",
                 logger_cfg=None, model=None, processed=None, metricsmodel=None, b_path=None, EXEC_ON_CLUSTER=False,
                 debug=False, pretrained_infer=True, batched=True, rank: Optional[int] = None, world: Optional[int] = None,
                 stamp: Optional[str] = None):
        if tag != "LPDM_infer" or not pretrained_infer:
            raise NotImplementedError("amuse_amd.trainer mirrors the evaluation entry points (tag LPDM_infer); training: "
                                      "amuse_amd/train_gesture.py")
        self.config, self.device, self.model, self.train_loader = config, device, model, train_loader
        self.tag, self.debug, self.processed, self.EXEC_ON_CLUSTER = tag, debug, processed, EXEC_ON_CLUSTER
        self.batched = batched
        # one process per GPU (main.py --gpus N -> torch.distributed.run): every rank builds this object over the same configuration
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        ld, test = config["TRAIN_PARAM"]["latent_diffusion"], config["TRAIN_PARAM"]["test"]
        self.smplx_data, self.skip_trans = ld["smplx_data"], ld["skip_trans"]
        self.viz_type = ld.get("viz_type", "CaMN")
        shuffle_type = ld.get("shuffle_type", "actors")
        self.model_path_r = Path(model_path) if model_path is not None else Path(".")
        self.model_dir_name = tag + "_" + datetime.now().strftime("%Y%m%d-%H%M%S") + "_" + shuffle_type
        self.model_path = self.model_path_r / f"{self.model_dir_name}_smplx"
        # the output directory's time stamp: ONE per run - the launcher hands its own to the ranks (AMUSE_RUN_STAMP)
        self.stamp = stamp or os.environ.get("AMUSE_RUN_STAMP") or datetime.now().strftime("%Y%m%d-%H%M%S")
        use = lambda k: bool(test.get(k, {}).get("use", False))
        self.style_transfer, self.emotion_control = use("style_transfer"), use("emotion_control")
        self.style_Xemo_transfer, self.content_control = use("style_Xemo_transfer"), use("content_control")
        overwrite = [test[k].get("overwrite") for k in ("style_transfer", "emotion_control", "style_Xemo_transfer") if use(k)]
        overwrite = [o for o in overwrite if o]
        if overwrite:                                                       # trainer.py:109-116
            print(f"[Trainer LATDIFF EVAL DIR] overwrite model dir name: {overwrite[0]} at {self.stamp}")
            self.model_path = self.model_path_r / overwrite[0]
        if self.style_transfer:
            self.style_transfer_actors, self.style_transfer_emotion = test["style_transfer"]["actors"], test["style_transfer"]["emotion"]
        if self.emotion_control:
            self.emotion_control_actor = test["emotion_control"]["actor"]
            self.emotion_control_take_element = test["emotion_control"]["take_element"]
        if self.content_control:
            raise Exception("Content control not supported yet")            # trainer.py:165
        if self.style_Xemo_transfer:
            self.style_Xemo_transfer_actors = test["style_Xemo_transfer"]["actors"]
            self.style_Xemo_transfer_emotion = test["style_Xemo_transfer"]["emotion"]
        self.demo_emotion_control = use("emotion_control_list")
        self.written: List[Path] = []

    # ------------------------------------------------------------------ visualizer stand-in
    def _animate(self, sample_dict, video_dump, n_clips: int = 1):
        """CaMNVisualizer.animate_ldm_sample_v1 / _v2 up to the NPZ (visualizer.py:298-364).  sample_dict None = a job another rank sampled
        and writes: only the file tags of its `n_clips` files are drawn (6 characters each, npz_writer.write_sample), so that every
        rank names ITS files as the single-process run names them."""
        if sample_dict is None:
            for _ in range(6 * int(n_clips)):
                random.choice(string.ascii_uppercase + string.ascii_lowercase + string.digits)
            return
        self.written += write_sample(sample_dict["feats"], video_dump, subject_of(sample_dict["info"]))

    def _embed(self, path, baseline=False):
        return self._embed_all([path], baseline)[0]

    def _embed_some(self, paths, baseline=False):
        waves = []
        for path in paths:
            a = load_wav(path)
            waves.append(a - a.mean())
        many = getattr(self.model, "process_seq_list", None)
        if many is not None:
            return many(waves, framerate=16000, baseline=baseline)
        return [self.model.process_single_seq(a, framerate=16000, baseline=baseline) for a in waves]

    def _embed_all(self, paths, baseline=False, needed=None, exchange: bool = False):
        """[(con, emo, sty), ...] for a list of WAVs: load, remove the mean (trainer.py:519-521), embed - as one batch where the
        model offers it (PretrainedLPDM_v1.process_seq_list), else call by call.  Row k of a batch is bitwise the single call's.
        needed (more than one rank): the indices this rank's jobs read - only those are embedded, the others come back None.
        exchange (more than one rank, torch.distributed initialised): the list is cut into contiguous shares, each rank embeds its
        share and the 3 x 256 floats per WAV are all-gathered - for job lists in which every rank needs most of the WAVs."""
        n = len(paths)
        if self.world == 1 or (needed is None and not exchange):
            return self._embed_some(paths, baseline)
        if exchange:
            import torch.distributed as dist
            from .shard import shard_range
            assert dist.is_available() and dist.is_initialized(), "exchange=True needs an initialised process group"
            from .shard import all_ranks_ok
            lo, hi = shard_range(n, self.rank, self.world)
            mine, err = None, None
            try:
                mine = self._embed_some(paths[lo:hi], baseline)
            except Exception as e:  # noqa: BLE001 - agreed on below, then raised on EVERY rank: no rank may enter the all_gather alone
                err = e
            if not all_ranks_ok(err is None, torch.device(self.device)):
                raise err if err is not None else RuntimeError("another rank failed to embed its share of the WAVs; not entering the all_gather")
            dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(self.device)
            per = -(-n // self.world)
            buf = torch.zeros(per, 3, 256, dtype=torch.float32, device=dev)
            for k, ces in enumerate(mine):
                buf[k] = torch.stack([t.reshape(256).to(dev, torch.float32) for t in ces])
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(parts, buf)
            out = []
            for r in range(self.world):
                a, b = shard_range(n, r, self.world)
                out += [tuple(parts[r][k, i].reshape(1, 256).to(self.device) for i in range(3)) for k in range(b - a)]
            return out
        idx = sorted(set(needed))
        got = self._embed_some([paths[k] for k in idx], baseline) if idx else []
        out = [None] * n
        for k, e in zip(idx, got):
            out[k] = e
        return out

    # ------------------------------------------------------------------ infer_gesture
    def _infer_prior_latdiff_from_audio_v1(self, baseline, ldm_epoch, audio_list, short_audio_list, modelversion, ammetric):
        start_time = time.time()
        renders = self.config["TRAIN_PARAM"]["baselines"]["renders"]
        task = renders["task"]
        if task != "custom_renders":
            raise NotImplementedError(f"Implement your own task: {task}")
        audios_r, target_path = Path(renders["custom_audios"]), Path(renders["custom_renders"])
        reps = self.config["TRAIN_PARAM"]["test"]["replication_times"]
        for rep_i in range(reps):                                           # seed change
            print(f" <===== INIT: AUDIO LIST LPDM EVALUATION, REP {rep_i + 1}/{reps} =====>")
            if baseline:
                raise Exception("Baseline not implemented")
            # (the reference takes the directory's own order, list(glob), which no two file systems need agree on; sorted here: every
            # rank - and every run - sees ONE order, and a sharded run names and fills its files like the single-process run)
            audios = sorted(audios_r.glob("*.wav"))
            # the reference embeds + samples + renders audio by audio, re-creating `rst` for each (trainer.py:516), so
            # every audio's result lands in <rep>/rst_0 and an NPZ of audio k sits beside those of the audios before
            # it (seq_0/<actor>_seq_0_<rand6>_motion_smplx.npz, told apart only by the random tag).  Same layout
            # here; the embeddings of all audios are computed first and the clips are sampled as one launch.
            # More than one rank: the plan first (which jobs are this rank's), then only this rank's audios through the front-end.
            mine = local_jobs([(1, False, False)] * len(audios), self.rank, self.world)
            embs = self._embed_all(audios, baseline, needed=[k for k, m in enumerate(mine) if m])
            jobs = [_job("scott", a.stem, *embs[k], 1, "scott", None, None, None) if mine[k] else _remote_job("scott", a.stem, 1, "scott")
                    for k, a in enumerate(audios)]
            rst_all = run_jobs(self.model, jobs, batched=self.batched, rank=self.rank, world=self.world)
            video_dump_r = target_path / f"Custom_audios_{self.stamp}_E{ldm_epoch}" / f"rep{rep_i}"
            assert self.viz_type in ["CaMN"], "[LDM EVAL] Invalid viz type: [%s]" % self.viz_type
            for sample_dict in rst_all:
                rst = [sample_dict]
                for i, sd in enumerate(rst):
                    if sd is not None:
                        print(f"VISUALIZATION: LIST AUDIOS {i} =====>")
                    self._animate(sd, video_dump_r / f"rst_{i}")
        print(f"[LDM EVAL] Audio list inference done, total time elapsed: {time.time() - start_time:.4f} s")

    # ------------------------------------------------------------------ edit tasks
    def eval_prior_latdiff_forward_backward_v1(self, baseline, ldm_epoch, audio_list, short_audio_list=False,
                                               modelversion=None, ammetric=False):
        if audio_list:
            self._infer_prior_latdiff_from_audio_v1(baseline, ldm_epoch, audio_list, short_audio_list, modelversion, ammetric)
            return self.written
        test = self.config["TRAIN_PARAM"]["test"]
        metrics_only = self.config["TRAIN_PARAM"].get("motion_extractor", {}).get("metrics_only", False)
        reps = test["replication_times"]
        self.metrics: Dict[str, list] = {}
        for rep_i in range(reps):
            print(f" <===== INIT: LPDM EVALUATION, REP {rep_i + 1}/{reps} =====>")
            eval_data = self.model.process_loader(self.train_loader) if self.train_loader is not None else {}
            if modelversion == "full":
                tasks = []
                if self.style_Xemo_transfer:
                    run_info = f"{self.style_Xemo_transfer_actors[1:-1]}_{self.style_Xemo_transfer_emotion[1:-1]}"
                    tasks.append(("style_Xemo_transfer", run_info, style_Xemo_transfer_jobs(
                        eval_data["style_Xemo_transfer"], self.style_Xemo_transfer_actors, self.style_Xemo_transfer_emotion)))
                if self.style_transfer:
                    run_info = f"{self.style_transfer_actors[1:-1]}_{self.style_transfer_emotion[1:-1]}"
                    tasks.append(("style_transfer", run_info, style_transfer_jobs(
                        eval_data["style_transfer"], self.style_transfer_actors, self.style_transfer_emotion)))
                if self.emotion_control:
                    run_info = f"{self.emotion_control_actor[1:-1]}_{self.emotion_control_take_element}"
                    tasks.append(("emotion_control", run_info, emotion_control_jobs(
                        eval_data["emotion_control"], self.emotion_control_take_element)))
                for name, run_info, jobs in tasks:
                    # (more than one rank: process_loader above ran on every rank - the latents of a take feed jobs of many ranks -
                    # and the sampling is cut here; metrics hold this rank's jobs)
                    rst = run_jobs(self.model, jobs, batched=self.batched, rank=self.rank, world=self.world)
                    self.metrics[name] = [{"actor": j["actor"], "take": j["take"], "swap_info": j["swap_info"] or "",
                                           "rst_info": j["info"], "src_motion": j["src_motion"], "rst_motion": r["feats"],
                                           "audio": j["audio"], "z_con": j["z_con"], "z_emo": j["z_emo"], "z_sty": j["z_sty"]}
                                          for j, r in zip(jobs, rst) if r is not None]
                    if not metrics_only:
                        video_dump_r = self.model_path / "viz" / f"{name}_{run_info}_{self.stamp}_E{ldm_epoch}" / f"rep{rep_i}"
                        assert self.viz_type in ["CaMN"], "[LDM EVAL] Invalid viz type: [%s]" % self.viz_type
                        for i, sample_dict in enumerate(rst):
                            self._animate(sample_dict, video_dump_r / f"rst_{i}", jobs[i]["bsz"])
                        if name == "emotion_control":                       # trainer.py:919
                            n = len(rst) if self.world > 1 else len([f for f in video_dump_r.iterdir() if f.is_dir()])   # (other ranks may still be writing)
                            assert n == 64, "[LDM EVAL] Invalid number of rst dirs: [%d]" % n
            if self.demo_emotion_control:                                   # trainer.py:1037-1075
                ecl = test["emotion_control_list"]
                actor = ecl["actor"]
                print(f"DEMO EMOTION CONTROL EDITS for {actor} =====>")
                audios = sorted(Path(ecl["audios"]).glob("*.wav"))
                srcs = [x for x in audios if "_source" in x.stem]
                tgts = [x for x in audios if "_target" in x.stem]
                target_path = Path(ecl["renders"])
                if not ecl.get("all_pairs"):
                    # the reference: the FIRST source and the FIRST target -> "Original" + "Emotion edited" (trainer.py:1044-1066)
                    src_a, tgt_a = srcs[0], tgts[0]
                    wavs = [src_a, tgt_a]
                    spec = [(f"Original {actor}", 0, 0), (f"Emotion edited {actor}", 0, 1)]                # (info, content/style WAV, emotion WAV)
                else:
                    # extension (BASELINE config 5's shape: 8 sources x 8 targets = 64 jobs): EVERY source edited with the emotion of EVERY target
                    wavs = srcs + tgts
                    spec = [(f"Emotion edited {actor}", i, len(srcs) + j) for i in range(len(srcs)) for j in range(len(tgts))]
                mine = local_jobs([(1, False, False)] * len(spec), self.rank, self.world)
                need = sorted({w for (_, a, b), m in zip(spec, mine) if m for w in (a, b)})
                # many-to-many: every rank needs most WAVs -> embed a share each and all-gather the embeddings (the path's one exchange)
                exchange = bool(ecl.get("all_pairs")) and self.world > 1 and _dist_ready()
                embs = self._embed_all(wavs, baseline, needed=need, exchange=exchange)
                jobs = [_job(actor, wavs[a].stem, embs[a][0], embs[b][1], embs[a][2], 1, info) if m else _remote_job(actor, wavs[a].stem, 1, info)
                        for (info, a, b), m in zip(spec, mine)]
                rst = run_jobs(self.model, jobs, batched=self.batched, rank=self.rank, world=self.world)   # fresh noise per job, like the two calls
                video_dump_r = target_path / f"Custom_audios_{self.stamp}_E{ldm_epoch}" / f"rep{rep_i}"
                assert self.viz_type in ["CaMN"], "[LDM EVAL] Invalid viz type: [%s]" % self.viz_type
                for i, sample_dict in enumerate(rst):
                    if sample_dict is not None:
                        print(f"VISUALIZATION: LIST AUDIOS {i} =====>")
                    self._animate(sample_dict, video_dump_r / f"rst_{i}")
                print(f"END VISUALIZATION: DEMO EMOTION CONTROL {rep_i + 1}/{reps} =====>")
        return self.written
