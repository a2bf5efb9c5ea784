"""Drop-in for the reference's `python main.py --fn {infer_gesture,edit_gesture} [--cfg base_new.json] [--wandb logger.json]`
(scripts/main.py:226-268 -> :113-222 -> trainer.eval_prior_latdiff_forward_backward_v1) for the path this library rebuilds:
WAV files in, SMPL-X NPZ files out, run from a reference-shaped tree.

  cd <amuse>/scripts && python -m amuse_amd.main --fn infer_gesture            # like the reference: dirname = cwd.parent
  python -m amuse_amd.main --fn edit_gesture --root <amuse>                     # or name the tree's root

What is read, exactly as the reference does (scripts/main.py:243-265):
  <root>/configs/base_new.json (or --cfg)          deep-merged with  <root>/scripts/overrides/<fn>.yaml
  <root>/configs/diff_latent_v2.json               deep-merged with  <root>/scripts/overrides/diff_o.yaml
  (prior_emotional_fing.json + prior_o.yaml only carry architecture constants the kernels are specialised for.)
The merge happens IN MEMORY: the reference rewrites the three JSON files on every start (main.py:263-265, and again
:22-26); this entry point never writes into the configuration tree.  The merged diff_latent_v2 is handed to
PretrainedLPDM_v1.setup through `config["_ldm_cfg_override"]`.
From the merged config: TRAIN_PARAM.baselines.renders.{custom_audios, custom_renders} (infer_gesture, trainer.py:507-508),
TRAIN_PARAM.test.emotion_control_list.{audios, renders, actor} (edit_gesture, trainer.py:1039-1043),
TRAIN_PARAM.test.replication_times, TRAIN_PARAM.seed, the checkpoint directories under <root>/saved-models.
The committed config carries the author's absolute paths (/home/kchhatre/...): --audios / --renders override the two
directories without touching the files.  Checkpoints: <root>/saved-models/<pretrained_lpdm>/{latdiff,prior}_*.pt and
<root>/saved-models/<pretrained_ast>/*.pt as the reference expects; --random-init substitutes the deterministic
random-init weights (no checkpoint ships with the reference) so that the path can be exercised end to end.

Outputs: <renders>/Custom_audios_<stamp>_E<epoch>/rep<i>/rst_<k>/seq_<n>/<actor>_seq_<n>_<rand6>_motion_smplx.npz
(trainer.py:534-538, visualizer.py:307-364).  Blender / ffmpeg rendering, wandb and the BEAT dataset objects
(dm.dm, LMDB) are out of scope; dataset-driven edit tasks take the dict `process_loader` receives via --eval-data.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import time
from pathlib import Path

import numpy as np
import torch

from . import audio_weights as aw
from . import weights as wts
from .infer_ldm import PretrainedLPDM_v1
from .trainer import trainer


def fixseed(seed: int):
    """scripts/utils/misc.py:93-101."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def merge_dicts(config: dict, override_dict) -> dict:
    """scripts/main.py:245-256 (recursive, in place on a shallow copy - nested dicts of `config` are updated)."""
    def merge_recursive(original, override):
        for key, value in override.items():
            if isinstance(value, dict) and key in original and isinstance(original[key], dict):
                merge_recursive(original[key], value)
            else:
                original[key] = value
    if override_dict is None:
        return config
    result = config.copy()
    merge_recursive(result, override_dict)
    return result


def load_config(root: Path, fn: str, cfg_path=None):
    """-> (base config, diff_latent_v2 config), each merged with its override YAML, nothing written back."""
    import yaml
    out = []
    for override, cfg in ((f"{fn}.yaml", cfg_path or root / "configs" / "base_new.json"),
                          ("diff_o.yaml", root / "configs" / "diff_latent_v2.json")):
        config_dict = json.load(open(cfg))
        ov = root / "scripts" / "overrides" / override
        override_dict = yaml.safe_load(open(ov)) if ov.exists() else None
        out.append(merge_dicts(config_dict, override_dict))
    return out[0], out[1]


def train_gesture_entry(args, dirname: Path, config: dict):
    """--fn train_gesture (scripts/main.py:116-155): the LPDM trainer on the latent-diffusion LMDB cache named by the
    configuration - batch size, epochs, save frequency and cache path from base_new.json + scripts/overrides/train_gesture.yaml -
    through amuse_amd.train_gesture's own entry point (one process per GPU with --gpus N)."""
    from . import train_gesture
    tp = config["TRAIN_PARAM"]
    assert not tp["pretrained_infer"], f"Arg: train_gesture and pretrained_infer: {tp['pretrained_infer']} mismatch!"   # main.py:126
    assert tp["motion_extractor"]["use"] is False, "Motion extractor should be False!"
    ld = tp["latent_diffusion"]
    assert ld.get("smplx_data", True), "smplx_data must be True!"                                                        # main.py:129
    # what the configuration asks for beyond the defaults (trainer.py:94-104,177-184): refuse what this path cannot do instead of
    # silently training another objective
    if ld.get("optimizer_name", "adamw") != "adamw":
        raise SystemExit(f"train_gesture: optimizer_name {ld.get('optimizer_name')!r}: the reference's LPDM trainer builds AdamW only (trainer.py:184)")
    if ld.get("vtex_displacement", False) and not args.skip_vtex_loss:
        raise SystemExit("train_gesture: TRAIN_PARAM.latent_diffusion.vtex_displacement is True (scripts/overrides/train_gesture.yaml:25): the "
                         "rec / gen vertex-displacement loss terms need the licensed SMPL-X body models (latent_losses.py:173-250), which this "
                         "path does not have.  Pass --skip-vtex-loss to train WITHOUT those two terms (checkpoint names then read vtexR0.0000 / "
                         "vtexG0.0000), or set vtex_displacement: False")
    if ld.get("vtex_displacement", False):
        print("[LPDM-T] WARNING: vtex_displacement is True in the configuration but --skip-vtex-loss drops both vertex-displacement terms", flush=True)
    ldm_cfg = config.get("_ldm_cfg")
    argv = ["--batch", str(ld.get("batch_size", 32)), "--epochs", str(args.epochs or ld.get("n_epochs", 12000)),
            "--save-freq", str(ld.get("model_save_freq", 200)), "--seed", str(tp.get("seed", 2024)), "--gpus", str(args.gpus),
            "--out", str(dirname / "saved-models"), "--lr", repr(float(ld.get("lr_base", 1e-4)))]
    tmp_cfg = None
    if ldm_cfg is not None:   # configs/<arch>.json merged with diff_o.yaml: loss weights and both schedulers (the ranks may be other processes)
        import tempfile
        f = tempfile.NamedTemporaryFile("w", suffix="_ldm_cfg.json", delete=False)
        json.dump(ldm_cfg, f)
        f.close()
        tmp_cfg = f.name
        argv += ["--ldm-cfg", tmp_cfg]
    cache = dirname / "data" / "BEAT-processed" / tp.get("diffusion", {}).get("lmdb_cache", "")
    if not args.synthetic:
        if not (cache.is_dir() and tp.get("diffusion", {}).get("lmdb_cache")):
            raise SystemExit(f"train_gesture: the LMDB cache {cache} does not exist (prepare_data is the reference's job; --synthetic trains "
                             f"on random batches of the collate function's shapes)")
        argv += ["--cache", str(cache)]     # (the ablation variant is derived from this id, trainer.py:396-401)
    elif tp.get("diffusion", {}).get("lmdb_cache"):
        from .train_gesture import ablation_kind
        argv += ["--kind", ablation_kind(tp["diffusion"]["lmdb_cache"])]
    if args.device != "cuda:0":
        argv += ["--device", args.device]
    if args.iters_per_epoch:
        argv += ["--iters-per-epoch", str(args.iters_per_epoch)]
    print(f"Experiment init: AMUSE, fn: train_gesture, time: {time.asctime()}")
    try:
        return train_gesture.main(argv)
    finally:
        if tmp_cfg is not None:       # (the ranks have finished reading it: main returns after they exit)
            try:
                os.unlink(tmp_cfg)
            except OSError:
                pass


def _launch_ranks(args, argv, dirname: Path):
    """`--fn infer_gesture | edit_gesture --gpus N` typed directly: this process never touches the GPU; it starts N ranks of this very module as
    CHILD processes (amuse_amd/launch.py: torch.distributed.run, never exec), hands them ONE time stamp for the output directory and a scratch
    directory for their manifests, and returns the paths all ranks wrote (rank order = job order)."""
    import sys
    import tempfile
    from datetime import datetime
    from . import launch
    av = list(sys.argv[1:] if argv is None else argv)
    if args.root is None:                      # the ranks start in another process: name the tree explicitly
        av += ["--root", str(dirname)]
    with tempfile.TemporaryDirectory(prefix="amuse_ranks_") as md:
        had_stamp = "AMUSE_RUN_STAMP" in os.environ
        os.environ["AMUSE_RUN_STAMP"] = os.environ.get("AMUSE_RUN_STAMP") or datetime.now().strftime("%Y%m%d-%H%M%S")
        os.environ["AMUSE_MANIFEST_DIR"] = md
        try:
            rc = launch.run_ranks("amuse_amd.main", av, args.gpus, module=True)
        finally:
            os.environ.pop("AMUSE_MANIFEST_DIR", None)
            if not had_stamp:
                os.environ.pop("AMUSE_RUN_STAMP", None)      # (the stamp is this run's: a later run in the same process draws its own)
        if rc != 0:
            raise SystemExit(rc)
        written = []
        for r in range(args.gpus):
            f = Path(md, f"rank{r}.json")
            written += [Path(p) for p in json.loads(f.read_text())] if f.exists() else []
    print(f"AMUSE: ({args.fn[0]}) {args.gpus} ranks wrote {len(written)} NPZ files")
    return written


def main(argv=None):
    ap = argparse.ArgumentParser(description="AMUSE (MI355X path)")
    ap.add_argument("--fn", nargs="*", required=True, help="infer_gesture, edit_gesture, train_gesture")
    ap.add_argument("--cfg", default=None, help="config file (default <root>/configs/base_new.json)")
    ap.add_argument("--wandb", default=None, help="accepted for command-line compatibility; wandb is out of scope")
    ap.add_argument("--root", default=None, help="the reference-shaped tree (default: cwd.parent, scripts/main.py:229)")
    ap.add_argument("--audios", default=None, help="override custom_audios / emotion_control_list.audios")
    ap.add_argument("--renders", default=None, help="override custom_renders / emotion_control_list.renders")
    ap.add_argument("--eval-data", default=None, help="torch-saved dict for process_loader (dataset-driven edit tasks)")
    ap.add_argument("--epoch", default="6000", help="checkpoint epoch; the reference hard-codes 6000 (main.py:155-157,193); 'best' = lowest total loss")
    ap.add_argument("--random-init", action="store_true", help="deterministic random-init weights instead of checkpoints")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--precision", default="fp32x", choices=["fp32", "bf16", "fp32x", "fp16"],
                    help="fp32x (default): results match the reference's fp32 modules (eps_hat <= 1e-5) on the 16-bit MFMA; fp32: the same bars on "
                         "the fp32 MFMA, 2.6 x slower; fp16: the throughput mode (2 x fp32x, ~3e-3 on eps_hat; operands must stay below 65504: an "
                         "overflow surfaces as a non-finite output); bf16: the same speed with 8 x the rounding - only for activations beyond fp16 range")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--gpus", type=int, default=1, help="one process per GPU on this node: train_gesture = data-parallel ranks (RCCL all-reduce); infer_gesture / "
                                                         "edit_gesture = the job list cut into contiguous ranges, each rank embeds, samples and writes its own (no collective; "
                                                         "the NPZ files are byte for byte the single-process run's)")
    ap.add_argument("--all-pairs", action="store_true", help="edit_gesture: every *_source.wav edited with the emotion of every *_target.wav (S x T jobs) instead of "
                                                             "the reference's first source / first target pair (sets TRAIN_PARAM.test.emotion_control_list.all_pairs)")
    ap.add_argument("--epochs", type=int, default=None, help="train_gesture: override TRAIN_PARAM.latent_diffusion.n_epochs")
    ap.add_argument("--synthetic", action="store_true", help="train_gesture: synthetic batches instead of the LMDB cache")
    ap.add_argument("--iters-per-epoch", type=int, default=None, help="train_gesture --synthetic: iterations per epoch")
    ap.add_argument("--skip-vtex-loss", action="store_true", help="train_gesture: train without the two vertex-displacement loss terms when the "
                                                                  "configuration asks for them (they need the SMPL-X body models)")
    args = ap.parse_args(argv)
    fn = args.fn[0]
    if fn not in ("infer_gesture", "edit_gesture", "train_gesture"):
        raise SystemExit(f"--fn {fn}: infer_gesture, edit_gesture and train_gesture run on this path")
    tic = time.time()
    dirname = Path(args.root) if args.root else Path.cwd().parent
    config, ldm_cfg = load_config(dirname, fn, args.cfg)
    tp = config["TRAIN_PARAM"]
    if fn == "train_gesture":
        config["_ldm_cfg"] = ldm_cfg
        return train_gesture_entry(args, dirname, config)
    from . import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        return _launch_ranks(args, argv, dirname)
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world > 1:
        if args.device not in ("cuda", "cuda:0") and os.environ.get("AMUSE_SHARE_GPU") != "1":
            raise SystemExit(f"--device {args.device} with {world} ranks would put every rank on one GPU: drop the index (rank r uses cuda:<LOCAL_RANK>)")
        if os.environ.get("AMUSE_SHARE_GPU") != "1":     # (tests on a one-GPU box: AMUSE_SHARE_GPU=1 keeps every rank on --device)
            args.device = f"cuda:{local_rank}"
        torch.cuda.set_device(torch.device(args.device))
    if args.all_pairs:
        tp["test"].setdefault("emotion_control_list", {})["all_pairs"] = True
    pg = False
    if world > 1 and fn == "edit_gesture" and tp["test"].get("emotion_control_list", {}).get("all_pairs"):
        # the ONE exchange of the inference path: S sources x T targets - every rank's jobs read most WAVs, so each rank embeds a share and
        # the embeddings (3 x 256 floats per WAV) are all-gathered.  RCCL between the GPUs; gloo when the ranks share a device (tests).
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("AMUSE_SHARE_GPU") == "1":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(args.device))
        pg = True
    assert tp["pretrained_infer"], f"Arg: {fn} and pretrained_infer: {tp['pretrained_infer']} mismatch!"   # main.py:129
    assert tp["motion_extractor"]["use"] is False, "Motion extractor should be False!"
    if args.audios or args.renders:
        r, ecl = tp["baselines"]["renders"], tp["test"].get("emotion_control_list", {})
        if args.audios:
            r["custom_audios"], ecl["audios"] = args.audios, args.audios
        if args.renders:
            r["custom_renders"], ecl["renders"] = args.renders, args.renders
    device = torch.device(args.device)
    processed = dirname / "data" / "BEAT-processed"
    model_path = dirname / "saved-models"
    print(f"Experiment init: AMUSE, device: {device}, time: {time.asctime()}")
    fixseed(tp["seed"])
    print(f"Inferring for epoch {args.epoch}")
    tp["latent_diffusion"]["pretrained_prior_lpdm_e"] = args.epoch if args.epoch == "best" else int(args.epoch)
    tp["latent_diffusion"]["pretrained_ldm_lpdm_e"] = tp["latent_diffusion"]["pretrained_prior_lpdm_e"]
    baseline, modelversion, diffonly = False, tp["wav_dtw_mfcc"]["ablation"], tp["test"]["diff_only"]
    audio_list = tp["test"]["audio_list"]["use"]
    short_audio_list = tp["test"]["audio_list"]["short_audio_list"]
    if args.random_init:
        print("[amuse_amd] --random-init: deterministic random-init denoiser / prior / AST weights (seed 0)")
        model = PretrainedLPDM_v1.from_state_dicts(wts.make_denoiser_weights(0), wts.make_prior_weights(0), ldm_cfg,
                                                   device, seed=tp["seed"])
        for k in ("style_transfer", "emotion_control", "content_control", "style_Xemo_transfer"):
            setattr(model, k, tp["test"][k]["use"])
        wd = tp["wav_dtw_mfcc"]
        model.set_audio_encoders(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS), wd.get("dataset_mean", -9.173025),
                                 wd.get("dataset_std", 5.062332), wd.get("frame_based_feats", True))
        ldm_epoch = 0
    else:
        config["_ldm_cfg_override"] = ldm_cfg
        model = PretrainedLPDM_v1(None)
        ldm_epoch = model.setup(config, device, processed, None, False, baseline, verbose=False, diffonly=diffonly)
    model.precision = args.precision
    model.set_sampler(args.sampler, args.steps)
    eval_loader = torch.load(args.eval_data, weights_only=False) if args.eval_data else None
    tr = trainer(config, device, train_loader=eval_loader, model_path=model_path, tag="LPDM_infer", logger_cfg=None,
                 model=model, processed=processed, metricsmodel=None, b_path=None, EXEC_ON_CLUSTER=False,
                 debug=tp.get("debug", True), pretrained_infer=True, rank=rank, world=world)
    written = tr.eval_prior_latdiff_forward_backward_v1(baseline, ldm_epoch, audio_list, short_audio_list,
                                                        modelversion=modelversion, ammetric=True)
    torch.cuda.synchronize()
    print(f"AMUSE: ({fn}) completed in: {(time.time() - tic) / 3600} hrs; {len(written)} NPZ files" + (f" on rank {rank} of {world}" if world > 1 else ""))
    if world > 1 and os.environ.get("AMUSE_MANIFEST_DIR"):      # the launcher collects what the ranks wrote
        Path(os.environ["AMUSE_MANIFEST_DIR"], f"rank{rank}.json").write_text(json.dumps([str(p) for p in written]))
    if pg:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return written


if __name__ == "__main__":
    main()
