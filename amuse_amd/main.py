"""Entry point mirroring the reference's `python main.py --fn {infer_gesture,edit_gesture}`
(scripts/main.py:226-268 -> trainer.eval_prior_latdiff_forward_backward_v1, scripts/trainer.py:500-554,
1037-1098) for the part this library rebuilds: condition embeddings in, SMPL-X NPZ files out.

Inputs, either
  --audios DIR   10 s WAV files as the reference reads them (infer_gesture: every *.wav, trainer.py:514-521;
                 edit_gesture: one *_source.wav + one *_target.wav, trainer.py:1041-1053), run through the HIP audio
                 front-end (fbank + 3 x AST; --ast-dir = the reference's pretrained_ast directory, else random-init), or
  --cond FILE    the three 256-d speech embeddings per clip precomputed (.npz with `con`, `emo`, `sty` of shape
                 (N,256); for edit_gesture additionally `tgt_emo`).
The Blender / ffmpeg rendering is out of scope: the outputs stop at the `*_motion_smplx.npz` files.

  python -m amuse_amd.main --fn infer_gesture --audios viz_dump/test/speech --out renders/ [--model-dir saved-models/LPDM_x]
"""
from __future__ import annotations

import argparse
import json
import random
import time
from pathlib import Path

import numpy as np
import torch

from . import audio_weights as aw
from . import checkpoint as ckpt
from . import weights as wts
from .infer_ldm import PretrainedLPDM_v1
from .npz_writer import pack_feats, write_sample


def fixseed(seed: int):
    """scripts/utils/misc.py:93-101."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def build_model(args) -> "tuple[PretrainedLPDM_v1, int]":
    ldm_cfg = json.load(open(args.ldm_cfg)) if args.ldm_cfg else None
    if args.model_dir:
        lat = ckpt.pick_checkpoint(Path(args.model_dir), "latdiff", args.epoch)
        epoch = ckpt.epoch_of(lat)
        pri = ckpt.pick_checkpoint(Path(args.model_dir), "prior", epoch if args.epoch == "best" else args.epoch)
        dsd, psd = ckpt.load_denoiser_checkpoint(lat), ckpt.load_prior_checkpoint(pri)
    else:
        print("[amuse_amd] no --model-dir: using the deterministic random-init weights (seed 0)")
        dsd, psd, epoch = wts.make_denoiser_weights(0), wts.make_prior_weights(0), 0
    m = PretrainedLPDM_v1.from_state_dicts(dsd, psd, ldm_cfg, args.device, seed=args.seed)
    m.precision = args.precision
    m.set_sampler(args.sampler, args.steps)
    return m, epoch


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


def embed_wav(model: PretrainedLPDM_v1, path):
    """trainer.py:519-521: load, remove the global mean, process_single_seq."""
    a = load_wav(path)
    a = a - a.mean()
    return model.process_single_seq(a, framerate=16000, baseline=False)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--fn", required=True, choices=["infer_gesture", "edit_gesture"])
    ap.add_argument("--cond", default=None, help=".npz with con/emo/sty (N,256) [+ tgt_emo for edit_gesture]")
    ap.add_argument("--audios", default=None, help="directory of 10 s WAV files (edit_gesture: *_source.wav, *_target.wav)")
    ap.add_argument("--ast-dir", default=None, help="the reference's pretrained_ast directory (AST_EVP state dict)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--model-dir", default=None, help="dir with latdiff_*.pt / prior_*.pt (reference format)")
    ap.add_argument("--epoch", default="best")
    ap.add_argument("--ldm-cfg", default=None, help="configs/diff_latent_v2.json of the reference")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ddpm"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--actor", default="scott")   # hard-coded in the reference (trainer.py:518)
    ap.add_argument("--betas-from", default=None, help="SMPL-X npz of the actor whose `betas` go into the outputs")
    args = ap.parse_args(argv)
    t0 = time.time()
    fixseed(args.seed)
    if (args.cond is None) == (args.audios is None):
        ap.error("give exactly one of --cond and --audios")
    model, epoch = build_model(args)
    tgt = None
    if args.audios is not None:
        if args.ast_dir:
            sds = ckpt.load_ast_checkpoint(ckpt.pick_ast_checkpoint(Path(args.ast_dir), "full"))
        else:
            print("[amuse_amd] no --ast-dir: using the deterministic random-init AST weights (seed 0)")
            sds = {n: aw.make_ast_weights(0, n) for n in aw.ENCODERS}
        model.set_audio_encoders(sds["con"], sds["emo"], sds["sty"])
        wavs = sorted(Path(args.audios).glob("*.wav"))
        if args.fn == "edit_gesture":      # trainer.py:1041-1053
            src = [x for x in wavs if "_source" in x.stem][0]
            tg = [x for x in wavs if "_target" in x.stem][0]
            con, emo, sty = embed_wav(model, src)
            tgt = embed_wav(model, tg)[1]
        else:
            if not wavs:
                raise FileNotFoundError(f"no *.wav under {args.audios}")
            embs = [embed_wav(model, w) for w in wavs]
            con, emo, sty = (torch.cat([e[k] for e in embs]) for k in range(3))
    else:
        z = np.load(args.cond)
        con, emo, sty = (torch.from_numpy(z[k]).float() for k in ("con", "emo", "sty"))
        if args.fn == "edit_gesture":
            tgt = torch.from_numpy(z["tgt_emo"]).float()
    stamp = time.strftime("%Y%m%d-%H%M%S")
    root = Path(args.out) / f"Custom_audios_{stamp}_E{epoch}" / "rep0"
    written = []
    betas = np.load(args.betas_from, allow_pickle=True)["betas"] if args.betas_from else None
    if args.fn == "infer_gesture":     # trainer.py:516-539: one diffusion_backward(1, ...) per audio
        for i in range(con.shape[0]):
            r = model.diffusion_backward(1, con[i:i + 1], emo[i:i + 1], sty[i:i + 1])
            written += write_sample(pack_feats(r["poses"], r["trans"]), root / f"rst_{i}", args.actor, betas=betas)
    else:                              # trainer.py:1037-1075: original, then the same with the target's emotion
        for i in range(con.shape[0]):
            c0 = model._clip_counter
            a = model.diffusion_backward(1, con[i:i + 1], emo[i:i + 1], sty[i:i + 1], clip_index0=c0)
            b = model.diffusion_backward(1, con[i:i + 1], tgt[i:i + 1], sty[i:i + 1], clip_index0=c0)
            model._clip_counter += 1
            written += write_sample(pack_feats(a["poses"], a["trans"]), root / f"pair_{i}" / "rst_0", args.actor, betas=betas)
            written += write_sample(pack_feats(b["poses"], b["trans"]), root / f"pair_{i}" / "rst_1", args.actor, betas=betas)
    torch.cuda.synchronize()
    print(f"[LDM EVAL] {args.fn} done: {len(written)} NPZ files under {root}, total time elapsed: {time.time() - t0:.4f} s")
    return written


if __name__ == "__main__":
    main()
