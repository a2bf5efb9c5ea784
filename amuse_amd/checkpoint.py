"""Readers (and a writer, for tests) for the two checkpoint formats the reference loads.

  latdiff_*_e<N>.pt : {"epoch", "model_state_dict" (keys "denoiser.<name>"), "optimizer_state_dict"}
                      - picked and copied at infer_ldm.py:75-104 (count must equal the 130-entry state dict)
  prior_*_e<N>.pt   : {"epoch", "model_state_dict"} with un-prefixed MotionPrior keys
                      - picked at infer_pretrained_vae.py:30-47
File choice: by trailing `_e<epoch>` when an epoch is given, else ("best") by the smallest total loss
encoded in the second-to-last `_` field of the file name (e.g. `..._total0.0123_e6000.pt`).
Saved by scripts/trainer.py:468-496.

  <pretrained_ast>/*.pt : the AST_EVP state dict itself (torch.load(best) -> load_state_dict,
                      models/audio/infer_pretrained_ast_evp.py:36-39), keys `{con,emo,sty}_enc.<ASTModel name>` plus the
                      fusion / decoder / classifier heads the inference path never touches.  File choice
                      (infer_pretrained_ast_evp.py:21-33): the number in the 4th `_` field of the stem (emotion
                      accuracy) is maximised - the 5th (person accuracy) for the "identity" ablation - and a winner
                      from epoch 0 is replaced by the first file with `_1_` in its path.
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Dict, Tuple, Union

import numpy as np
import torch

from . import audio_weights as aw
from . import weights as wts


def pick_checkpoint(model_dir: Path, prefix: str, epoch: Union[str, int] = "best") -> Path:
    files = [f for f in Path(model_dir).iterdir()
             if f.is_file() and "experiment_args.json" not in str(f) and f.stem.split("_")[0] == prefix]
    if not files:
        raise FileNotFoundError(f"no {prefix}_* checkpoint in {model_dir}")
    if epoch == "best":
        best, total = None, np.inf
        for f in files:
            t = float(re.findall(r"\d+\.\d+", f.stem.split("_")[-2])[0])
            if t < total:
                total, best = t, f
        return best
    hits = [f for f in files if int(re.search(r"\d+", f.stem.split("_")[-1]).group()) == int(epoch)]
    if not hits:
        raise FileNotFoundError(f"no {prefix}_* checkpoint for epoch {epoch} in {model_dir}")
    return hits[0]


def epoch_of(path: Path) -> int:
    return int(re.search(r"\d+", Path(path).stem.split("_")[-1]).group())


def load_denoiser_checkpoint(path: Path, arch: str = "trans_enc", diffusion_only: bool = False) -> Dict[str, np.ndarray]:
    """infer_ldm.py:86-104: every `denoiser.*` tensor of the checkpoint, the count asserted against the state dict of
    Denoiser(arch, diffusion_only) (130 entries for the shipped configuration; 176 / 134 / 180 for the variants)."""
    chk = torch.load(path, map_location="cpu", weights_only=False)
    spec = wts.denoiser_param_spec(arch, diffusion_only)
    out, count = {}, 0
    for name, p in chk["model_state_dict"].items():
        if name.startswith("denoiser"):
            count += 1
            key = name[len("denoiser") + 1:]
            assert key in spec, f"key {key} not found in denoiser"
            out[key] = p.detach().cpu().numpy().astype(np.float32)
    assert count == len(spec), f"state_dict_count {count} != len(tgt_state) {len(spec)} for denoiser"
    return out


def load_prior_checkpoint(path: Path) -> Dict[str, np.ndarray]:
    chk = torch.load(path, map_location="cpu", weights_only=False)
    spec = wts.prior_param_spec()
    sd = chk["model_state_dict"]
    missing = [k for k in spec if k not in sd]
    if missing:
        raise KeyError(f"prior checkpoint misses {len(missing)} keys, e.g. {missing[:3]}")
    return {k: sd[k].detach().cpu().numpy().astype(np.float32) for k in spec}


def save_reference_format(model_dir: Path, denoiser_sd, prior_sd, epoch: int = 6000, total: float = 0.0123) -> Tuple[Path, Path]:
    """Write a (latdiff, prior) pair in the reference's on-disk format (used by tests / demos)."""
    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    lat = model_dir / f"latdiff_model_wOpt_total{total:.4f}_e{epoch}.pt"
    pri = model_dir / f"prior_model_NoOpt_total{total:.4f}_e{epoch}.pt"
    torch.save({"epoch": epoch, "optimizer_state_dict": {},
                "model_state_dict": {f"denoiser.{k}": torch.from_numpy(np.asarray(v)) for k, v in denoiser_sd.items()}}, lat)
    torch.save({"epoch": epoch, "model_state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in prior_sd.items()}}, pri)
    return lat, pri


def _first_number(x: str):
    """Pretrained_AST_EVP._get_num (infer_pretrained_ast_evp.py:53-58)."""
    chars = [c if c.isdigit() or c == "." else " " for c in x]
    parts = "".join(chars).split()
    return float(parts[0]) if parts else None


def pick_ast_checkpoint(model_dir: Path, audio_ablation: str = "full") -> Path:
    assert audio_ablation in ("full", "identity", "emotion", "ast_baseline", None), f"[LATDIFF] Invalid audio ablation flag: {audio_ablation}"
    files = [f for f in Path(model_dir).iterdir() if f.is_file() and "experiment_args.json" not in str(f)]
    if not files:
        raise FileNotFoundError(f"no AST checkpoint in {model_dir}")
    field = 4 if audio_ablation == "identity" else 3
    best, acc = None, -np.inf
    for f in files:
        a = _first_number(f.stem.split("_")[field])
        if a is not None and a > acc:
            acc, best = a, f
    if best is None:
        raise FileNotFoundError(f"no AST checkpoint with an accuracy field in {model_dir}")
    if int(_first_number(best.stem.split("_")[1])) == 0:
        best = [f for f in files if "_1_" in str(f)][0]
    return best


def load_ast_checkpoint(path: Path) -> Dict[str, Dict[str, np.ndarray]]:
    """-> {"con": sd, "emo": sd, "sty": sd} with the forward-pass tensors of each ASTModel (audio_weights.ast_param_spec)."""
    sd = torch.load(path, map_location="cpu", weights_only=False)
    spec = aw.ast_param_spec()
    out = {}
    for enc in aw.ENCODERS:
        missing = [k for k in spec if f"{enc}_enc.{k}" not in sd]
        if missing:
            raise KeyError(f"AST checkpoint misses {len(missing)} keys of {enc}_enc, e.g. {missing[:3]}")
        out[enc] = {k: sd[f"{enc}_enc.{k}"].detach().cpu().numpy().astype(np.float32) for k in spec}
    return out


def save_ast_reference_format(model_dir: Path, sds: Dict[str, Dict[str, np.ndarray]], epoch: int = 12, emo_acc: float = 0.91,
                              person_acc: float = 0.88) -> Path:
    """Write an AST_EVP state dict in the reference's on-disk form (tests / demos)."""
    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    path = model_dir / f"model_e{epoch}_loss0.1234_tEAcc{emo_acc:.4f}_tPAcc{person_acc:.4f}.pt"
    flat = {f"{enc}_enc.{k}": torch.from_numpy(np.asarray(v)) for enc, sd in sds.items() for k, v in sd.items()}
    torch.save(flat, path)
    return path
