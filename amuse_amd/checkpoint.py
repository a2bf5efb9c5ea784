"""Readers (and a writer, for tests) for the two checkpoint formats the reference loads.

  latdiff_*_e<N>.pt : {"epoch", "model_state_dict" (keys "denoiser.<name>"), "optimizer_state_dict"}
                      - picked and copied at infer_ldm.py:75-104 (count must equal the 130-entry state dict)
  prior_*_e<N>.pt   : {"epoch", "model_state_dict"} with un-prefixed MotionPrior keys
                      - picked at infer_pretrained_vae.py:30-47
File choice: by trailing `_e<epoch>` when an epoch is given, else ("best") by the smallest total loss
encoded in the second-to-last `_` field of the file name (e.g. `..._total0.0123_e6000.pt`).
Saved by scripts/trainer.py:468-496.
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Dict, Tuple, Union

import numpy as np
import torch

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


def load_denoiser_checkpoint(path: Path) -> Dict[str, np.ndarray]:
    chk = torch.load(path, map_location="cpu", weights_only=False)
    spec = wts.denoiser_param_spec()
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
