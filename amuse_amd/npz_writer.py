"""SMPL-X NPZ output of the reference (models/diffusion/viz/visualizer.py:344-364) and the output
packing of its call sites (scripts/trainer.py:524-526).  Rendering (Blender / ffmpeg) is out of scope."""
from __future__ import annotations

import random
import string
from pathlib import Path
from typing import Optional

import numpy as np
import torch

LOWER_BODY_JOINTS = [1, 2, 4, 5, 7, 8, 10, 11]  # "Lock below hips" (visualizer.py:346)
# dm/utils/ldm_evals.py:67-71 subject2genderbeta: the BEAT actors by gender, and the per-actor SMPL-X shape vectors
# (betas, 300 float64) the reference keeps as numpy literals (ldm_evals.py:348-379 fetchbetas, :456-2314) and writes into
# every NPZ.  amuse_amd/data/smplx_betas.npz is that table, extracted by tools/extract_betas.py in the build container.
MALE = ("wayne", "scott", "solomon", "lawrence", "stewart", "nidal", "zhao", "lu", "zhang", "carlos", "jorge", "itoi", "daiki", "jaime", "li")
FEMALE = ("carla", "sophie", "catherine", "miranda", "kieks", "ayana", "luqi", "hailing", "kexin", "goto", "reamey", "yingqing", "tiffnay", "hanieh", "katya")
_BETAS_FILE = Path(__file__).resolve().parent / "data" / "smplx_betas.npz"
_betas_table = None


def subject2gender(subject: str) -> str:
    if subject in MALE:
        return "male"
    if subject in FEMALE:
        return "female"
    raise KeyError(f"unknown BEAT actor {subject!r} (dm/utils/ldm_evals.py:67-71)")


def fetchbetas(actor: str) -> np.ndarray:
    """ldm_evals.py:348-379: the actor's (300,) float64 shape vector; actors without a MoSh fit raise like the reference."""
    global _betas_table
    if _betas_table is None:
        with np.load(_BETAS_FILE) as z:
            _betas_table = {k: z[k] for k in z.files}
    if actor not in _betas_table:
        raise NotImplementedError(f"Actor not found {actor}")
    return _betas_table[actor].copy()


def subject2genderbeta(subject: str):
    """ldm_evals.py:67-71 -> (gender <U7 array, betas)."""
    return np.array(subject2gender(subject), dtype="<U7"), fetchbetas(subject)


def pack_feats(poses: torch.Tensor, trans: torch.Tensor) -> torch.Tensor:
    """rearrange(poses, "b t j d -> b t (j d)") ++ trans -> (B, 300, 168)   (trainer.py:524-526)."""
    return torch.cat((poses.reshape(poses.shape[0], poses.shape[1], -1), trans), dim=-1)


def smplx_npz_fields(feat: np.ndarray, gender: str = "neutral", betas: Optional[np.ndarray] = None, fps: float = 30.0):
    """feat: (300, 168) = 55 joints x 3 axis-angle ++ 3 trans.  Returns the dict np.savez receives."""
    f = np.array(feat, dtype=np.float32).reshape(feat.shape[0], -1, 3)
    if f.shape[1] == 56:
        f = f[:, :-1, :]              # drop the translation row
    assert f.shape[1] == 55, f"expected 55 joints, got {f.shape[1]}"
    f[:, LOWER_BODY_JOINTS, :] = f[0, LOWER_BODY_JOINTS, :]   # freeze the lower body to frame 0
    return {"poses": f, "trans": np.zeros((f.shape[0], 3)), "gender": np.array(gender, dtype="<U7"),
            "betas": np.zeros(300) if betas is None else np.asarray(betas, dtype=np.float64),   # write_sample passes the actor's
            "mocap_frame_rate": np.array(fps, dtype="float64")}


def write_sample(feats: torch.Tensor, out_dir: Path, subject: str = "scott", rng: Optional[random.Random] = None,
                 betas: Optional[np.ndarray] = None):
    """feats: (n, 300, 168).  Writes <out_dir>/seq_<i>/<subject>_seq_<i>_<rand6>_motion_smplx.npz like
    CaMNVisualizer.animate_ldm_sample_v1 (visualizer.py:307-364); returns the paths.  gender and betas are the
    actor's (subject2genderbeta, visualizer.py:357) unless `betas` overrides the shape."""
    rng = rng or random
    paths = []
    if betas is None:
        betas = fetchbetas(subject)
    for i, feat in enumerate(feats):
        d = Path(out_dir) / f"seq_{i}"
        d.mkdir(parents=True, exist_ok=True)
        tag = "".join(rng.choice(string.ascii_uppercase + string.ascii_lowercase + string.digits) for _ in range(6))
        p = d / f"{subject}_seq_{i}_{tag}_motion_smplx.npz"
        np.savez(p, **smplx_npz_fields(feat.detach().cpu().numpy(), subject2gender(subject), betas))
        paths.append(p)
    return paths
