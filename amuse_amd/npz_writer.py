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
# dm/utils/ldm_evals.py:67-71 subject2genderbeta: the BEAT actors by gender.  The per-actor SMPL-X shape vectors
# (betas, 300 values) come from MoSh fits that are not part of the reference repository: pass them in (`betas=`, CLI
# --betas-from <any SMPL-X npz of that actor>), otherwise zeros (the mean shape) are written.
MALE = ("wayne", "scott", "solomon", "lawrence", "stewart", "nidal", "zhao", "lu", "zhang", "carlos", "jorge", "itoi", "daiki", "jaime", "li")
FEMALE = ("carla", "sophie", "catherine", "miranda", "kieks", "ayana", "luqi", "hailing", "kexin", "goto", "reamey", "yingqing", "tiffnay", "hanieh", "katya")


def subject2gender(subject: str) -> str:
    if subject in MALE:
        return "male"
    if subject in FEMALE:
        return "female"
    raise KeyError(f"unknown BEAT actor {subject!r} (dm/utils/ldm_evals.py:67-71)")


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
            "betas": np.zeros(300) if betas is None else np.asarray(betas, dtype=np.float64),
            "mocap_frame_rate": np.array(fps, dtype="float64")}


def write_sample(feats: torch.Tensor, out_dir: Path, subject: str = "scott", rng: Optional[random.Random] = None,
                 betas: Optional[np.ndarray] = None):
    """feats: (n, 300, 168).  Writes <out_dir>/seq_<i>/<subject>_seq_<i>_<rand6>_motion_smplx.npz like
    CaMNVisualizer.animate_ldm_sample_v1 (visualizer.py:307-364); returns the paths."""
    rng = rng or random
    paths = []
    for i, feat in enumerate(feats):
        d = Path(out_dir) / f"seq_{i}"
        d.mkdir(parents=True, exist_ok=True)
        tag = "".join(rng.choice(string.ascii_uppercase + string.ascii_lowercase + string.digits) for _ in range(6))
        p = d / f"{subject}_seq_{i}_{tag}_motion_smplx.npz"
        np.savez(p, **smplx_npz_fields(feat.detach().cpu().numpy(), subject2gender(subject), betas))
        paths.append(p)
    return paths
