"""GPU: `python -m amuse_amd.main --fn infer_gesture | edit_gesture --gpus 2` as typed - the parent starts two ranks of the module
(amuse_amd/launch.py), each with its own HIP engines - against the single-process run of the same tree: the same NPZ files (directories
and random tags) with the SAME BYTES.  On this box's one GPU the ranks share the device (AMUSE_SHARE_GPU=1; the all_pairs exchange then
goes over gloo instead of RCCL, which refuses two ranks on one device)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]


def _tree(tmp_path, name, n_wavs, pairs=1):
    from conftest import make_reference_tree
    from scipy.io import wavfile
    root = make_reference_tree(tmp_path / name, n_infer_wavs=n_wavs)
    rng = np.random.default_rng(5)
    for k in range(1, pairs):
        for kind in ("source", "target"):
            wavfile.write(root / f"viz_dump/test/e_speech/{9 + k}_miranda_{kind}.wav", 16000, (rng.standard_normal(90000) * 3000).astype(np.int16))
    return root


def _cli(root, fn, gpus, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "AMUSE_RUN_STAMP")}
    env.update(AMUSE_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", AMUSE_RUN_STAMP="20260101-000000")
    r = subprocess.run([sys.executable, "-m", "amuse_amd.main", "--fn", fn, "--root", str(root), "--random-init", "--gpus", str(gpus), *extra],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout


def _files(root, sub):
    base = root / "viz_dump/test" / sub
    return {str(p.relative_to(base)): p.read_bytes() for p in base.rglob("*.npz")}


@pytest.mark.parametrize("precision", ["fp32x", "bf16"])
def test_infer_gesture_two_ranks_bytes_equal_single_process(tmp_path, precision):
    one, two = _tree(tmp_path, "one", 5), _tree(tmp_path, "two", 5)
    _cli(one, "infer_gesture", 1, "--precision", precision)
    out = _cli(two, "infer_gesture", 2, "--precision", precision)
    assert "2 ranks wrote 5 NPZ files" in out and "on rank 0 of 2" in out and "on rank 1 of 2" in out
    f1, f2 = _files(one, "gesture"), _files(two, "gesture")
    assert len(f1) == 5 and set(f1) == set(f2)
    assert all(f1[k] == f2[k] for k in f1)
    # (not trivially equal: the five clips differ from each other)
    poses = [np.load(p)["poses"] for p in sorted((one / "viz_dump/test/gesture").rglob("*.npz"))]
    assert all(not np.array_equal(poses[0], q) for q in poses[1:])


def test_edit_gesture_all_pairs_two_ranks_bytes_equal_single_process(tmp_path):
    """3 sources x 3 targets = 9 jobs; each rank embeds 3 of the 6 WAVs, the embeddings are all-gathered, rank 0 samples jobs 0-4, rank 1 jobs 5-8."""
    one, two = _tree(tmp_path, "one", 1, pairs=3), _tree(tmp_path, "two", 1, pairs=3)
    _cli(one, "edit_gesture", 1, "--all-pairs")
    out = _cli(two, "edit_gesture", 2, "--all-pairs")
    assert "2 ranks wrote 9 NPZ files" in out
    f1, f2 = _files(one, "e_gesture"), _files(two, "e_gesture")
    assert len(f1) == 9 and set(f1) == set(f2)
    assert all(f1[k] == f2[k] for k in f1)


def test_edit_gesture_reference_pair_two_ranks(tmp_path):
    one, two = _tree(tmp_path, "one", 1), _tree(tmp_path, "two", 1)
    _cli(one, "edit_gesture", 1)
    _cli(two, "edit_gesture", 2)
    f1, f2 = _files(one, "e_gesture"), _files(two, "e_gesture")
    assert len(f1) == 2 and set(f1) == set(f2) and all(f1[k] == f2[k] for k in f1)
