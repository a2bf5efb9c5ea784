"""CPU, two gloo ranks: the multi-rank form of the product's own infer_gesture / edit_gesture entry points (amuse_amd/trainer.py with
rank / world; what `main.py --fn ... --gpus N` runs per GPU).  A stub model stands in for the HIP engine (no GPU here): embeddings are a
deterministic function of the waveform, poses of (embedding, GLOBAL clip index) - so a job sampled on the wrong rank, with the wrong clip
index or from the wrong WAV changes the bytes of its NPZ.  Checked against the single-process run of the same tree:
  * the job -> rank map is shard.job_range's (contiguous, balanced),
  * the union of the ranks' files is the single-process file SET - same directories, same random tags - and every file has the same bytes,
  * a rank embeds only the WAVs its jobs read; with the many-to-many edit batch (all_pairs) each rank embeds a share and the embeddings are
    all-gathered over gloo (the path's one exchange).
The GPU counterpart (real engine, two processes on one MI355X): tests/test_gpu_shard_cli.py.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]

_WORKER = r'''
import json, os, random, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, str(Path(sys.argv[1]) / "tests"))
root, fn, out_json = Path(sys.argv[2]), sys.argv[3], sys.argv[4]
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
from amuse_amd.main import load_config, fixseed
from amuse_amd.trainer import trainer

class Stub:
    """poses = f(embedding row, global clip index); records what it was asked for."""
    def __init__(self):
        self.device, self._clip_counter, self.embedded, self.calls = torch.device("cpu"), 0, [], []
    def process_seq_list(self, waves, framerate=16000, baseline=False):
        out = []
        for w in waves:
            self.embedded.append(float(w.abs().sum()))
            g = torch.Generator().manual_seed(int(w.abs().sum() * 1000) % (2 ** 31))
            out.append(tuple(torch.randn(1, 256, generator=g) for _ in range(3)))
        return out
    def diffusion_backward(self, bsz, z_con, z_emo, z_sty, clip_index0=None, return_latents=False):
        c0 = self._clip_counter if clip_index0 is None else clip_index0
        if clip_index0 is None:
            self._clip_counter += bsz
        self.calls.append((bsz, c0))
        idx = torch.arange(c0, c0 + bsz, dtype=torch.float32)
        feat = (z_con[:, :1] + 2 * z_emo[:, :1] + 3 * z_sty[:, :1])[:, :, None, None]
        poses = (feat + 0.001 * idx[:, None, None, None]).expand(bsz, 300, 55, 3).contiguous()
        return {"poses": poses, "trans": torch.zeros(bsz, 300, 3)}

config, _ = load_config(root, fn)
tp = config["TRAIN_PARAM"]
if len(sys.argv) > 5 and sys.argv[5] == "all_pairs":
    tp["test"]["emotion_control_list"]["all_pairs"] = True
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
fixseed(tp["seed"])
m = Stub()
tr = trainer(config, "cpu", model=m, model_path=root / "saved-models", processed=root / "data/BEAT-processed", rank=rank, world=world,
             stamp="20260101-000000")
written = tr.eval_prior_latdiff_forward_backward_v1(False, 7, tp["test"]["audio_list"]["use"], False, modelversion=tp["wav_dtw_mfcc"]["ablation"])
json.dump({"written": [str(p) for p in written], "calls": m.calls, "n_embedded": len(m.embedded)}, open(out_json, "w"))
if world > 1 and torch.distributed.is_initialized():
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "AMUSE_RUN_STAMP")}
    env.update(OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    return env


def _run(tmp_path, root, fn, world, port, *extra):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    outs = [tmp_path / f"{fn}_{world}_{r}.json" for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO), str(root), fn, str(outs[r]), *extra],
                              env=dict(_env(), RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT=str(port)) if world > 1 else _env(),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    return [json.load(open(o)) for o in outs]


def _tree(tmp_path, name, n_wavs=5, pairs=1):
    from conftest import make_reference_tree
    import numpy as np
    from scipy.io import wavfile
    root = make_reference_tree(tmp_path / name, n_infer_wavs=n_wavs)
    rng = np.random.default_rng(5)
    for k in range(1, pairs):          # more *_source / *_target pairs for the many-to-many edit batch
        for kind in ("source", "target"):
            wavfile.write(root / f"viz_dump/test/e_speech/{9 + k}_miranda_{kind}.wav", 16000, (rng.standard_normal(40000) * 3000).astype(np.int16))
    return root


def _files(root, sub):
    base = root / "viz_dump/test" / sub
    return {str(p.relative_to(base)): p.read_bytes() for p in base.rglob("*.npz")}


def test_infer_gesture_two_ranks_write_the_single_process_files(tmp_path):
    one, two = _tree(tmp_path, "one"), _tree(tmp_path, "two")
    r1 = _run(tmp_path, one, "infer_gesture", 1, 0)[0]
    r2 = _run(tmp_path, two, "infer_gesture", 2, 29671)
    assert r1["calls"] == [[5, 0]] and r1["n_embedded"] == 5
    # job -> rank: 5 one-clip jobs over 2 ranks = [0, 3) and [3, 5), clip indices global; a rank embeds its own audios only
    assert r2[0]["calls"] == [[3, 0]] and r2[1]["calls"] == [[2, 3]]
    assert r2[0]["n_embedded"] == 3 and r2[1]["n_embedded"] == 2
    assert len(r2[0]["written"]) == 3 and len(r2[1]["written"]) == 2 and not set(r2[0]["written"]) & set(r2[1]["written"])
    f1, f2 = _files(one, "gesture"), _files(two, "gesture")
    assert len(f1) == 5 and set(f1) == set(f2)                     # same directories, same random tags
    assert all(f1[k] == f2[k] for k in f1)                         # same bytes
    # written order on a rank = job order: concatenated over ranks it is the single-process order
    rel = lambda ps, root: [str(Path(p).relative_to(root)) for p in ps]
    assert rel(r2[0]["written"] + r2[1]["written"], two) == rel(r1["written"], one)


def test_edit_gesture_demo_pair_two_ranks(tmp_path):
    """The reference's demo (first source, first target -> "Original" + "Emotion edited"): 2 jobs, one per rank; rank 0 reads the source only, rank 1 both."""
    one, two = _tree(tmp_path, "one"), _tree(tmp_path, "two")
    r1 = _run(tmp_path, one, "edit_gesture", 1, 0)[0]
    r2 = _run(tmp_path, two, "edit_gesture", 2, 29672)
    assert r1["calls"] == [[2, 0]] and r2[0]["calls"] == [[1, 0]] and r2[1]["calls"] == [[1, 1]]
    assert r2[0]["n_embedded"] == 1 and r2[1]["n_embedded"] == 2
    f1, f2 = _files(one, "e_gesture"), _files(two, "e_gesture")
    assert len(f1) == 2 and set(f1) == set(f2) and all(f1[k] == f2[k] for k in f1)


def test_edit_gesture_all_pairs_exchanges_embeddings_over_gloo(tmp_path):
    """3 sources x 3 targets = 9 jobs over 2 ranks ([0, 5) and [5, 9)): each rank embeds its share of the 6 WAVs (3 each) and the rows are
    all-gathered; files equal the single-process run's, which embeds all 6 itself."""
    one, two = _tree(tmp_path, "one", pairs=3), _tree(tmp_path, "two", pairs=3)
    r1 = _run(tmp_path, one, "edit_gesture", 1, 0, "all_pairs")[0]
    r2 = _run(tmp_path, two, "edit_gesture", 2, 29673, "all_pairs")
    assert r1["calls"] == [[9, 0]] and r1["n_embedded"] == 6
    assert r2[0]["calls"] == [[5, 0]] and r2[1]["calls"] == [[4, 5]]
    assert r2[0]["n_embedded"] == 3 and r2[1]["n_embedded"] == 3
    f1, f2 = _files(one, "e_gesture"), _files(two, "e_gesture")
    assert len(f1) == 9 and set(f1) == set(f2) and all(f1[k] == f2[k] for k in f1)


def test_job_range_cuts_on_tile_boundaries():
    from amuse_amd import shard
    from amuse_amd.trainer import local_jobs
    # 300 one-clip jobs with five tokens: ceil(300 / 128) = 3 clips per tile -> cuts at multiples of 3
    g = shard.job_plan(300, 5)["clips_per_group"]
    assert g == 3
    cuts = [shard.job_range([1] * 300, r, 4, align=g) for r in range(4)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 300 and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    assert all(a % g == 0 for a, _ in cuts)
    # jobs of unequal size: a cut only where the clip offset is a multiple of the tile
    bszs = [2, 2, 1, 3, 2, 2, 4, 1, 1, 2] * 20      # 400 clips -> 4 clips per tile (3 tokens allow 5; ceil(400 / 128) = 4)
    g = shard.job_plan(sum(bszs), 3)["clips_per_group"]
    offs = [0]
    for b in bszs:
        offs.append(offs[-1] + b)
    for w in (2, 3, 8):
        rs = [shard.job_range(bszs, r, w, align=g) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == len(bszs) and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert all(offs[a] % g == 0 for a, _ in rs)
    # runs with different token sets are cut one by one
    specs = [(1, False, False)] * 4 + [(1, True, False)] * 2 + [(1, False, False)] * 3
    m0, m1 = local_jobs(specs, 0, 2), local_jobs(specs, 1, 2)
    assert [a ^ b for a, b in zip(m0, m1)] == [True] * 9
    assert m0 == [True, True, False, False, True, False, True, True, False]


def test_main_gpus_2_starts_its_own_ranks_for_infer_gesture(tmp_path):
    """`python -m amuse_amd.main --fn infer_gesture --gpus 2` typed directly: the parent starts two ranks of the module (amuse_amd/launch.py) with
    one time stamp; without a GPU the CHILDREN stop at the HIP engine - the parent itself never needed one."""
    root = _tree(tmp_path, "t", n_wavs=2)
    r = subprocess.run([sys.executable, "-m", "amuse_amd.main", "--fn", "infer_gesture", "--gpus", "2", "--root", str(root), "--random-init",
                        "--renders", str(tmp_path / "out")], cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "torch.distributed" in out or "ChildFailedError" in out or "rank" in out.lower(), out[-2000:]
    assert "no HIP" in out or "hip" in out.lower(), out[-2000:]


def test_job_range_properties_hypothesis():
    """shard.job_range / trainer.local_jobs over random job lists: the ranks' ranges are disjoint, ordered and cover every job exactly once; every cut lies on a job
    boundary whose clip offset is a multiple of the launch's tile size (or at the list's end); one rank takes everything."""
    from hypothesis import given, settings, strategies as st
    from amuse_amd import shard
    from amuse_amd.trainer import job_runs, local_jobs

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.tuples(st.integers(1, 9), st.booleans(), st.booleans()), min_size=0, max_size=60), st.integers(1, 9))
    def check(specs, world):
        masks = [local_jobs(specs, r, world) for r in range(world)]
        n = len(specs)
        assert all(sum(m[j] for m in masks) == 1 for j in range(n))                      # every job on exactly one rank
        if world == 1:
            assert masks[0] == [True] * n
        dicts = [{"bsz": b, "no_emo": e, "no_sty": s} for b, e, s in specs]
        for (no_emo, no_sty), idx in job_runs(dicts):
            bszs = [specs[j][0] for j in idx]
            g = shard.job_plan(sum(bszs), 5 - no_emo - no_sty)["clips_per_group"]
            offs = [0]
            for b in bszs:
                offs.append(offs[-1] + b)
            prev_end = 0
            for r in range(world):
                ja, jb = shard.job_range(bszs, r, world, align=g)
                assert ja == prev_end and ja <= jb                                           # contiguous, ordered
                assert offs[ja] % g == 0 or ja == len(bszs)                                  # cut on a tile boundary of the launch
                assert [masks[r][idx[k]] for k in range(len(idx))] == [ja <= k < jb for k in range(len(idx))]
                prev_end = jb
            assert prev_end == len(bszs)
    check()
