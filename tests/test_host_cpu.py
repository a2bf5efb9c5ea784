"""CPU: host-side mirror of the reference interface - checkpoint formats, NPZ writer, sharding (gloo, 2 ranks)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO


def test_checkpoint_roundtrip_and_best_pick(tmp_path):
    from amuse_amd import checkpoint as ckpt, weights as wts
    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    ckpt.save_reference_format(tmp_path, wd, wp, epoch=6000, total=0.0123)
    ckpt.save_reference_format(tmp_path, wts.make_denoiser_weights(1), wp, epoch=5800, total=0.0456)
    (tmp_path / "experiment_args.json").write_text("{}")
    best = ckpt.pick_checkpoint(tmp_path, "latdiff", "best")
    assert ckpt.epoch_of(best) == 6000                       # smallest total loss wins (infer_ldm.py:78-84)
    assert ckpt.epoch_of(ckpt.pick_checkpoint(tmp_path, "latdiff", 5800)) == 5800
    got = ckpt.load_denoiser_checkpoint(best)
    assert list(got.keys()) == list(wd.keys()) or set(got.keys()) == set(wd.keys())
    assert all(np.array_equal(got[k], wd[k]) for k in wd)
    gp = ckpt.load_prior_checkpoint(ckpt.pick_checkpoint(tmp_path, "prior", 6000))
    assert all(np.array_equal(gp[k], wp[k]) for k in wp)
    with pytest.raises(FileNotFoundError):
        ckpt.pick_checkpoint(tmp_path, "latdiff", 1)
    # a checkpoint missing a tensor trips the reference's count assertion (infer_ldm.py:103)
    bad = torch.load(best, weights_only=False)
    bad["model_state_dict"].pop("denoiser.encoder.norm.bias")
    torch.save(bad, tmp_path / "latdiff_model_wOpt_total0.0001_e7000.pt")
    with pytest.raises(AssertionError):
        ckpt.load_denoiser_checkpoint(tmp_path / "latdiff_model_wOpt_total0.0001_e7000.pt")


def test_npz_writer_matches_reference_layout(tmp_path):
    from amuse_amd.npz_writer import LOWER_BODY_JOINTS, pack_feats, write_sample
    lay = json.load(open(GOLDEN / "npz_layout.json"))
    poses = torch.randn(2, 300, 55, 3)
    trans = torch.randn(2, 300, 3)
    feats = pack_feats(poses, trans)
    assert feats.shape == (2, 300, 168)
    assert torch.equal(feats[..., :165].reshape(2, 300, 55, 3), poses) and torch.equal(feats[..., 165:], trans)
    paths = write_sample(feats, tmp_path / "rst_0", "scott")
    assert [p.parent.name for p in paths] == ["seq_0", "seq_1"] and paths[0].name.startswith("scott_seq_0_")
    z = np.load(paths[0], allow_pickle=True)
    ref = next(iter(lay.values()))["fields"]
    for k, (dt, shape) in ref.items():
        assert str(z[k].dtype) == dt and list(z[k].shape) == shape, k
    assert str(z["gender"]) == "male"                       # scott (dm/utils/ldm_evals.py:67-71)
    from amuse_amd.npz_writer import subject2gender
    assert subject2gender("miranda") == "female"
    with pytest.raises(KeyError):
        subject2gender("nobody")
    z2 = np.load(write_sample(feats[:1], tmp_path / "rst_1", "miranda", betas=np.arange(300.0))[0], allow_pickle=True)
    assert str(z2["gender"]) == "female" and np.array_equal(z2["betas"], np.arange(300.0))
    assert np.all(z["trans"] == 0) and float(z["mocap_frame_rate"]) == 30.0
    assert np.all(z["poses"][:, LOWER_BODY_JOINTS] == z["poses"][0, LOWER_BODY_JOINTS])
    other = [j for j in range(55) if j not in LOWER_BODY_JOINTS]
    assert np.array_equal(z["poses"][:, other], poses[0].numpy()[:, other])


def test_shard_range_partitions():
    from amuse_amd.shard import shard_range
    for total in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(256, 3, 8) == (96, 128)
    # shards aligned to the job's clips per workgroup tile (bitwise reproducibility across shardings)
    from amuse_amd.shard import job_clips_per_group
    assert [job_clips_per_group(n) for n in (1, 128, 129, 256, 257, 4096)] == [1, 1, 2, 2, 3, 3]
    assert job_clips_per_group(4096, tokens=3) == 5
    for total, world, g in ((300, 4, 3), (10, 4, 3), (256, 8, 2), (7, 3, 2)):
        r = [shard_range(total, k, world, align=g) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(lo % g == 0 for lo, _ in r)
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from amuse_amd import weights as wts
from amuse_amd.shard import sample_sharded
from oracle import amuse_oracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
Wd = orc.to_torch(wts.make_denoiser_weights(0))
sched = orc.DDPM(4)
def sample_fn(bsz, con, emo, sty, clip_index0=0):   # CPU stand-in for PretrainedLPDM_v1.diffusion_backward
    clips = np.arange(clip_index0, clip_index0 + bsz)
    x = torch.from_numpy(orc.counter_normal(7, clips, 0, 0))
    nz = torch.stack([torch.from_numpy(orc.counter_normal(7, clips, s, 1)) for s in range(4)])
    return {"latents": orc.sample_latents(Wd, sched, con, emo, sty, x, nz)}
g = torch.Generator().manual_seed(3)
con, emo, sty = (torch.randn(5, 256, generator=g) for _ in range(3))
full = sample_sharded(sample_fn, con, emo, sty, rank, world, gather=True)
if rank == 0:
    single = sample_fn(5, con, emo, sty, clip_index0=0)
    assert full["latents"].shape == (5, 128)
    # (CPU BLAS picks kernels by batch size, so CPU shards agree to rounding; the GPU test
    #  test_in_kernel_noise_is_shard_invariant asserts BITWISE equality for the HIP path)
    assert torch.allclose(full["latents"], single["latents"], atol=1e-5), "sharded result differs from single-process result"
    x0 = torch.from_numpy(orc.counter_normal(7, np.arange(3, 5), 0, 0))
    assert torch.equal(x0, torch.from_numpy(orc.counter_normal(7, np.arange(0, 5), 0, 0))[3:])  # noise keyed by global index
    print("SHARD_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding_is_invisible(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO)], env=dict(env, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARD_OK" in outs[0]


def test_process_loader_swaps_like_the_reference():
    from amuse_amd.infer_ldm import PretrainedLPDM_v1
    m = PretrainedLPDM_v1()
    m.style_transfer, m.emotion_control, m.style_Xemo_transfer = False, True, False
    mk = lambda v: {"ld_z_con": v, "ld_z_emo": v + 10, "ld_z_sty": v + 20}
    data = {"wayne": {f"take{i}": mk(i) for i in range(3)}}
    out = m.process_loader({"emotion_control": data, "emotion_control_info": "[wayne]_[neutral]_first"})
    t0 = out["emotion_control"]["wayne"]["take0"]
    assert t0["ld_z_emo_take1"] == 11 and t0["ld_z_emo_take2"] == 12 and "ld_z_emo_take0" not in t0
    m.style_transfer, m.emotion_control = True, False
    f1, f2 = "0_103_103", "0_104_104"                      # the two "fear" takes (dm/utils/ldm_evals.py:86)
    lab = lambda d: dict(d, ld_emo_label="fear")
    data = {"ayana": {f1: lab(mk(1)), f2: lab(mk(3))}, "scott": {f1: lab(mk(2)), f2: lab(mk(4))}}
    out = m.process_loader({"style_transfer": data, "style_transfer_info": "[ayana-scott]_[fear]"})
    # the reference's crosswise quirk: partner's EMO latent is filed under "sty" (infer_ldm.py:371-381)
    assert out["style_transfer"]["ayana"][f1]["ld_z_sty_scott"] == 12
    assert out["style_transfer"]["ayana"][f1]["ld_z_emo_scott"] == 22
    assert out["style_transfer"]["scott"][f2]["ld_z_sty_ayana"] == 13
    data["scott"][f2]["ld_emo_label"] = "happy"
    with pytest.raises(AssertionError, match="Emotion labels are not the same"):
        m.process_loader({"style_transfer": data, "style_transfer_info": "[ayana-scott]_[fear]"})
    # style x emotion transfer (infer_ldm.py:264-322): four takes named in the info string, emotion AND style swapped
    m.style_transfer, m.style_Xemo_transfer = False, True
    ta, th = "0_73_73", "0_65_65"
    la = lambda d, e: dict(d, ld_emo_label=e)
    data = {"lu": {ta: la(mk(1), "angry"), th: la(mk(2), "happy")}, "lawrence": {ta: la(mk(3), "angry"), th: la(mk(4), "happy")}}
    info = "[lu-lawrence]_[angry-happy]_*lu_angry_0_73_73*lu_happy_0_65_65*lawrence_angry_0_73_73*lawrence_happy_0_65_65*"
    out = m.process_loader({"style_Xemo_transfer": data, "style_Xemo_transfer_info": info})["style_Xemo_transfer"]
    assert out["takes"] == f"{ta}*{th}*{ta}*{th}"
    assert out["lu"][ta][f"ld_z_emo_lawrence_{th}"] == 14 and out["lu"][ta][f"ld_z_sty_lawrence_{th}"] == 24
    assert out["lawrence"][th][f"ld_z_emo_lu_{ta}"] == 11
    # a take with neither raw data nor latents is an error; unknown emotions as in mapinfo2takes
    m.style_Xemo_transfer, m.emotion_control = False, True
    with pytest.raises(KeyError):
        m.process_loader({"emotion_control": {"wayne": {"t": {"ld_z_con": 1}}}, "emotion_control_info": "[wayne]_[neutral]_first"})
    from amuse_amd.infer_ldm import mapinfo2takes
    assert mapinfo2takes("[a-b]_[sad]") == ["0_81_81", "0_82_82"]
    with pytest.raises(Exception):
        mapinfo2takes("[a-b]_[bored]")
    with pytest.raises(NotImplementedError):
        m.process_single_seq(torch.zeros(1, 160000))


def test_ast_checkpoint_choice_and_reader(tmp_path, monkeypatch):
    """infer_pretrained_ast_evp.py:21-39: best emotion accuracy (person accuracy for the identity ablation), epoch-0
    winners replaced by the first `_1_` file; the state dict is the checkpoint itself, keys `<enc>_enc.<name>`."""
    from collections import OrderedDict

    import torch

    from amuse_amd import audio_weights as aw, checkpoint as ckpt
    d = tmp_path / "wav_dtw_mfcc_x"
    d.mkdir()
    for name in ("model_e3_loss0.5_tEAcc0.71_tPAcc0.95.pt", "model_e7_loss0.4_tEAcc0.83_tPAcc0.60.pt", "experiment_args.json"):
        (d / name).write_bytes(b"")
    assert ckpt.pick_ast_checkpoint(d, "full").name.startswith("model_e7")
    assert ckpt.pick_ast_checkpoint(d, "identity").name.startswith("model_e3")
    (d / "model_e0_loss0.9_tEAcc0.99_tPAcc0.10.pt").write_bytes(b"")
    (d / "model_1_loss0.8_tEAcc0.20_tPAcc0.20.pt").write_bytes(b"")
    assert "_1_" in ckpt.pick_ast_checkpoint(d, "full").name
    with pytest.raises(AssertionError):
        ckpt.pick_ast_checkpoint(d, "bogus")
    tiny = OrderedDict([("v.cls_token", (1, 1, 4)), ("feature_head.1.weight", (2, 4))])
    monkeypatch.setattr(aw, "ast_param_spec", lambda: tiny)
    sds = {e: {k: np.full(s, i, np.float32) for k, s in tiny.items()} for i, e in enumerate(aw.ENCODERS)}
    path = ckpt.save_ast_reference_format(tmp_path / "out", sds)
    back = ckpt.load_ast_checkpoint(path)
    assert set(back) == {"con", "emo", "sty"} and float(back["sty"]["v.cls_token"].max()) == 2.0
    sd = torch.load(path, weights_only=False)
    del sd["emo_enc.v.cls_token"]
    torch.save(sd, path)
    with pytest.raises(KeyError):
        ckpt.load_ast_checkpoint(path)
