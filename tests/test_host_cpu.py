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
    # default = the actor's own shape vector, bit-equal to what the reference's writer put into its committed sample
    # outputs (tests/golden/sample_npz_betas.npz <- viz_dump/test/**/*_motion_smplx.npz; ldm_evals.py:348-379)
    gold = np.load(GOLDEN / "sample_npz_betas.npz")
    from amuse_amd.npz_writer import fetchbetas, subject2genderbeta
    for actor in ("scott", "miranda"):
        zz = np.load(write_sample(feats[:1], tmp_path / f"rst_{actor}", actor)[0], allow_pickle=True)
        assert zz["betas"].dtype == np.float64 and np.array_equal(zz["betas"], gold[actor]), actor
        assert str(zz["gender"]) == str(gold[actor + "_gender"])
        g_, b_ = subject2genderbeta(actor)
        assert g_.dtype == np.dtype("<U7") and np.array_equal(b_, gold[actor])
    assert np.array_equal(z["betas"], gold["scott"])
    with pytest.raises(NotImplementedError):                # zhang / jaime / kexin / hanieh have no fit (ldm_evals.py:362-376)
        fetchbetas("zhang")
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
    from amuse_amd.shard import job_plan
    assert [job_plan(n)["clips_per_group"] for n in (1, 128, 129, 256, 257, 4096)] == [1, 1, 2, 2, 3, 3]
    assert job_plan(4096, tokens=3)["clips_per_group"] == 5
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


def test_cli_config_is_merged_in_memory_like_scripts_main(tmp_path):
    """amuse_amd.main.load_config == scripts/main.py:243-265 without the write-back: override YAML deep-merged over the
    JSON (dicts recurse, leaves replace), the files on disk untouched."""
    import hashlib

    from conftest import make_reference_tree

    from amuse_amd.main import load_config, merge_dicts
    root = make_reference_tree(tmp_path / "amuse")
    before = {p.name: hashlib.sha256(p.read_bytes()).hexdigest() for p in (root / "configs").iterdir()}
    base, ldm = load_config(root, "infer_gesture")
    tp = base["TRAIN_PARAM"]
    assert tp["pretrained_infer"] is True and tp["wav_dtw_mfcc"]["ablation"] == "full"          # replaced leaves
    assert tp["wav_dtw_mfcc"]["dataset_std"] == 5.062332 and tp["seed"] == 2024                # untouched siblings
    assert tp["test"]["audio_list"]["use"] is True and tp["test"]["replication_times"] == 1
    assert tp["latent_diffusion"]["pretrained_lpdm"] == "LPDM_test" and tp["latent_diffusion"]["smplx_rep"] == "6D"
    assert base["DATA_PARAM"]["Bvh"] == {"train_pose_framelen": 300, "fps": 30, "bvh2smplbvh": False}
    assert ldm["scheduler"]["num_inference_timesteps"] == 50 and ldm["scheduler"]["steps_offset"] == 1  # diff_o.yaml
    assert ldm["scheduler"]["beta_end"] == 0.012 and ldm["noisy_scheduler"]["variance_type"] == "fixed_small"
    edit, _ = load_config(root, "edit_gesture")
    assert edit["TRAIN_PARAM"]["test"]["emotion_control_list"]["actor"] == "miranda"           # key absent from the JSON
    assert edit["TRAIN_PARAM"]["test"]["audio_list"]["use"] is False
    assert {p.name: hashlib.sha256(p.read_bytes()).hexdigest() for p in (root / "configs").iterdir()} == before
    assert merge_dicts({"a": {"b": 1, "c": 2}, "d": 3}, {"a": {"b": {"x": 1}}, "e": 4}) == {"a": {"b": {"x": 1}, "c": 2}, "d": 3, "e": 4}
    assert merge_dicts({"a": 1}, None) == {"a": 1}


class _StubLPDM:
    """Records diffusion_backward calls; poses carry the global clip index so that job -> clip mapping is checkable."""
    def __init__(self):
        self.device, self._clip_counter, self.calls = torch.device("cpu"), 0, []

    def diffusion_backward(self, bsz, z_con, z_emo, z_sty, clip_index0=None, return_latents=False):
        assert z_con.shape[0] == bsz
        c0 = self._clip_counter if clip_index0 is None else clip_index0
        if clip_index0 is None:
            self._clip_counter += bsz
        self.calls.append((bsz, c0, z_emo is None, z_sty is None))
        idx = torch.arange(c0, c0 + bsz, dtype=torch.float32)
        out = {"poses": idx[:, None, None, None] + z_con[:, :1, None, None].expand(bsz, 300, 55, 3) * 0,
               "trans": torch.zeros(bsz, 300, 3)}
        if return_latents:
            out["latents"] = idx[:, None].expand(bsz, 128)
        return out


def test_edit_task_job_lists_follow_the_reference_order():
    """emotion_control / style_transfer / style_Xemo_transfer job construction (trainer.py:559-631,705-772,839-901) and
    the batched runner's clip bookkeeping, on a stub model."""
    from amuse_amd.trainer import emotion_control_jobs, run_jobs, style_transfer_jobs, style_Xemo_transfer_jobs, subject_of
    g = torch.Generator().manual_seed(0)
    z = lambda n: torch.randn(n, 256, generator=g)
    attr = lambda a: (a, "male", "native", "x", "30")

    def take(a, n):
        return {"ld_z": torch.zeros(n, 128), "ld_z_con": z(n), "ld_z_emo": z(n), "ld_z_sty": z(n), "ld_attr": attr(a),
                "ld_wav": np.arange(n * 10000), "ld_motion": None}
    # --- emotion control: 8 takes, every take gets the other 7 emotions (infer_ldm.py:403-410)
    takes = ["0_9_9", "0_65_65", "0_73_73", "0_81_81", "0_87_87", "0_95_95", "0_103_103", "0_111_111"]
    data = {"wayne": {t: take("wayne", 2 if i else 3) for i, t in enumerate(takes)}}
    for t in takes:
        for o in takes:
            if o != t:
                data["wayne"][t][f"ld_z_emo_{o}"] = data["wayne"][o]["ld_z_emo"]
    jobs = emotion_control_jobs(data, "first")
    assert len(jobs) == 64
    assert jobs[0]["z_emo_key"] == "ld_z_emo" and jobs[0]["info"] == "wayne male native 30 yrs 9 original neutral"
    assert jobs[1]["info"] == "wayne male native 30 yrs 9 swap emo happy in element first"
    assert jobs[0]["bsz"] == 3 and jobs[1]["bsz"] == 2        # the swapped-in emotion latent is shorter ...
    assert jobs[7]["bsz"] == 2 and jobs[7]["z_con"].shape[0] == 2 and len(jobs[7]["audio"]) == 20000   # ... and it sticks
    assert subject_of(jobs[0]["info"]) == "wayne"
    m = _StubLPDM()
    m._clip_counter = 7
    rst = run_jobs(m, jobs, return_latents=True)
    total = sum(j["bsz"] for j in jobs)
    assert len(m.calls) == 1 and m.calls[0][:2] == (total, 7) and m._clip_counter == 7 + total
    off = 7
    for j, r in zip(jobs, rst):
        assert r["feats"].shape == (j["bsz"], 300, 168) and float(r["feats"][0, 0, 0]) == off and float(r["latents"][-1, 0]) == off + j["bsz"] - 1
        off += j["bsz"]
    m2 = _StubLPDM()
    m2._clip_counter = 7
    seq = run_jobs(m2, jobs, batched=False)
    assert len(m2.calls) == 64 and all(torch.equal(a["feats"], b["feats"]) for a, b in zip(rst, seq))
    # --- style transfer: "[lu-lawrence]" "[angry]" -> 8 jobs, originals then swaps per take; NB the crosswise keys
    st = {a: {t: take(a, 2) for t in ("0_73_73", "0_74_74")} for a in ("lu", "lawrence")}
    for t in ("0_73_73", "0_74_74"):
        for a, b in (("lu", "lawrence"), ("lawrence", "lu")):
            st[a][t][f"ld_z_sty_{b}"], st[a][t][f"ld_z_emo_{b}"] = st[b][t]["ld_z_emo"], st[b][t]["ld_z_sty"]
    sj = style_transfer_jobs(st, "[lu-lawrence]", "[angry]")
    assert [(j["actor"], j["take"], bool(j["swap_info"].startswith("Swapped"))) for j in sj] == \
        [("lu", "0_73_73", False), ("lawrence", "0_73_73", False), ("lu", "0_73_73", True), ("lawrence", "0_73_73", True),
         ("lu", "0_74_74", False), ("lawrence", "0_74_74", False), ("lu", "0_74_74", True), ("lawrence", "0_74_74", True)]
    assert sj[2]["info"] == "Style Transfer - lu male native 30 yrs 73 angry" and torch.equal(sj[2]["z_emo"], st["lu"]["0_73_73"]["ld_z_emo_lawrence"])
    assert sj[0]["swap_info"] == "Not swapped, original"
    # --- style X emotion transfer: "[scott-lu]" "[happy-angry]"
    sx = {a: {t: take(a, 1) for t in ("0_65_65", "0_73_73")} for a in ("scott", "lu")}
    t1, t2 = "0_65_65", "0_73_73"
    for (xa, xt), (ya, yt) in ((("scott", t1), ("lu", t2)), (("lu", t1), ("scott", t2)), (("scott", t2), ("lu", t1)), (("lu", t2), ("scott", t1))):
        sx[xa][xt][f"ld_z_emo_{ya}_{yt}"], sx[xa][xt][f"ld_z_sty_{ya}_{yt}"] = sx[ya][yt]["ld_z_emo"], sx[ya][yt]["ld_z_sty"]
    sx["takes"] = f"{t1}*{t2}*{t1}*{t2}"
    xj = style_Xemo_transfer_jobs(sx, "[scott-lu]", "[happy-angry]")
    assert len(xj) == 8 and xj[2]["swap_info"].startswith("Swapped") and torch.equal(xj[2]["z_sty"], sx["lu"][t2]["ld_z_sty"])
    assert xj[7]["info"] == "Style X Emo Transfer - lu male native 30 yrs 73 happy-angry"
    # jobs that drop a token are launched apart, clip indices still in job order
    mixed = [dict(sj[0]), dict(sj[1], z_emo=None), dict(sj[2])]
    m3 = _StubLPDM()
    r3 = run_jobs(m3, mixed)
    assert sorted(c[1] for c in m3.calls) == [0, 2, 4] and [float(r["feats"][0, 0, 0]) for r in r3] == [0.0, 2.0, 4.0]


def test_shard_range_properties_hypothesis():
    """Property form of the sharding contract (hypothesis): for any job size, world size and tile alignment the rank ranges
    partition [0, total) in rank order, every boundary but the last is a multiple of the alignment (so a clip keeps its slot in
    its tile - what makes shards bitwise the single-GPU result), and the ranks' loads differ by at most one aligned unit."""
    from hypothesis import given, settings, strategies as st
    from amuse_amd.shard import shard_range, job_plan

    @settings(max_examples=300, deadline=None)
    @given(st.integers(0, 5000), st.integers(1, 16), st.integers(1, 5))
    def check(total, world, align):
        ranges = [shard_range(total, r, world, align=align) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a0 <= a1
        for lo, hi in ranges:
            assert lo % align == 0 or lo == total
            assert hi % align == 0 or hi == total
        units = [-(-(hi - lo) // align) for lo, hi in ranges]
        assert max(units) - min(units) <= 1
    check()

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 100000), st.integers(3, 5))
    def tiling(total, tokens):
        g = job_plan(total, tokens)["clips_per_group"]
        assert 1 <= g <= 16 // tokens
        assert g == 16 // tokens or total <= 128 * g          # more clips per tile only once 128 tiles are full
    tiling()


def test_bench_reports_pmc_traffic_only_for_the_kernel_that_was_counted(monkeypatch):
    """bench.py's `roofline.traffic` comes from committed PMC passes, and only while the sampler of this tree IS the one that was counted: the same build inputs
    (tools/kernel_id.py: source closure + Makefile) or, failing that, the same object bits; anything else gives (None, reason)."""
    import importlib
    import json
    sys.path.insert(0, str(REPO))
    sys.path.insert(0, str(REPO / "tools"))
    bench, kid = importlib.import_module("bench"), importlib.import_module("kernel_id")
    d = next(x for x in bench.PMC_DIRS["bf16"] if (REPO / x / "kernel_id.json").exists())
    counted = json.load(open(REPO / d / "kernel_id.json"))["k_sample8"]
    monkeypatch.setattr(kid, "kernel_id", lambda name: dict(counted))                                   # the counted kernel itself
    v, why = bench.pmc_traffic_bytes(256, 1000, "bf16")
    assert isinstance(v, int) and v > 1 << 20 and "source id" in why
    monkeypatch.setattr(kid, "kernel_id", lambda name: {**counted, "source_sha256": "0" * 64})         # a header edit that left the object as it was
    v, why = bench.pmc_traffic_bytes(256, 1000, "bf16")
    assert isinstance(v, int) and "object" in why
    monkeypatch.setattr(kid, "kernel_id", lambda name: {"source_sha256": "0" * 64, "object_sha256": "1" * 64})   # another kernel
    v, why = bench.pmc_traffic_bytes(256, 1000, "bf16")
    assert v is None and "retake" in why
    monkeypatch.setattr(kid, "kernel_id", lambda name: {"source_sha256": "0" * 64})                     # no object at hand, other sources
    assert bench.pmc_traffic_bytes(256, 1000, "bf16")[0] is None
    assert bench.pmc_traffic_bytes(64, 1000, "bf16")[0] is None                                          # no pass at that shape


def test_job_plan_known_values():
    """amuse_plan (the library's own rule, through shard.job_plan - nothing in Python restates it; tests/test_plan_cpu.py sweeps it against what the entry points
    take): known values of the fp32x rule - staged below 64 clips, the no-split-K kernels from 64, the per-clip decoder ("clip") where the job's clips fill rounds of
    the chip's 256 CUs - and its per-precision forms."""
    from amuse_amd import _lib, shard
    want = {1: "staged", 63: "staged", 64: "fused", 159: "fused", 160: "clip", 256: "clip", 257: "fused", 419: "fused", 420: "clip", 512: "clip", 513: "fused",
            625: "fused", 626: "clip", 768: "clip", 831: "fused", 832: "clip", 1024: "clip", 1025: "fused", 1038: "clip", 4096: "clip"}
    assert {n: shard.job_plan(n)["decode_path"] for n in want} == want
    assert all(shard.job_plan(n)["decode_path"] == "clip" for n in range(1280, 4097, 7))          # from the sixth round on every count qualifies
    for n, w in want.items():
        assert shard.job_plan(n, precision=_lib.PREC_F32)["decode_path"] == "staged"
        for prec in (_lib.PREC_BF16, _lib.PREC_F16):
            p = shard.job_plan(n, precision=prec)
            assert p["decode_path"] == ("staged" if n < 64 else "fused") and p["encode_path"] == "staged"
        px = shard.job_plan(n)
        assert px["encode_path"] == w and px["step_path"] == "staged"                               # (the latent Denoiser has no per-step choice)
        assert shard.job_plan(n, arch=_lib.ARCH_ENC_POSE)["step_path"] == w and shard.job_plan(n, arch=_lib.ARCH_ENC_POSE)["clips_per_group"] == 1
        assert shard.job_plan(n, arch=_lib.ARCH_DEC_POSE)["step_path"] == "staged"
    with pytest.raises(_lib.AmuseHipError):
        shard.job_plan(0)
    with pytest.raises(_lib.AmuseHipError):
        shard.job_plan(8, tokens=6)
