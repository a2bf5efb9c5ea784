"""CPU tests of the train_gesture path (BASELINE config 4): the autograd modules against the reference's golden vectors,
the loss set, one optimisation step, the checkpoint writer, and the data-parallel gradient exchange (2 ranks, gloo)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO


@pytest.fixture(scope="module")
def nets():
    from amuse_amd import weights as wts
    from amuse_amd.nn_modules import Denoiser, MotionPrior, load_numpy_state
    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    return load_numpy_state(Denoiser(), wd).eval(), load_numpy_state(MotionPrior(), wp).eval(), wd, wp


def test_autograd_modules_are_the_reference_networks(nets):
    """Same state-dict keys / shapes as the reference modules and, in eval mode, the same numbers as the golden vectors
    the reference's own Denoiser / MotionPrior produced (oracle/gen_golden.py)."""
    d, p, wd, wp = nets
    spec = json.load(open(GOLDEN / "state_dict_spec.json"))
    assert {k: list(v.shape) for k, v in d.state_dict().items()} == {k: list(v) for k, v in spec["denoiser"].items()}
    assert {k: list(v.shape) for k, v in p.state_dict().items()} == {k: list(v) for k, v in spec["prior"].items()}
    assert sum(v.numel() for v in d.parameters()) + sum(v.numel() for v in p.parameters()) == 6835661
    g = np.load(GOLDEN / "denoiser_steps.npz")
    x, con, emo, sty = (torch.from_numpy(g[k]) for k in ("x_t", "con", "emo", "sty"))
    err = lambda a, b: float((a - torch.from_numpy(b)).abs().max())
    with torch.no_grad():
        for t in (981, 501, 1):
            assert err(d(x[:, None], t, con, emo, sty)[0][:, 0], g[f"eps_t{t}"]) < 1e-5
        assert err(d(x[:, None], 501, con, None, sty)[0][:, 0], g["eps_t501_noemo"]) < 1e-5
        assert err(d(x[:, None], 501, con, None, None)[0][:, 0], g["eps_t501_consolo"]) < 1e-5
        assert err(d(x[:, None], torch.from_numpy(g["timesteps_batch"]), con, emo, sty)[0][:, 0], g["eps_batch_t"]) < 1e-5
        gv = np.load(GOLDEN / "vae_decode.npz")
        z = torch.from_numpy(gv["z"])
        assert err(p.decode(z[None], [300] * 3), gv["feats"]) < 2e-5
        assert err(p.decode(z[None, :2], [300, 173]), gv["feats_ragged"]) < 2e-5
        ge = np.load(GOLDEN / "vae_encode.npz")
        fe = torch.from_numpy(ge["feats"].astype(np.float32))
        lat, dist = p.encode(fe, [300] * fe.shape[0])
        assert lat.shape == (1, fe.shape[0], 128)
        assert err(dist.loc[0], ge["mu"]) < 2e-5 and err(dist.scale[0], ge["std"]) < 2e-5
        _, dr = p.encode(fe, [int(v) for v in ge["lengths_ragged"]])
        assert err(dr.loc[0], ge["mu_ragged"]) < 2e-5


def test_motion_to_feats_matches_oracle():
    from amuse_amd.train_gesture import motion_to_feats
    from oracle import amuse_oracle as orc
    g = torch.Generator().manual_seed(0)
    m = torch.cat([0.7 * torch.randn(2, 5, 165, generator=g), torch.randn(2, 5, 3, generator=g)], -1)
    m[0, 0, :3] = 0.0                                                       # the small-angle branch
    ref = torch.cat([orc.axis_angle_to_rotation_6d(m[..., :165].reshape(2, 5, 55, 3)).reshape(2, 5, 330), m[..., 165:]], -1)
    assert float((motion_to_feats(m) - ref).abs().max()) < 1e-6


def test_loss_set_and_one_training_step(tmp_path):
    from amuse_amd import checkpoint as ckpt
    from amuse_amd.train_gesture import LatentPriorLosses, build_trainer, synthetic_batch
    torch.manual_seed(0)
    tr = build_trainer("cpu", use_hip_sampler=False)
    assert tr.n_grad_elements() == 6835661 and tr.lpdm_losses.losses == ["inst_loss", "recons_feature", "recons_joints", "kl_motion",
                                                                         "gen_feature", "gen_joints", "total"]
    batch = synthetic_batch(2, 1)
    before = {k: v.detach().clone() for k, v in tr.model["ldm"].state_dict().items()}
    loss = tr.train_step(batch)
    ld = {k: float(v) for k, v in tr.lpdm_losses.compute().items()}
    assert np.isfinite(float(loss)) and abs(ld["total"] - float(loss)) < 1e-6
    # total = recons + 1e-4 kl + inst (no inner sampler here -> no gen_feature term)   (latent_losses.py:101-151)
    assert abs(ld["total"] - (ld["recons_feature"] + 1e-4 * ld["kl_motion"] + ld["inst_loss"])) < 1e-5 * max(1.0, ld["total"])
    after = tr.model["ldm"].state_dict()
    assert not torch.equal(before["denoiser.encoder.norm.weight"], after["denoiser.encoder.norm.weight"])
    assert torch.equal(before["denoiser.mem_pos.pe"], after["denoiser.mem_pos.pe"])   # never reached: no grad, no weight decay
    # every p.grad is a view of the one flat bucket
    off = 0
    for p in tr.params:
        assert p.grad.data_ptr() == tr.flat_grad.data_ptr() + 4 * off
        off += p.numel()
    # ... and every parameter a view of one flat buffer that IS the two images the HIP library's re-pack takes (state-dict order)
    from amuse_amd import weights as wts
    from amuse_amd.engine import flatten_on_device
    assert torch.equal(tr.flat_param[:tr.n_prior], flatten_on_device(tr.model["prior"].state_dict(), wts.prior_param_spec()))
    assert torch.equal(tr.flat_param[tr.n_prior:], flatten_on_device(tr.model["ldm"].denoiser.state_dict(), wts.denoiser_param_spec()))
    assert all(q.data_ptr() >= tr.flat_param.data_ptr() and q.data_ptr() < tr.flat_param.data_ptr() + 4 * tr.flat_param.numel() for q in tr.params)
    # checkpoint writer: the reference's file names; the readers pick them up and return the trained weights
    p1, p2 = tr.save_checkpoint(tmp_path / "LPDM_x", epoch=199)
    pat = r"_recF\d+\.\d{4}_recJ\d+\.\d{4}_kl\d+\.\d{4}_genF\d+\.\d{4}_genJ\d+\.\d{4}_instL\d+\.\d{4}_vtexR\d+\.\d{4}_vtexG\d+\.\d{4}_total\d+\.\d{4}_e200\.pt$"
    assert p1.name.startswith("prior_model_NoOpt") and p2.name.startswith("latdiff_model_wOpt") and re.search(pat, p1.name) and re.search(pat, p2.name)
    lat = ckpt.pick_checkpoint(tmp_path / "LPDM_x", "latdiff", "best")
    assert lat == p2 and ckpt.epoch_of(lat) == 200
    dsd = ckpt.load_denoiser_checkpoint(lat)
    assert np.array_equal(dsd["encoder.norm.weight"], after["denoiser.encoder.norm.weight"].numpy())
    psd = ckpt.load_prior_checkpoint(ckpt.pick_checkpoint(tmp_path / "LPDM_x", "prior", 200))
    assert np.array_equal(psd["final_layer.bias"], tr.model["prior"].state_dict()["final_layer.bias"].numpy())
    assert set(torch.load(p2, weights_only=False)) == {"epoch", "model_state_dict", "optimizer_state_dict"}
    # the term set follows the config like latent_losses.py:36-56
    assert LatentPriorLosses({"stage": "diffusion"}).losses == ["inst_loss", "total"]
    with pytest.raises(NotImplementedError):
        LatentPriorLosses({"vtex_displacement": True})
    with pytest.raises(RuntimeError):
        build_trainer("cpu", use_hip_sampler=True)                          # the in-loop sampler has no CPU path


def test_lean_attention_paths_equal_nn_multihead_attention():
    """nn_modules.mha_self / mha_one_key (batch-first, no transposed copies, value path only for a one-token memory) against
    torch's own nn.MultiheadAttention on the sequence-first layout of the reference (cross_attention.py:259-262,323-331):
    outputs and every gradient, with a ragged key-padding mask; and MotionPrior with ragged lengths against full lengths."""
    from amuse_amd.nn_modules import MotionPrior, mha_one_key, mha_self
    torch.manual_seed(3)
    attn = torch.nn.MultiheadAttention(128, 4, dropout=0.0)
    x = torch.randn(3, 11, 128, requires_grad=True)
    kpm = torch.zeros(3, 11, dtype=torch.bool)
    kpm[1, 7:] = True
    kpm[2, 10:] = True
    w = torch.randn(3, 11, 128)

    def grads(out):
        attn.zero_grad()
        x.grad = None
        (out * w).sum().backward()
        return [x.grad.clone()] + [p.grad.clone() for p in attn.parameters()]
    xs = x.transpose(0, 1)
    ref = attn(xs, xs, xs, key_padding_mask=kpm, need_weights=False)[0].transpose(0, 1)
    got = mha_self(attn, x, kpm)
    assert float((ref - got).detach().abs().max()) < 2e-6
    for a, b in zip(grads(ref), grads(got)):
        assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(a.abs().max()))
    assert float((mha_self(attn, x, None) - attn(xs, xs, xs, need_weights=False)[0].transpose(0, 1)).detach().abs().max()) < 2e-6
    mem = torch.randn(3, 1, 128, requires_grad=True)
    ms = mem.transpose(0, 1)
    ref = attn(xs, ms, ms, need_weights=False)[0].transpose(0, 1)
    got = mha_one_key(attn, x, mem)
    assert float((ref - got).detach().abs().max()) < 2e-6
    g_ref = torch.autograd.grad((ref * w).sum(), [mem] + list(attn.parameters()), allow_unused=True)
    g_got = torch.autograd.grad((got * w).sum(), [mem] + list(attn.parameters()), allow_unused=True)
    for a, b in zip(g_ref, g_got):
        b = torch.zeros_like(a) if b is None else b
        assert float((a - b).abs().max()) < 1e-5 * max(1.0, float(a.abs().max()))
    # training mode: the one-key path drops whole (clip, query, head) value vectors with probability p and rescales the rest
    attn_p = torch.nn.MultiheadAttention(128, 4, dropout=0.25).train()
    o = mha_one_key(attn_p, torch.zeros(2, 500, 128), torch.randn(2, 1, 128))
    base = mha_one_key(attn_p.eval(), torch.zeros(2, 500, 128), torch.randn(2, 1, 128))
    assert o.shape == base.shape == (2, 500, 128)
    # ragged lengths: frames beyond a clip's length are masked as keys and zeroed in the output; the valid part of a clip does not
    # depend on how long the OTHER clips of the batch are
    prior = MotionPrior().eval()
    f = torch.randn(2, 12, 333)
    with torch.no_grad():
        _, d = prior.encode(f, [9, 12])
        assert float((d.loc[:, :1] - prior.encode(f[:1, :9], [9])[1].loc).abs().max()) < 1e-5
        out = prior.decode(d.loc, [9, 12])
        assert out.shape == (2, 12, 333) and float(out[0, 9:].abs().max()) == 0.0 and float(out[1, 9:].abs().min()) > 0.0
        assert float((out[0, :9] - prior.decode(d.loc[:, :1], [9])[0]).abs().max()) < 1e-5


_DP_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from amuse_amd.train_gesture import build_trainer, synthetic_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
B = 2                                                   # per rank
full = synthetic_batch(B * world, 5)
g = torch.Generator().manual_seed(9)
noise = torch.randn(B * world, 1, 128, generator=g)
ts = torch.randint(0, 1000, (B * world,), generator=g)
e1, e2 = torch.randn(1, B * world, 128, generator=g), torch.randn(1, B * world, 128, generator=g)
sl = slice(rank * B, (rank + 1) * B)
shard = {k: (v[sl] if torch.is_tensor(v) else v[sl]) for k, v in full.items()}
tr = build_trainer("cpu", rank, world, use_hip_sampler=False, dropout=0.0)
loss = tr.forward_losses(shard, noise=noise[sl], timesteps=ts[sl], eps_enc=e1[:, sl], eps_inf=e2[:, sl])
tr.flat_grad.zero_(); loss.backward(); tr.allreduce_gradients()
if rank == 0:
    ref = build_trainer("cpu", 0, 1, use_hip_sampler=False, dropout=0.0)
    l1 = ref.forward_losses(full, noise=noise, timesteps=ts, eps_enc=e1, eps_inf=e2)
    ref.flat_grad.zero_(); l1.backward()
    d = (tr.flat_grad - ref.flat_grad).abs().max().item()
    scale = ref.flat_grad.abs().max().item()
    assert d < 2e-6 * max(1.0, scale), (d, scale)      # mean over the global batch == average of the per-rank means
    assert ref.flat_grad.abs().sum().item() > 0
    # after the exchange + the same AdamW step every rank holds the same weights
tr.lpdm_opt.step()
w = torch.cat([p.detach().flatten() for p in tr.params])
ws = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(ws, w)
assert all(torch.equal(ws[0], x) for x in ws)
# the same through train_step itself (gradients handed over, packed into the bucket, exchanged, optimizer): different data and
# draws per rank, identical weights afterwards
torch.manual_seed(100 + rank)
tr.train_step(synthetic_batch(B, 50 + rank))
assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(tr.params, tr.views))
w = tr.flat_param.detach().clone()
ws = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(ws, w)
assert all(torch.equal(ws[0], x) for x in ws) and tr.pop_allreduce_ms() == []
if rank == 0:
    print("DP_OK")
dist.destroy_process_group()
'''


def test_two_rank_gloo_gradients_equal_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(REPO)], env=dict(env, RANK=str(r), WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DP_OK" in outs[0]


def test_latent_diffusion_cache_reader_and_collate():
    """amuse_amd/dataload.py against an in-memory environment with the reference's key / tuple layout (dataload.py:250-271,287-308)."""
    import pickle
    import numpy as np
    from amuse_amd.dataload import LatentDiffusionCache, latdiff_long_collate_fn_v1, make_loader
    from amuse_amd.train_gesture import build_trainer

    rng = np.random.default_rng(0)
    store = {}
    for i in range(6):
        sample = (0.1 * rng.standard_normal((300, 168)).astype(np.float32), ("scott", "male"), np.array(i % 8), rng.standard_normal(1000 + 37 * i).astype(np.float32),
                  rng.standard_normal(256).astype(np.float32), rng.standard_normal(256).astype(np.float32), rng.standard_normal(256).astype(np.float32))
        store["{:005}".format(i).encode("ascii")] = pickle.dumps(sample)

    class Txn:
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def get(self, k): return store.get(k)
        def stat(self): return {"entries": len(store)}

    class Env:
        def begin(self, write=False): return Txn()

    ds = LatentDiffusionCache(Env(), pickle.loads)
    assert len(ds) == 6
    it = ds[3]
    assert it["ld_motion"].shape == (300, 168) and it["ld_motion"].dtype == torch.float32 and it["ld_emo_label"].dtype == torch.int64
    assert it["ld_attr"] == ("scott", "male") and it["ld_audio"].shape == (1000 + 37 * 3,)
    with pytest.raises(IndexError):
        ds[6]
    b = latdiff_long_collate_fn_v1([ds[i] for i in (0, 5, 2)])
    assert b["ld_motion"].shape == (3, 300, 168) and b["ld_audio"].shape == (3, 1185) and b["ld_audio_length"].tolist() == [1000, 1185, 1074]
    assert float(b["ld_audio"][0, 1000:].abs().max()) == 0.0 and b["ld_audio_con"].shape == (3, 256) and len(b["ld_attr"]) == 3
    # two ranks read disjoint shards; a batch from the loader drives the training step
    seen = [sorted(int(x) for bt in make_loader(ds, 1, rank=r, world=2, seed=1) for x in bt["ld_emo_label"]) for r in (0, 1)]
    assert len(seen[0]) == len(seen[1]) == 3 and sorted(seen[0] + seen[1]) == sorted(i % 8 for i in range(6))
    tr = build_trainer("cpu", use_hip_sampler=False)
    loss = tr.train_step(next(iter(make_loader(ds, 2, shuffle=False))))
    assert bool(torch.isfinite(loss))


def test_ablation_kind_reaches_the_trainer_and_freezes_the_dropped_projection():
    """The ablation variants of the LMDB id (trainer.py:396-401): `identity` trains without the emotion embedding, `emotion` /
    `baseline` without the style one - the projection of the condition that is never fed gets no gradient in the reference and,
    under zero_grad(set_to_none=True), no weight decay either: it must stay bit-identical over optimizer steps (advisor finding)."""
    from amuse_amd.train_gesture import ablation_kind, build_trainer, synthetic_batch
    assert ablation_kind("BEAT-cache/2023-10-28_30F_fing_smplx_MOSH_full_v1_feat_based_300") == "full"
    assert ablation_kind("BEAT-cache/2023-10-28_30F_fing_smplx_MOSH_identity_v1_feat_based_300") == "identity"
    assert ablation_kind("x/2023_smplx_emotion_v1_300") == "emotion" and ablation_kind(None) is None
    with pytest.raises(AssertionError, match="Invalid lmdb_id"):
        ablation_kind("BEAT-cache/none")
    torch.manual_seed(0)
    for kind, frozen, live in (("identity", "emb_proj_emo", "emb_proj_sty"), ("emotion", "emb_proj_sty", "emb_proj_emo")):
        tr = build_trainer("cpu", use_hip_sampler=False, kind=kind, lr=3e-4)
        assert tr.kind == kind and abs(tr.lpdm_opt.param_groups[0]["lr"] - 3e-4) < 1e-12
        sd0 = {k: v.detach().clone() for k, v in tr.model["ldm"].state_dict().items()}
        for i in range(2):
            tr.train_step(synthetic_batch(2, 10 + i))
        sd1 = tr.model["ldm"].state_dict()
        for leaf in ("weight", "bias"):
            assert torch.equal(sd0[f"denoiser.{frozen}.1.{leaf}"], sd1[f"denoiser.{frozen}.1.{leaf}"]), (kind, frozen)
            assert not torch.equal(sd0[f"denoiser.{live}.1.{leaf}"], sd1[f"denoiser.{live}.1.{leaf}"]), (kind, live)
    full = build_trainer("cpu", use_hip_sampler=False, kind="full")
    assert full.kind is None


def test_train_mode_inner_sampler_is_the_references_loop():
    """TrainModeInnerSampler (opt-in, `--inner-sampler train`): the no-gradient DDIM-50 + decode of an iteration on the trainer's own modules.
    eval mode: its latents / features are the oracle's diffusion_backward on the same initial noise (<= 1e-4: the same loop, no dropout);
    train mode (what the reference's training loop runs, trainer.py:357-358,413-415): dropout is live - another result, fresh masks on every call -
    and the trainer takes a step with it as the gen_feature term's input."""
    from amuse_amd.train_gesture import TrainModeInnerSampler, build_trainer, synthetic_batch
    from oracle import amuse_oracle as orc
    from amuse_amd import weights as wts
    torch.manual_seed(0)
    tr = build_trainer("cpu", inner="train")
    s = tr.inner_sampler
    assert isinstance(s, TrainModeInnerSampler) and s.serial and list(s.table.timesteps[:3]) == [981, 961, 941]
    g = torch.Generator().manual_seed(3)
    con, emo, sty, x0 = (torch.randn(2, n, generator=g) for n in (256, 256, 256, 128))
    for m in tr.model.values():
        m.eval()
    feats, lat = s(con, emo, sty, 2, x_init=x0, return_latents=True)
    ref = orc.diffusion_backward(orc.to_torch(wts.make_denoiser_weights(0)), orc.to_torch(wts.make_prior_weights(0)), orc.DDIM(), con, emo, sty, x0)
    assert float((lat - ref["latents"]).abs().max()) < 1e-4 and float((feats - ref["feats"]).abs().max()) < 2e-4
    for m in tr.model.values():
        m.train()
    f1 = s(con, emo, sty, 2, x_init=x0)
    f2 = s(con, emo, sty, 2, x_init=x0)
    assert float((f1 - feats).abs().max()) > 1e-3 and float((f1 - f2).abs().max()) > 1e-3 and bool(torch.isfinite(f1).all())
    # drawn latents: seeded, per global clip index, advancing
    a = s.initial_latents(2)
    b = s.initial_latents(2)
    assert a.shape == (2, 128) and not torch.equal(a, b)
    loss = tr.train_step(synthetic_batch(2, 1))
    ld = {k: float(v) for k, v in tr.lpdm_losses.compute().items()}
    assert np.isfinite(float(loss)) and ld["gen_feature"] > 0
    with pytest.raises(ValueError):
        build_trainer("cpu", inner="bogus")
