"""CPU: the oracle's restatement of the Denoiser VARIANTS (oracle/amuse_oracle.py denoiser_forward_variant) against vectors produced
by the reference's own Denoiser class built with arch = "trans_dec" and / or diffusion_only = true (tests/golden/denoiser_variants.npz,
written by oracle/gen_golden.py).  Reference: models/latent_diffusion/denoiser.py:64-66,116-131,174-204,
utils/cross_attention.py:195-234,297-345."""
import json

import numpy as np
import pytest
import torch

from amuse_amd import weights as wts
from oracle import amuse_oracle as orc
from conftest import GOLDEN

VARIANTS = (("trans_dec", False), ("trans_enc", True), ("trans_dec", True))
ROWS = slice(0, 300, 6)   # the fixture keeps every 6th frame of the pose-space outputs (oracle/gen_golden.py POSE_ROWS)


def tag_of(arch, pose):
    return f"{arch}{'_pose' if pose else ''}"


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN / "denoiser_variants.npz")


def inputs(g, pose):
    con, emo, sty = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty"))
    x = torch.from_numpy(g["x_pose"].astype(np.float32)) if pose else torch.from_numpy(g["x_lat"])
    return con, emo, sty, x


def cut(e, pose):
    return e[:, ROWS].numpy() if pose else e.numpy()


def test_variant_state_dict_specs_match_reference():
    spec = json.load(open(GOLDEN / "state_dict_spec_variants.json"))
    want = {"trans_dec": 2657536, "trans_enc_pose": 2278093, "trans_dec_pose": 2743245}
    for arch, pose in VARIANTS:
        mine, ref = wts.denoiser_param_spec(arch, pose), spec[tag_of(arch, pose)]
        assert list(ref.keys()) == list(mine.keys())
        assert all(tuple(ref[k]) == tuple(s) for k, s in mine.items())
        assert wts.n_params(wts.make_denoiser_weights(0, arch, pose)) == want[tag_of(arch, pose)]
        assert wts.arch_of_id(wts.arch_id(arch, pose)) == (arch, pose)
    # the variants' generators are independent of the shipped configuration's
    assert not np.array_equal(wts.make_denoiser_weights(0)["time_embedding.linear_1.weight"],
                              wts.make_denoiser_weights(0, "trans_dec")["time_embedding.linear_1.weight"])


@pytest.mark.parametrize("arch,pose", VARIANTS)
def test_variant_eps_vs_reference_module(g, arch, pose):
    W = orc.to_torch(wts.make_denoiser_weights(0, arch, pose))
    con, emo, sty, x = inputs(g, pose)
    tag = tag_of(arch, pose)
    for t in (981, 501, 1):
        taps = {}
        eps = orc.denoiser_forward_variant(W, x, t, con, emo, sty, arch, pose, taps=taps)
        assert np.abs(cut(eps, pose) - g[f"{tag}/eps_t{t}"]).max() < 1e-5, t
        if t == 981 and arch == "trans_dec":
            assert np.abs(taps["memory"].numpy() - g[f"{tag}/tap981/memory"]).max() < 3e-6
            for k in ("tokens", "decoder.layers.0", "decoder.layers.8"):
                assert np.abs(cut(taps[k], pose) - g[f"{tag}/tap981/{k}"]).max() < 1e-5, k
    e = orc.denoiser_forward_variant(W, x, 501, con, None, sty, arch, pose)
    assert np.abs(cut(e, pose) - g[f"{tag}/eps_t501_noemo"]).max() < 1e-5
    e = orc.denoiser_forward_variant(W, x, 501, con, None, None, arch, pose)
    assert np.abs(cut(e, pose) - g[f"{tag}/eps_t501_consolo"]).max() < 1e-5
    e = orc.denoiser_forward_variant(W, x, [int(v) for v in g["timesteps_batch"]], con, emo, sty, arch, pose)
    assert np.abs(cut(e, pose) - g[f"{tag}/eps_batch_t"]).max() < 1e-5
    if pose:
        lens = [int(v) for v in g["lengths_ragged"]]
        e = orc.denoiser_forward_variant(W, x, 501, con, emo, sty, arch, pose, lengths=lens)
        got = cut(e, pose)
        assert np.abs(got - g[f"{tag}/eps_t501_ragged"]).max() < 1e-5
        assert np.all(e[1, lens[1]:].numpy() == 0) and np.any(e[1, lens[1] - 1].numpy() != 0)
        # the padded frames are still attended keys (no key mask, denoiser.py:182): the valid rows equal the full-length result
        full = orc.denoiser_forward_variant(W, x, 501, con, emo, sty, arch, pose)
        assert np.array_equal(e[1, :lens[1]].numpy(), full[1, :lens[1]].numpy())


@pytest.mark.parametrize("arch,pose", VARIANTS)
def test_variant_ddim50_trajectory(g, arch, pose):
    W = orc.to_torch(wts.make_denoiser_weights(0, arch, pose))
    con, emo, sty, x = inputs(g, pose)
    tag = tag_of(arch, pose)
    sched = orc.DDIM()
    traj = []
    orc.sample_variant(W, sched, con, emo, sty, x, arch, pose, traj=traj)
    for n in (10, 50):
        assert np.abs(cut(traj[n - 1], pose) - g[f"{tag}/x_after_{n}"]).max() < 1e-4, n


def test_trans_dec_single_token_self_attention_is_the_value_path(g):
    """With ONE target token the decoder layer's self-attention is softmax over one key = 1: out_proj(v_proj(x)) - what the HIP
    kernel computes (q / k projections unused)."""
    W = orc.to_torch(wts.make_denoiser_weights(0, "trans_dec", False))
    x = torch.from_numpy(g["x_lat"])[:, None]
    p = "decoder.layers.3.self_attn"
    full = orc.mha(orc.Ops(), x, x, W, p)
    w, b = W[p + ".in_proj_weight"], W[p + ".in_proj_bias"]
    v = x @ w[256:].T + b[256:]
    short = v @ W[p + ".out_proj.weight"].T + W[p + ".out_proj.bias"]
    assert (full - short).abs().max() < 2e-6
