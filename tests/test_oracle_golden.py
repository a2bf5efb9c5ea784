"""CPU: the oracle restatement (oracle/amuse_oracle.py) against vectors produced by the REFERENCE's own
modules (tests/golden/, written by oracle/gen_golden.py).  This is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

from amuse_amd import weights as wts
from oracle import amuse_oracle as orc
from conftest import GOLDEN


@pytest.fixture(scope="module")
def Wd():
    return orc.to_torch(wts.make_denoiser_weights(0))


@pytest.fixture(scope="module")
def Wp():
    return orc.to_torch(wts.make_prior_weights(0))


def test_state_dict_spec_matches_reference():
    spec = json.load(open(GOLDEN / "state_dict_spec.json"))
    for tag, mine in (("denoiser", wts.denoiser_param_spec()), ("prior", wts.prior_param_spec())):
        assert list(spec[tag].keys()) == list(mine.keys())
        for k, s in mine.items():
            assert tuple(spec[tag][k]) == tuple(s), k
    assert len(spec["denoiser"]) == 130  # infer_ldm.py:103 asserts this count
    assert wts.n_params(wts.make_denoiser_weights(0)) == 2192384
    assert wts.n_params(wts.make_prior_weights(0)) == 4643277


def test_weights_are_deterministic():
    a, b = wts.make_denoiser_weights(0), wts.make_denoiser_weights(0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = wts.make_denoiser_weights(1)
    assert not np.array_equal(a["encoder.norm.weight"], c["encoder.norm.weight"])


def test_denoiser_eps_and_taps(Wd):
    g = np.load(GOLDEN / "denoiser_steps.npz")
    con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_t"))
    for t in (981, 501, 1):
        taps = {}
        eps = orc.denoiser_forward(Wd, x, t, con, emo, sty, taps=taps)
        assert np.abs(eps.numpy() - g[f"eps_t{t}"]).max() < 1e-5
        if t == 981:
            assert np.abs(orc.time_embed(Wd, t).numpy()[None] - g["tap981/time_embedding"]).max() < 2e-6
            assert np.abs(orc.cond_project(Wd, "emo", emo).numpy()[:, None] - g["tap981/emb_proj_emo"]).max() < 2e-6
            for k in ("tokens", "encoder.input_blocks.0", "encoder.middle_block", "encoder.output_blocks.3"):
                assert np.abs(taps[k].numpy() - g[f"tap981/{k}"]).max() < 1e-5, k
    e4 = orc.denoiser_forward(Wd, x, 501, con, None, sty)
    e3 = orc.denoiser_forward(Wd, x, 501, con, None, None)
    assert np.abs(e4.numpy() - g["eps_t501_noemo"]).max() < 1e-5
    assert np.abs(e3.numpy() - g["eps_t501_consolo"]).max() < 1e-5


def test_denoiser_fp64_agrees(Wd):
    g = np.load(GOLDEN / "denoiser_steps.npz")
    W64 = {k: v.double() for k, v in Wd.items()}
    con, emo, sty, x = (torch.from_numpy(g[k]).double() for k in ("con", "emo", "sty", "x_t"))
    eps = orc.denoiser_forward(W64, x, 501, con, emo, sty)
    assert np.abs(eps.numpy() - g["eps_t501"]).max() < 1e-5


def test_ddim50_trajectory(Wd):
    g = np.load(GOLDEN / "ddim50_traj.npz")
    sched = orc.DDIM()
    assert sched.timesteps == list(range(981, 0, -20))
    assert sched.init_noise_sigma == 1.0
    traj = []
    orc.sample_latents(Wd, sched, *(torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_T")), traj=traj)
    for i in (10, 20, 30, 40, 50):
        assert np.abs(traj[i - 1].numpy() - g[f"x_after_{i}"]).max() < 1e-4, i


def test_scheduler_self_checks():
    d = orc.DDIM()
    # eps == 0: x0 = clamp(x / sqrt(abar_t)), x' = sqrt(abar_prev) * x0
    x = torch.tensor([[0.3, -2.0, 5.0]])
    t = 981
    out = d.step(torch.zeros_like(x), t, x)
    a_t, a_p = d.alphas_cumprod[t], d.alphas_cumprod[t - 20]
    exp = a_p.sqrt() * (x / a_t.sqrt()).clamp(-1, 1)
    assert torch.allclose(out, exp, atol=1e-7)
    # last DDIM step uses final_alpha_cumprod = abar[0] (set_alpha_to_one = false)
    out = d.step(torch.zeros_like(x), 1, x)
    exp = d.alphas_cumprod[0].sqrt() * (x / d.alphas_cumprod[1].sqrt()).clamp(-1, 1)
    assert torch.allclose(out, exp, atol=1e-7)
    p = orc.DDPM()
    assert p.timesteps[0] == 999 and p.timesteps[-1] == 0 and len(p.timesteps) == 1000
    assert not p.needs_noise(0) and p.needs_noise(1)
    # DDPM posterior mean with eps = 0 and x0-consistency: x = sqrt(abar_t) x0 -> mean = c0 x0 + cx x
    x0 = torch.tensor([[0.5, -0.25]])
    t = 500
    xt = p.alphas_cumprod[t].sqrt() * x0
    mean = p.step(torch.zeros_like(xt), t, xt, torch.zeros_like(xt))
    a_t, a_p = p.alphas_cumprod[t].double(), p.alphas_cumprod[t - 1].double()
    al = a_t / a_p
    ref = (a_p.sqrt() * (1 - al) / (1 - a_t)) * x0.double() + (al.sqrt() * (1 - a_p) / (1 - a_t)) * xt.double()
    assert torch.allclose(mean.double(), ref, atol=1e-6)


def test_vae_decode(Wp):
    g = np.load(GOLDEN / "vae_decode.npz")
    z = torch.from_numpy(g["z"])
    taps = {}
    feats = orc.vae_decode(Wp, z, taps=taps)
    assert feats.shape == (3, 300, 333)
    assert np.abs(feats.numpy() - g["feats"]).max() < 2e-5
    for k in ("decoder.input_blocks.0", "decoder.output_blocks.3"):
        assert np.abs(taps[k].numpy()[:, ::25] - g[f"tap/{k}"]).max() < 2e-5
    fr = orc.vae_decode(Wp, z[:2], lengths=[300, 173])
    assert np.abs(fr.numpy() - g["feats_ragged"]).max() < 2e-5
    assert np.all(fr.numpy()[1, 173:] == 0)


def test_vae_encode(Wp):
    g = np.load(GOLDEN / "vae_encode.npz")
    feats = torch.from_numpy(g["feats"].astype(np.float32))
    mu, std = orc.vae_encode(Wp, feats)
    assert np.abs(mu.numpy() - g["mu"]).max() < 2e-5 and np.abs(std.numpy() - g["std"]).max() < 2e-5
    mu, std = orc.vae_encode(Wp, feats, lengths=[300, 211])
    assert np.abs(mu.numpy() - g["mu_ragged"]).max() < 2e-5 and np.abs(std.numpy() - g["std_ragged"]).max() < 2e-5
    # axis-angle -> 6D front end of _loader_helper_v1 (infer_ldm.py:459-463) round-trips through the decoder-side conversion
    aa = 0.8 * torch.randn(50, 3, generator=torch.Generator().manual_seed(0))
    d6 = orc.axis_angle_to_rotation_6d(aa.double())
    assert (orc.rotation_6d_to_matrix(d6) - orc.axis_angle_to_matrix(aa.double())).abs().max() < 1e-9


def test_rotation_conversions():
    g = np.load(GOLDEN / "rotation.npz")
    d6 = torch.from_numpy(g["d6"])
    mat = orc.rotation_6d_to_matrix(d6)
    assert np.abs(mat.numpy() - g["mat"]).max() < 1e-6
    allm = torch.cat([torch.from_numpy(g["mat"]), torch.from_numpy(g["mat_extra"])])
    ql = orc.matrix_to_quaternion(allm, "legacy")
    assert np.abs(ql.numpy() - g["quat_legacy"]).max() < 1e-6
    aal = orc.matrix_to_axis_angle(allm, "legacy")
    assert np.abs(aal.numpy() - g["aa_legacy"]).max() < 2e-5
    # candidate-selection variant: pinned as a rotation (and equal to legacy up to quaternion sign)
    # rows 50:60 are the nearly-parallel (a1, a2) stress inputs: fp32 Gram-Schmidt leaves them
    # non-orthogonal (b1.b2 ~ 3e-3), so they are checked value-wise above but not "as a rotation".
    ok = torch.ones(len(allm), dtype=torch.bool)
    ok[50:60] = False
    qp = orc.matrix_to_quaternion(allm.double(), "p3d")
    assert torch.allclose((qp * qp).sum(-1)[ok], torch.ones(int(ok.sum()), dtype=torch.float64), atol=1e-5)
    same = torch.minimum((qp - ql.double()).abs().max(-1).values, (qp + ql.double()).abs().max(-1).values)
    assert same[ok].max() < 5e-4   # legacy loses precision near pi (sqrt of small positive part)
    aap = orc.matrix_to_axis_angle(allm.double(), "p3d")
    back = orc.axis_angle_to_matrix(aap)
    assert (back - allm.double())[ok].abs().max() < 1e-5
    assert np.abs(orc.axis_angle_to_matrix(aal).numpy() - g["aa2mat"]).max() < 1e-5
    # the deployed variant can return |aa| > pi (committed sample outputs do: npz_layout.json)
    assert torch.linalg.vector_norm(aap, dim=-1).max() > np.pi


def test_npz_layout_contract():
    lay = json.load(open(GOLDEN / "npz_layout.json"))
    assert len(lay) == 3
    f = orc.npz_fields(np.random.default_rng(0).standard_normal((300, 168)).astype(np.float32))
    for name, facts in lay.items():
        assert facts["lower_body_constant"] and facts["trans_zero"] and facts["mocap_frame_rate"] == 30.0
        for k, (dt, shape) in facts["fields"].items():
            if k in ("gender",):
                continue
            assert str(f[k].dtype) == dt and list(f[k].shape) == shape, (name, k)
    assert max(v["max_aa_norm"] for v in lay.values()) > np.pi
    assert np.all(f["poses"][:, orc.LOWER_BODY] == f["poses"][0, orc.LOWER_BODY])


def test_counter_noise_statistics():
    z = orc.counter_normal(2024, np.arange(64), 3, 1)
    assert z.shape == (64, 128) and z.dtype == np.float32
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    z2 = orc.counter_normal(2024, np.arange(32, 64), 3, 1)
    assert np.array_equal(z[32:], z2)  # keyed by global clip index: shard-invariant
    assert not np.array_equal(z, orc.counter_normal(2024, np.arange(64), 4, 1))
    # Philox4x32-10 known-answer (Random123 kat_vectors: ctr = key = 0)
    r = orc.philox4x32_10(np.zeros((1, 4), dtype=np.uint64), (0, 0))[0]
    assert [int(v) for v in r] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]


def test_denoiser_per_sample_timesteps(oracle_env=None):
    """Denoiser with one timestep per sample, as LatentDiffusionModel.diffusion_forward calls it (ldm.py:75-97)."""
    from amuse_amd import weights as wts
    from oracle import amuse_oracle as orc
    g = np.load(GOLDEN / "denoiser_steps.npz")
    W = orc.to_torch(wts.make_denoiser_weights(0))
    t = lambda k: torch.from_numpy(g[k])
    eps = orc.denoiser_forward(W, t("x_t"), [int(v) for v in g["timesteps_batch"]], t("con"), t("emo"), t("sty"))
    assert float((eps - t("eps_batch_t")).abs().max()) < 1e-5
    # add_noise + forward: with noise = 0 and t such that sqrt(abar) ~ 1 the noisy latent is the clean one
    out = orc.diffusion_forward(W, t("x_t"), torch.zeros(3, 128), [0, 0, 0], t("con"), t("emo"), t("sty"))
    assert float((out["noisy_latents"] - float(orc.SchedulerBase().alphas_cumprod[0].sqrt()) * t("x_t")).abs().max()) < 1e-6


def test_gelu_poly_model_is_close_to_the_reference_activation():
    """The bf16 sampling kernel's polynomial GELU (oracle gelu_poly = csrc gelu_poly4) against the reference's exact-erf
    GELU: absolute error bound everywhere, sub-ulp of the bf16 rounding that follows it where activations are large."""
    import torch
    from oracle import amuse_oracle as orc
    x = torch.linspace(-10.0, 10.0, 400001)
    ref = orc.gelu(x.double())
    d = (orc.gelu_poly(x).double() - ref).abs()
    assert float(d.max()) < 2e-4
    m = (x >= -2.0) & (x <= 4.0)
    ulp = 2.0 ** (torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
    assert float((d / ulp)[m].max()) < 0.35
    assert float(orc.gelu_poly(torch.tensor([0.0]))) == 0.0


def test_split_precision_emulation(Wd):
    """The arithmetic of the fp32x mode (include/amuse_hip.h AMUSE_PREC_F32X), emulated on the CPU against the reference modules'
    golden eps_hat - the measurement that chose the scheme before a kernel existed: every GEMM operand split
    into two fp16 pieces, products Wh.xh + Wh.xl + Wl.xh, holds the fp32 bar (1e-5); the bf16 split of the same byte count
    (weights hi + lo, activations hi + mid + lo, five products) does not."""
    g = np.load(GOLDEN / "denoiser_steps.npz")
    con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_t"))

    def split(t, dt, n):
        parts, r = [], t.clone()
        for _ in range(n):
            p = r.to(dt).to(torch.float32)
            parts.append(p)
            r = r - p
        return parts

    class SplitOps(orc.Ops):
        def __init__(self, dt, nw, nx, terms):
            super().__init__(False, False)
            self.dt, self.nw, self.nx, self.terms = dt, nw, nx, terms

        def _prod(self, a, b, nb):
            ap, bp = split(a, self.dt, self.nx), split(b, self.dt, nb)
            return sum((ap[i].double() @ bp[j].double()).float() for i, j in self.terms if i < self.nx and j < nb)

        def lin(self, x_, w, b=None):
            y = self._prod(x_, w.transpose(-1, -2), self.nw)
            return y if b is None else y + b

        def mm(self, a, b):
            return self._prod(a, b, self.nx)

    def eps_err(ops, t=501):
        xs = orc.denoiser_tokens(Wd, x, t, con, emo, sty)
        out = orc.skip_stack(ops, xs, Wd, "encoder", lambda h, p: orc.enc_block(ops, h, Wd, p))[:, 0]
        return float(np.abs(out.numpy() - g[f"eps_t{t}"]).max())

    f16x2 = eps_err(SplitOps(torch.float16, 2, 2, [(0, 0), (1, 0), (0, 1)]))
    bf16x = eps_err(SplitOps(torch.bfloat16, 2, 3, [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1)]))
    assert f16x2 < 1e-5, f16x2          # measured 3.0e-6 (the fp32 oracle itself: 1.8e-6)
    assert bf16x > 1e-5, bf16x          # measured 2.9e-5


def test_wellcond_second_seed_whole_path():
    """A SECOND weight draw (seed 1) with a well-conditioned decoder (weights.make_wellcond_prior_weights), B = 4 + an emotion-dropped
    job, fixtures by the reference's own Denoiser / MotionPrior.decode / rotation_6d_to_matrix (oracle/gen_golden.py --wellcond):
    the oracle's whole path - DDIM-50 -> decode -> 6D -> axis-angle - meets the north star's bar on EVERY joint
    (infer_ldm.py:130-178), plus the per-kernel goldens of that draw."""
    g = np.load(GOLDEN / "wellcond.npz")
    Wd1, Wp1 = orc.to_torch(wts.make_denoiser_weights(1)), orc.to_torch(wts.make_wellcond_prior_weights(1))
    con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_T"))
    for t in (981, 501, 1):
        assert np.abs(orc.denoiser_forward(Wd1, x, t, con, emo, sty).numpy() - g[f"eps_t{t}"]).max() < 1e-5
    for tag, n, e in (("full", 4, emo), ("noemo", 2, None)):
        assert float(g[f"{tag}/min_pivot"]) >= 0.5 and float(g[f"{tag}/tie_margin"]) > 1e-2
        out = orc.diffusion_backward(Wd1, Wp1, orc.DDIM(), con[:n], None if e is None else e[:n], sty[:n], x[:n])
        assert np.abs(out["latents"].numpy() - g[f"{tag}/latents"]).max() < 1e-4
        assert np.abs(out["feats"].numpy() - g[f"{tag}/feats"]).max() < 1e-4
        d = np.linalg.norm(out["poses"].numpy() - g[f"{tag}/poses"], axis=-1)
        assert d.shape == (n, 300, 55) and d.max() < 1e-4, d.max()
        assert (np.linalg.norm(g[f"{tag}/poses"], axis=-1) > np.pi).sum() > 100      # the |aa| > pi regime is in the fixture
    lat = torch.from_numpy(g["full/latents"])
    fr = orc.vae_decode(Wp1, lat[:2], [300, 173])
    assert np.abs(fr.numpy() - g["feats_ragged"]).max() < 2e-5
    fe = torch.from_numpy(g["enc_feats"].astype(np.float32))
    for lens, sfx in ((None, ""), ([300, 211], "_ragged")):
        mu, std = orc.vae_encode(Wp1, fe, lens)
        assert np.abs(mu.numpy() - g["mu" + sfx]).max() < 2e-5
        assert np.abs(std.numpy() / g["std" + sfx] - 1).max() < 2e-5
