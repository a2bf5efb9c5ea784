"""GPU: the north star's parity sentence taken literally - "outputs match the reference PyTorch path ... per-joint L2 vs reference
< 1e-4" - on EVERY joint, end to end through the C ABI (x_T -> DDIM-50 -> MotionPrior.decode -> 6D -> axis-angle), against
fixtures written by the reference's own modules on a SECOND weight draw (seed 1) whose decoder emits well-conditioned 6D rotations
as a trained one does (tests/golden/wellcond.npz, oracle/gen_golden.py --wellcond; weights.make_wellcond_prior_weights).
Both parity modes (fp32, fp32x) on every decode kernel family; the 16-bit throughput modes are REPORTED as rotation distances and
gated loosely.  The per-kernel goldens of the second draw (eps_hat, decode ragged, encode) at the bars of tests/test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
PARITY = ("fp32", "fp32x")


def _err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def env():
    from amuse_amd import scheduler as sch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(1), wts.make_wellcond_prior_weights(1), "cuda:0")
    eng.set_schedule(sch.ddim_table())
    g = np.load(GOLDEN / "wellcond.npz")
    yield {"eng": eng, "g": g, **{k: torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_T")}}
    eng.close()


def _jobs(env):
    return (("full", 4, env["emo"]), ("noemo", 2, None))


@pytest.mark.parametrize("prec", PARITY)
def test_second_seed_eps_hat_vs_reference_module(env, prec):
    eng, g = env["eng"], env["g"]
    for t in (981, 501, 1):
        eps = eng.denoise_step(env["x_T"], t, env["con"], env["emo"], env["sty"], prec)
        assert _err(eps, g[f"eps_t{t}"]) < 1e-5, t


@pytest.mark.parametrize("path", ("auto", "staged", "fused", "clip"))
@pytest.mark.parametrize("prec", PARITY)
def test_every_joint_end_to_end_vs_reference_modules(env, prec, path):
    """amuse_diffusion_backward (ONE call: sampler + decode + rotation epilogue) against reference Denoiser x DDIM-50 -> reference
    MotionPrior.decode -> rotation_6d_to_matrix -> matrix_to_axis_angle: latents, translation and EVERY joint's axis-angle within
    1e-4; then the same latents decoded with features returned: features within 1e-4."""
    if prec == "fp32" and path != "auto":
        pytest.skip("the fp32 mode has one decode kernel family")
    eng, g = env["eng"], env["g"]
    eng.set_decode_path(path)
    try:
        for tag, n, emo in _jobs(env):
            out = eng.diffusion_backward(env["con"][:n], None if emo is None else emo[:n], env["sty"][:n], prec, x_init=env["x_T"][:n])
            assert _err(out["latents"], g[f"{tag}/latents"]) < 1e-4
            d = np.linalg.norm(out["poses"].cpu().numpy() - g[f"{tag}/poses"], axis=-1)
            assert d.shape == (n, 300, 55)
            assert d.max() < 1e-4, (tag, float(d.max()), int((d >= 1e-4).sum()))
            assert _err(out["trans"], g[f"{tag}/feats"][..., -3:]) < 1e-4
            f = eng.vae_decode(out["latents"], None, prec, return_feats=True)
            assert _err(f["feats"], g[f"{tag}/feats"]) < 1e-4
            assert torch.equal(f["poses"], out["poses"])
    finally:
        eng.set_decode_path("auto")


@pytest.mark.parametrize("prec", PARITY)
def test_every_joint_inside_a_chip_filling_launch(env, prec):
    """The same four clips as rows 0-3 and 252-255 of a 256-clip launch (the kernels AUTO takes at BASELINE config 3's size: two clips
    per sampler tile, the per-clip decoders): every joint within 1e-4 of the reference-module fixture there too."""
    eng, g = env["eng"], env["g"]
    rep = lambda t: torch.cat([t, t.flip(0).repeat(62, 1), t])            # 4 + 248 + 4
    out = eng.diffusion_backward(rep(env["con"]), rep(env["emo"]), rep(env["sty"]), prec, x_init=rep(env["x_T"]))
    for sl in (slice(0, 4), slice(252, 256)):
        assert _err(out["latents"][sl], g["full/latents"]) < 1e-4
        d = np.linalg.norm(out["poses"][sl].cpu().numpy() - g["full/poses"], axis=-1)
        assert d.max() < 1e-4, float(d.max())


@pytest.mark.parametrize("prec", PARITY)
def test_second_seed_decode_ragged_and_encode_vs_reference_module(env, prec):
    eng, g = env["eng"], env["g"]
    o = eng.vae_decode(g["full/latents"][:2], [300, 173], prec, return_feats=True)
    assert _err(o["feats"], g["feats_ragged"]) < 2e-5
    assert float(o["feats"][1, 173:].abs().max()) == 0.0 and float(o["poses"][1, 173:].abs().max()) == 0.0
    fe = torch.from_numpy(g["enc_feats"].astype(np.float32))
    for lens, sfx in ((None, ""), ([300, 211], "_ragged")):
        e = eng.vae_encode(fe, lens, prec)
        assert _err(e["mu"], g["mu" + sfx]) < 2e-5
        assert _err(e["std"].cpu() / torch.from_numpy(g["std" + sfx]), np.ones((2, 128))) < 5e-5


def _geodesic_deg(orc, a, b):
    Ra, Rb = orc.axis_angle_to_matrix(torch.as_tensor(a).double()), orc.axis_angle_to_matrix(torch.as_tensor(b).double())
    tr = (Ra * Rb).sum(dim=(-1, -2))
    return torch.rad2deg(torch.acos(((tr - 1) / 2).clamp(-1, 1)))


@pytest.mark.parametrize("prec", ("bf16", "fp16"))
def test_throughput_modes_report_rotation_distance(env, prec, record_property):
    """The 16-bit operand modes against the same reference-module fixture: not a < 1e-4 claim (operand rounding is 2^-9 / 2^-12) -
    the geodesic distance between the rotations, median and p99, reported and gated at a loose multiple of what was measured."""
    from oracle import amuse_oracle as orc
    eng, g = env["eng"], env["g"]
    out = eng.diffusion_backward(env["con"], env["emo"], env["sty"], prec, x_init=env["x_T"])
    deg = _geodesic_deg(orc, out["poses"].cpu(), g["full/poses"])
    med, p99 = float(deg.median()), float(deg.flatten().kthvalue(int(0.99 * deg.numel())).values)
    lat = _err(out["latents"], g["full/latents"])
    record_property("geodesic_median_deg", med)
    record_property("geodesic_p99_deg", p99)
    print(f"{prec}: DDIM-50 end to end vs reference modules: latents max {lat:.3e}, geodesic median {med:.3f} deg, p99 {p99:.3f} deg")
    bar = {"bf16": (0.6, 1.5, 0.3), "fp16": (0.09, 0.2, 0.045)}[prec]     # measured (MI355X): 0.295 / 0.70 deg, 0.149; 0.043 / 0.094 deg, 0.022
    assert med < bar[0] and p99 < bar[1] and lat < bar[2]
