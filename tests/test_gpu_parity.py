"""GPU parity tests proper: the HIP path (through the C ABI, via amuse_amd.engine) against the oracle
and against the golden vectors produced by the reference's own modules.

Tolerances
  fp32 mode  - teacher-forced eps_hat <= 1e-5, DDIM-50 latents <= 1e-4, decoder features <= 2e-5,
               per-joint L2 on SMPL-X axis-angle < 1e-4 (BASELINE.json north_star)
  bf16 mode  - bf16 operand rounding is discontinuous, so a 1e-6 upstream difference can flip a
               rounding and move a whole-network output by O(1e-2).  The tight check is therefore
               per block, teacher-forced on the kernel's OWN block inputs (<= 1e-4 for most blocks);
               whole-network eps_hat is held to 5e-2 abs (|eps| ~ 3).
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# the two parity modes are held to the SAME bars: "fp32" (v_mfma_f32_16x16x4_f32) and "fp32x" (split-fp16 operands, three
# v_mfma_f32_16x16x32_f16 per product - include/amuse_hip.h AMUSE_PREC_F32X)
PARITY = ("fp32", "fp32x")


def _conditioning(orc, feats):
    """Per joint: |a1|, |a2 - (b1.a2) b1| (Gram-Schmidt pivots of the 6D rotation) and the margin between
    the two largest |q| candidates of matrix_to_quaternion.  Random weights (no checkpoint ships) emit many
    ill-conditioned 6D vectors (pivots down to 1e-3), where fp32 Gram-Schmidt amplifies an input difference
    of eps to eps / pivot; trained decoders emit near-orthonormal 6D."""
    d6 = feats[..., :-3].reshape(*feats.shape[:-1], 55, 6).double()
    a1, a2 = d6[..., :3], d6[..., 3:]
    n1 = a1.norm(dim=-1)
    b1 = a1 / n1[..., None]
    n2 = (a2 - (b1 * a2).sum(-1, keepdim=True) * b1).norm(dim=-1)
    m = orc.rotation_6d_to_matrix(d6)
    t = [m[..., i, i] for i in range(3)]
    qa = torch.stack([1 + t[0] + t[1] + t[2], 1 + t[0] - t[1] - t[2], 1 - t[0] + t[1] - t[2],
                      1 - t[0] - t[1] + t[2]], -1).clamp(min=0).sqrt()
    top = qa.topk(2, dim=-1).values
    return torch.minimum(n1, n2), top[..., 0] - top[..., 1]


_ORACLE = {}   # CPU-oracle results that do not depend on the kernel's precision mode: computed once per session, shared by the parametrisations


def _once(key, fn):
    if key not in _ORACLE:
        _ORACLE[key] = fn()
    return _ORACLE[key]


def _ddpm1000_jobs(orc, Wd):
    """tests/tools/ddpm1000_job.py: the two 1000-step oracle runs of this file as ONE (shared by both tests and both parity modes) - taken from the child process
    tests/conftest.py started at the session's start when there is one (the CPU work then ran beside the GPU tests in front of these), else computed here."""
    def make():
        import time
        sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
        import ddpm1000_job as job
        j = job.inputs(orc)
        pre = getattr(pytest, "_amuse_ddpm1000", None)
        ref = None
        if pre is not None:
            proc, out = pre
            t0 = time.time()
            while proc.poll() is None and time.time() - t0 < 600:
                time.sleep(0.2)
            if proc.poll() == 0 and out.exists():
                ref = torch.from_numpy(np.load(out))
        if ref is None:
            ref = job.oracle_latents(orc, Wd, j)
        j["ref"] = ref
        return j
    return _once("ddpm1000", make)


def _err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def env():
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    from oracle import amuse_oracle as orc
    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    eng = HipEngine(wd, wp, "cuda:0")
    yield {"eng": eng, "Wd": orc.to_torch(wd), "Wp": orc.to_torch(wp), "orc": orc}
    eng.close()


def test_native_library_is_loaded(env):
    from amuse_amd import _lib
    maps = open("/proc/self/maps").read()
    assert str(_lib.LIB_PATH) in maps


def test_counter_normal_matches_restatement(env):
    orc, eng = env["orc"], env["eng"]
    for step, stream in ((0, 0), (7, 1), (999, 1)):
        z = eng.counter_normal(2024, 100, 64, step, stream)
        ref = orc.counter_normal(2024, np.arange(100, 164), step, stream)
        assert _err(z, ref) < 5e-5  # hardware log2/sin/cos vs libm
    z = eng.counter_normal(2024, 0, 4096, 1, 1).cpu().numpy()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


@pytest.mark.parametrize("prec", PARITY)
def test_denoise_step_fp32_vs_reference_golden(env, prec):
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    con, emo, sty, x = (g[k] for k in ("con", "emo", "sty", "x_t"))
    for t in (981, 501, 1):
        eps, tap = eng.denoise_step(x, t, con, emo, sty, prec, taps=True)
        assert _err(eps, g[f"eps_t{t}"]) < 1e-5
        if t == 981:
            tp = tap.cpu().numpy()
            assert _err(tp[0, :5], g["tap981/tokens"][0]) < 1e-5
            assert _err(tp[1, :5], g["tap981/encoder.input_blocks.0"][0]) < 1e-5
            assert _err(tp[5, :5], g["tap981/encoder.middle_block"][0]) < 1e-5
            assert _err(tp[9, :5], g["tap981/encoder.output_blocks.3"][0]) < 1e-5
    # token dropping: emo / sty None (denoiser.py:159-171) -> S = 4 / 3
    assert _err(eng.denoise_step(x, 501, con, None, sty, prec), g["eps_t501_noemo"]) < 1e-5
    assert _err(eng.denoise_step(x, 501, con, None, None, prec), g["eps_t501_consolo"]) < 1e-5


@pytest.mark.parametrize("prec", PARITY)
def test_denoise_step_clip_tiling_is_invisible(env, prec):
    """1, 2 or 3 clips per workgroup tile, ragged last tile: same numbers."""
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    gen = torch.Generator().manual_seed(1)
    c, e, s, x = (torch.randn(7, n, generator=gen) for n in (256, 256, 256, 128))
    ref = orc.denoiser_forward(Wd, x, 321, c, e, s)
    outs = []
    try:
        for G in (1, 2, 3):
            eng.set_clips_per_group(G)
            outs.append(eng.denoise_step(x, 321, c, e, s, prec).cpu())
            assert _err(outs[-1], ref) < 1e-5
    finally:
        eng.set_clips_per_group(0)
    # (not bitwise: the softmax partial sums associate differently when a clip sits at another tile row)
    assert _err(outs[0], outs[1]) < 1e-5 and _err(outs[0], outs[2]) < 1e-5


def test_denoise_step_bf16_blockwise(env):
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_t"))
    eps, tap = eng.denoise_step(x, 981, con, emo, sty, "bf16", taps=True)
    ref = orc.denoiser_forward(Wd, x, 981, con, emo, sty, emulate_bf16=True)
    assert _err(eps, ref) < 5e-2 and _err(eps, g["eps_t981"]) < 8e-2
    tp = tap.cpu()
    ops = orc.Ops(True, poly_gelu=True)   # the bf16 sampling kernel's arithmetic (oracle gelu_poly)
    names = [f"encoder.input_blocks.{i}" for i in range(4)] + ["encoder.middle_block"] + \
            [f"encoder.output_blocks.{i}" for i in range(4)]
    errs = []
    for b, n in enumerate(names):  # block b maps slot b -> slot b + 1, teacher-forced on the kernel's inputs
        xin = tp[b, :5][None]
        if b >= 5:
            sk = tp[1 + (8 - b), :5][None]  # xs.pop(): output of input block 8 - b
            xin = ops.lin(torch.cat([xin, sk], -1), Wd[f"encoder.linear_blocks.{b - 5}.weight"],
                          Wd[f"encoder.linear_blocks.{b - 5}.bias"])
        errs.append(_err(tp[b + 1, :5][None], orc.enc_block(ops, xin, Wd, n)))
    assert max(errs) < 2e-2, errs
    assert sorted(errs)[6] < 1e-4, errs   # at least 7 of 9 blocks free of rounding flips
    # token dropping (S = 4 / 3 rows per clip, up to 5 clips per workgroup tile) through the 8-wave kernel, against the
    # reference goldens with the whole-network bf16 tolerance, for every tiling
    try:
        for G in (0, 1, 2, 3):
            eng.set_clips_per_group(G)
            assert _err(eng.denoise_step(x, 501, con, None, sty, "bf16"), g["eps_t501_noemo"]) < 8e-2
            assert _err(eng.denoise_step(x, 501, con, None, None, "bf16"), g["eps_t501_consolo"]) < 8e-2
            assert _err(eng.denoise_step(x, 501, con, emo, sty, "bf16"), g["eps_t501"]) < 8e-2
    finally:
        eng.set_clips_per_group(0)


@pytest.mark.parametrize("prec", PARITY)
def test_diffusion_forward_per_clip_timesteps(env, prec):
    """amuse_diffusion_forward (ldm.py:71-97): per-clip timesteps vs the reference Denoiser golden and the oracle."""
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    ts = [int(v) for v in g["timesteps_batch"]]
    # noise = 0 and z0 = x_t / sqrt(abar_t) reproduce the golden's noisy latent
    ac = orc.SchedulerBase().alphas_cumprod
    z0 = torch.from_numpy(g["x_t"]) / ac[torch.tensor(ts)].sqrt()[:, None]
    out = eng.diffusion_forward(z0, torch.zeros(3, 128), ts, g["con"], g["emo"], g["sty"], prec)
    assert _err(out["noisy_latents"], g["x_t"]) < 1e-6
    assert _err(out["noise_pred"], g["eps_batch_t"]) < 1e-5
    # general case against the oracle: 70 clips (clip tiles of several kinds), token dropping
    gen = torch.Generator().manual_seed(17)
    B = 70
    z, n, con, emo = (torch.randn(B, d, generator=gen) for d in (128, 128, 256, 256))
    ts = torch.randint(0, 1000, (B,), generator=gen).tolist()
    ref = orc.diffusion_forward(Wd, z, n, ts, con, emo, None)
    out = eng.diffusion_forward(z, n, ts, con, emo, None, prec)
    assert _err(out["noisy_latents"], ref["noisy_latents"]) < 1e-6
    assert _err(out["noise_pred"], ref["noise_pred"]) < 1e-5
    with pytest.raises(ValueError):
        eng.diffusion_forward(z, n, [1000] * B, con, emo, None)


@pytest.mark.parametrize("prec", PARITY)
def test_ddim50_fp32_matches_reference_trajectory(env, prec):
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    tr = np.load(GOLDEN / "ddim50_traj.npz")
    eng.set_schedule(sch.ddim_table())
    assert list(eng.schedule.timesteps) == list(range(981, 0, -20))
    lat, traj = eng.sample(tr["con"], tr["emo"], tr["sty"], prec, x_init=tr["x_T"], return_traj=True)
    for i in (10, 20, 30, 40, 50):
        assert _err(traj[i - 1], tr[f"x_after_{i}"]) < 1e-4, i
    assert torch.equal(lat, traj[-1])
    # bf16 throughput mode: measured drift, bounded (O(0.3) on rms 0.64)
    latb = eng.sample(tr["con"], tr["emo"], tr["sty"], "bf16", x_init=tr["x_T"])
    assert _err(latb, tr["x_after_50"]) < 0.3   # 64-clip statistics are gated in test_gpu_configs.py (rms 0.08 / max 0.46)


@pytest.mark.parametrize("prec", PARITY)
def test_ddpm_explicit_noise_vs_oracle(env, prec):
    """Ancestral DDPM (strided to 100 steps so the CPU oracle stays fast) with explicit x_T and noise."""
    from amuse_amd import scheduler as sch
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    T, B = 100, 2
    gen = torch.Generator().manual_seed(5)
    c, e, s, x = (torch.randn(B, n, generator=gen) for n in (256, 256, 256, 128))
    nz = torch.randn(T, B, 128, generator=gen)
    eng.set_schedule(sch.ddpm_table(T))
    lat, traj = eng.sample(c, e, s, prec, x_init=x, step_noise=nz, return_traj=True)
    osched = orc.DDPM(T)
    assert list(eng.schedule.timesteps) == osched.timesteps
    otraj = []
    orc.sample_latents(Wd, osched, c, e, s, x, nz, traj=otraj)
    scale = float(otraj[-1].abs().max())
    assert _err(traj[0], otraj[0]) < 1e-5
    assert _err(lat, otraj[-1]) < 2e-4 * max(1.0, scale)


@pytest.mark.parametrize("prec", PARITY)
def test_ddpm1000_fp32_single_clip_vs_oracle(env, prec):
    """BASELINE config 2 shape: 1 clip, the full 1000-step ancestral DDPM, explicit x_T and per-step noise."""
    from amuse_amd import scheduler as sch
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    j = _ddpm1000_jobs(orc, Wd)
    c, e, s, x, nz = j["single"]
    eng.set_schedule(sch.ddpm_table())
    lat = eng.sample(c, e, s, prec, x_init=x, step_noise=nz)
    ref = j["ref"][4:]
    scale = float(ref.abs().max())          # random weights drive the latent to rms ~ 30 (no clipping in DDPM)
    assert _err(lat, ref) < 3e-5 * scale    # measured 1.5e-4 abs on rms 31 (4.6e-6 relative)


def test_in_kernel_noise_is_shard_invariant(env):
    """Counter-based noise keyed by the GLOBAL clip index: 8 clips at once == two shards of 4, bitwise."""
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    gen = torch.Generator().manual_seed(9)
    c, e, s = (torch.randn(8, 256, generator=gen) for _ in range(3))
    eng.set_schedule(sch.ddpm_table(50))
    for prec in ("fp32", "bf16", "fp32x", "fp16"):
        full = eng.sample(c, e, s, prec, seed=2024, clip_index0=16)
        a = eng.sample(c[:4], e[:4], s[:4], prec, seed=2024, clip_index0=16)
        b = eng.sample(c[4:], e[4:], s[4:], prec, seed=2024, clip_index0=20)
        assert torch.equal(full, torch.cat([a, b]))
        assert torch.isfinite(full).all()
        other = eng.sample(c, e, s, prec, seed=2025, clip_index0=16)
        assert not torch.equal(full, other)


def test_in_kernel_noise_equals_explicit_counter_noise(env):
    """The noise the sampling kernel draws in-kernel (wave-distributed Philox + bpermute gather) is exactly the
    stream amuse_counter_normal exposes: feeding that stream back as explicit noise reproduces the run bitwise,
    for 1, 2 and 3 clips per workgroup tile."""
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    T, B, seed, c0 = 12, 7, 99, 40
    gen = torch.Generator().manual_seed(4)
    c, e, s = (torch.randn(B, 256, generator=gen) for _ in range(3))
    eng.set_schedule(sch.ddpm_table(T))
    x0 = eng.counter_normal(seed, c0, B, 0, 0)
    nz = torch.stack([eng.counter_normal(seed, c0, B, st, 1) for st in range(T)])
    try:
        for G in (1, 2, 3):
            eng.set_clips_per_group(G)
            for prec in ("fp32", "bf16", "fp32x", "fp16"):   # two different kernels (k_sampler.hip / k_sampler8.hip)
                a = eng.sample(c, e, s, prec, seed=seed, clip_index0=c0)
                b = eng.sample(c, e, s, prec, x_init=x0, step_noise=nz)
                assert torch.equal(a, b), (G, prec)
        # token dropping makes room for more clips per 16-row tile: S = 4 (con + sty) -> up to 4, S = 3 (con only) -> up
        # to 5.  The 4-wave kernel draws the tile's noise in ceil(G / 2) wave-wide Philox calls: the 5th clip of a tile
        # must get its ancestral noise too (it silently kept z = 0 before), and B = 7 leaves a ragged last tile.
        for G, (ee, ss) in ((4, (None, s)), (4, (None, None)), (5, (None, None))):
            eng.set_clips_per_group(G)
            for prec in ("fp32", "bf16", "fp32x", "fp16"):
                a = eng.sample(c, ee, ss, prec, seed=seed, clip_index0=c0)
                b = eng.sample(c, ee, ss, prec, x_init=x0, step_noise=nz)
                assert torch.equal(a, b), (G, prec, ee is None, ss is None)
                eng.set_clips_per_group(1)
                one = eng.sample(c, ee, ss, prec, seed=seed, clip_index0=c0)     # one clip per tile: same noise, same run
                eng.set_clips_per_group(G)
                assert float((a - one).abs().max()) < (0.5 if prec == "bf16" else 1e-3) * max(1.0, float(one.abs().max()))
    finally:
        eng.set_clips_per_group(0)
    with pytest.raises(Exception):
        eng.set_clips_per_group(6)


@pytest.mark.parametrize("prec", ["fp32", "fp32x", "bf16"])
def test_staged_decode_key_mask_edge_lengths(env, prec):
    """The staged attention kernels treat pairs of key tiles (32 keys) below a clip's length and the pair the length falls into differently:
    lengths on both sides of those boundaries - 1, 17 (inside the first pair), 32 / 64 / 288 (exactly on a boundary), 33, 173, 300 - against the CPU
    oracle on the same latents, on the staged kernels and (fp32x) on the unsplit eight-wave attention behind the row kernel without split-K."""
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    lengths = [300, 173, 1, 17, 32, 33, 64, 288]
    z = torch.randn(len(lengths), 128, generator=torch.Generator().manual_seed(11))
    ref = orc.vae_decode(Wp, z, lengths, emulate_bf16=(prec == "bf16"))
    tol = 5e-2 if prec == "bf16" else 2e-5
    for path in ("staged", "fused") if prec == "fp32x" else ("staged",):
        try:
            eng.set_decode_path(path)
            o = eng.vae_decode(z, lengths, prec, return_feats=True)
        finally:
            eng.set_decode_path("auto")
        assert _err(o["feats"], ref) < tol, (path, _err(o["feats"], ref))
        for b, n in enumerate(lengths):
            assert float(o["feats"][b, n:].abs().max() if n < 300 else 0.0) == 0.0
            assert float(o["feats"][b, :n].abs().max()) > 0.0
        # a clip alone gives the bits it has inside the batch (its mask does not leak into its neighbours' tiles)
        one = eng.vae_decode(z[3:4], [17], prec, return_feats=True) if path == "staged" else None
        if one is not None:
            assert torch.equal(one["feats"][0], o["feats"][3])


@pytest.mark.parametrize("prec", PARITY)
def test_vae_decode_fp32_vs_reference_golden(env, prec):
    orc, eng = env["orc"], env["eng"]
    g = np.load(GOLDEN / "vae_decode.npz")
    out = eng.vae_decode(g["z"], None, prec, return_feats=True)
    assert out["feats"].shape == (3, 300, 333) and out["poses"].shape == (3, 300, 55, 3)
    assert _err(out["feats"], g["feats"]) < 2e-5
    # ragged: padded keys masked, padded frames zeroed (vae.py:217,262,274)
    o2 = eng.vae_decode(g["z"][:2], [300, 173], prec, return_feats=True)
    assert _err(o2["feats"], g["feats_ragged"]) < 2e-5
    assert float(o2["feats"][1, 173:].abs().max()) == 0.0
    assert float(o2["poses"][1, 173:].abs().max()) == 0.0 and float(o2["trans"][1, 173:].abs().max()) == 0.0
    # rotation epilogue against the oracle applied to the SAME features
    for mode in ("p3d", "legacy"):
        o = eng.vae_decode(g["z"], None, prec, quat_mode=mode, return_feats=True)
        feats = o["feats"].cpu()
        poses, trans = orc.feats_to_smplx(feats, mode)             # fp32 oracle on the SAME features
        assert torch.equal(o["trans"].cpu(), feats[..., -3:])
        pivot, margin = _conditioning(orc, feats)
        ok = (pivot > 0.1) & (margin > 1e-3)
        assert float(ok.float().mean()) > 0.95
        d = torch.linalg.vector_norm(o["poses"].cpu() - poses, dim=-1)
        if mode == "p3d":
            assert float(d[margin > 1e-3].max()) < 1e-4            # per-joint L2 on axis-angle, every joint
            assert float(d[ok].max()) < 2e-5
        else:  # the legacy quaternion takes sqrt(max(0, 1 +- m00 +- m11 +- m22)): a component near 0 carries
            # sqrt(fp32 eps) ~ 3e-4 of rounding noise on BOTH sides, by construction of that algorithm
            assert float(d[ok].max()) < 2e-3 and float(d[ok].median()) < 2e-6
        R_gpu = orc.axis_angle_to_matrix(o["poses"].cpu().double())
        R_ref = orc.rotation_6d_to_matrix(feats[..., :-3].reshape(3, 300, 55, 6).double())
        rot_tol = 5e-5 if mode == "p3d" else 2e-3
        assert float((R_gpu - R_ref).abs().amax(dim=(-1, -2))[pivot > 0.1].max()) < rot_tol   # as a rotation, vs fp64
    assert float(torch.linalg.vector_norm(eng.vae_decode(g["z"], None, prec)["poses"], dim=-1).max()) > 0
    if prec == "fp32x":
        # the same bars on the OTHER pair of fp32x decode kernels - the row stages without split-K (k_vae_rows8.hip; "fused" pins them,
        # AUTO takes them from 64 clips) and the unsplit eight-wave attention: same function, another summation order
        try:
            eng.set_decode_path("fused")
            o8 = eng.vae_decode(g["z"], None, prec, return_feats=True)
            assert _err(o8["feats"], g["feats"]) < 2e-5 and not torch.equal(o8["feats"], out["feats"])
            o8r = eng.vae_decode(g["z"][:2], [300, 173], prec, return_feats=True)
            assert _err(o8r["feats"], g["feats_ragged"]) < 2e-5
            assert float(o8r["feats"][1, 173:].abs().max()) == 0.0 and float(o8r["poses"][1, 173:].abs().max()) == 0.0
            big = eng.vae_decode(torch.from_numpy(g["z"]).repeat(40, 1), None, prec, return_feats=True)   # 120 clips: 10-wave workgroups
            assert torch.equal(big["feats"][:3], o8["feats"]) and torch.equal(big["feats"][117:], o8["feats"])
            # ... and on the THIRD fp32x decoder - one persistent workgroup per clip (k_vae_fusedx.hip; "clip" pins it, AUTO takes it where the clips fill rounds of
            # the chip): the golden, the ragged golden, the padded frames, and a clip's bits independent of the launch it sits in
            eng.set_decode_path("clip")
            oc = eng.vae_decode(g["z"], None, prec, return_feats=True)
            assert _err(oc["feats"], g["feats"]) < 2e-5 and _err(oc["feats"], o8["feats"]) < 5e-6 and not torch.equal(oc["feats"], o8["feats"])
            ocr = eng.vae_decode(g["z"][:2], [300, 173], prec, return_feats=True)
            assert _err(ocr["feats"], g["feats_ragged"]) < 2e-5
            assert float(ocr["feats"][1, 173:].abs().max()) == 0.0 and float(ocr["poses"][1, 173:].abs().max()) == 0.0 and float(ocr["trans"][1, 173:].abs().max()) == 0.0
            big = eng.vae_decode(torch.from_numpy(g["z"]).repeat(90, 1), None, prec, return_feats=True)   # 270 clips: a second round of workgroups on the chip
            assert torch.equal(big["feats"][:3], oc["feats"]) and torch.equal(big["feats"][267:], oc["feats"]) and torch.equal(big["poses"][132:135], oc["poses"])
            poses, _ = orc.feats_to_smplx(oc["feats"].cpu(), "p3d")                                        # the rotation epilogue on its own features
            pivot, margin = _conditioning(orc, oc["feats"].cpu())
            dd = torch.linalg.vector_norm(oc["poses"].cpu() - poses, dim=-1)
            assert float(dd[margin > 1e-3].max()) < 1e-4 and float(dd[(pivot > 0.1) & (margin > 1e-3)].max()) < 2e-5
            eng.set_decode_path("auto")                                                                    # AUTO at 256 clips = the per-clip kernel (shard.fusedx_rule)
            auto = eng.vae_decode(torch.from_numpy(g["z"]).repeat(86, 1)[:256], None, prec, return_feats=True)
            assert torch.equal(auto["feats"][:3], oc["feats"])
        finally:
            eng.set_decode_path("auto")


def test_vae_decode_bf16_bounded(env):
    eng = env["eng"]
    g = np.load(GOLDEN / "vae_decode.npz")
    out = eng.vae_decode(g["z"], None, "bf16", return_feats=True)
    assert _err(out["feats"], g["feats"]) < 6e-2  # |feats| ~ 3


@pytest.mark.parametrize("prec", PARITY)
def test_vae_encode_fp32_vs_reference_golden(env, prec):
    """MotionPrior.encode (vae.py:154-214) through amuse_vae_encode: mu / std against the reference's own module."""
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    g = np.load(GOLDEN / "vae_encode.npz")
    feats = torch.from_numpy(g["feats"].astype(np.float32))
    out = eng.vae_encode(feats, None, prec)
    assert out["mu"].shape == (2, 128) and out["std"].shape == (2, 128)
    assert _err(out["mu"], g["mu"]) < 2e-5
    assert _err(out["std"], g["std"]) < 2e-5 * float(g["std"].max())
    assert torch.equal(out["latent"], out["mu"])                       # no eps supplied -> latent = mu
    # ragged: padded frames masked as attention keys, the two distribution tokens always visible (vae.py:176-181)
    o2 = eng.vae_encode(feats, [300, 211], prec)
    assert _err(o2["mu"], g["mu_ragged"]) < 2e-5
    assert _err(o2["std"], g["std_ragged"]) < 2e-5 * float(g["std_ragged"].max())
    assert _err(o2["mu"][1], g["mu"][1]) > 1e-3                        # the mask is live
    # masked frames are dead inputs
    f2 = feats.clone()
    f2[1, 211:] = 7.0
    o3 = eng.vae_encode(f2, [300, 211], prec)
    assert torch.equal(o3["mu"], o2["mu"]) and torch.equal(o3["std"], o2["std"])
    # rsample with an explicit draw: latent = mu + std * eps, exactly
    eps = torch.randn(2, 128, generator=torch.Generator().manual_seed(5))
    o4 = eng.vae_encode(feats, None, prec, eps=eps)
    assert torch.equal(o4["latent"].cpu(), out["mu"].cpu() + out["std"].cpu() * eps)
    # more clips than one chunk row group, every clip independent of its neighbours
    fb = feats[:1].repeat(5, 1, 1)
    fb[3] = feats[1]
    o5 = eng.vae_encode(fb, None, prec)
    assert torch.equal(o5["mu"][0], out["mu"][0]) and torch.equal(o5["mu"][3], out["mu"][1])
    assert torch.equal(o5["mu"][4], out["mu"][0])
    # oracle on fresh inputs (not the golden draw)
    f6 = 0.7 * torch.randn(3, 300, 333, generator=torch.Generator().manual_seed(11))
    mu, std = orc.vae_encode(Wp, f6, [300, 17, 1])
    o6 = eng.vae_encode(f6, [300, 17, 1], prec)
    assert _err(o6["mu"], mu) < 2e-5 and _err(o6["std"] / std.to(o6["std"].device), torch.ones_like(std)) < 5e-5


def test_vae_encode_fp32x_on_the_row_kernel_without_split_k(env):
    """fp32x encode with stages 1..9 on k_vae_rows8x<ENC> (what a call of >= 64 clips takes; pinned here like the decode's kernel choice): the
    reference goldens at the fp32 bar, full and ragged, and clip-by-clip equal to itself in a larger batch across workgroup shapes."""
    eng = env["eng"]
    g = np.load(GOLDEN / "vae_encode.npz")
    feats = torch.from_numpy(g["feats"].astype(np.float32))
    try:
        eng.set_decode_path("staged")
        four = eng.vae_encode(feats, [300, 211], "fp32x")
        eng.set_decode_path("fused")
        out = eng.vae_encode(feats, None, "fp32x")
        assert _err(out["mu"], g["mu"]) < 2e-5 and _err(out["std"], g["std"]) < 2e-5 * float(g["std"].max())
        o2 = eng.vae_encode(feats, [300, 211], "fp32x")
        assert _err(o2["mu"], g["mu_ragged"]) < 2e-5 and _err(o2["std"], g["std_ragged"]) < 2e-5 * float(g["std_ragged"].max())
        assert not torch.equal(o2["mu"], four["mu"]) and _err(o2["mu"], four["mu"]) < 2e-5       # really two kernels, the same arithmetic
        fb = feats[:1].repeat(70, 1, 1)                                                             # 1,330 tiles: ten-wave workgroups, several per clip boundary
        fb[33] = feats[1]
        o5 = eng.vae_encode(fb, None, "fp32x")
        assert torch.equal(o5["mu"][0], out["mu"][0]) and torch.equal(o5["mu"][33], out["mu"][1]) and torch.equal(o5["mu"][69], out["mu"][0])
        assert torch.equal(o5["std"][33], out["std"][1])
        # ... and as ONE persistent workgroup per clip (k_den_fusedx<encode> in k_vae_fusedx.hip; "clip" pins it, AUTO takes it where the takes fill rounds of the chip):
        # the same goldens at the same bar, the key mask live, masked frames dead inputs, a take's bits independent of its launch
        eng.set_decode_path("clip")
        oc = eng.vae_encode(feats, None, "fp32x")
        assert _err(oc["mu"], g["mu"]) < 2e-5 and _err(oc["std"], g["std"]) < 2e-5 * float(g["std"].max())
        assert not torch.equal(oc["mu"], out["mu"]) and _err(oc["mu"], out["mu"]) < 2e-5
        oc2 = eng.vae_encode(feats, [300, 211], "fp32x")
        assert _err(oc2["mu"], g["mu_ragged"]) < 2e-5 and _err(oc2["std"], g["std_ragged"]) < 2e-5 * float(g["std_ragged"].max())
        f2 = feats.clone()
        f2[1, 211:] = 7.0
        oc3 = eng.vae_encode(f2, [300, 211], "fp32x")
        assert torch.equal(oc3["mu"], oc2["mu"]) and torch.equal(oc3["std"], oc2["std"])
        fb = feats[:1].repeat(270, 1, 1)                                                            # a second round of workgroups on the chip
        fb[133] = feats[1]
        oc5 = eng.vae_encode(fb, None, "fp32x")
        assert torch.equal(oc5["mu"][0], oc["mu"][0]) and torch.equal(oc5["mu"][133], oc["mu"][1]) and torch.equal(oc5["std"][269], oc["std"][0])
        eng.set_decode_path("auto")                                                                 # AUTO at 256 takes = the per-clip kernel
        assert torch.equal(eng.vae_encode(fb[:256], None, "fp32x")["mu"][133], oc["mu"][1])
    finally:
        eng.set_decode_path("auto")


def test_vae_encode_bf16_bounded(env):
    eng = env["eng"]
    g = np.load(GOLDEN / "vae_encode.npz")
    out = eng.vae_encode(torch.from_numpy(g["feats"].astype(np.float32)), None, "bf16")
    assert _err(out["mu"], g["mu"]) < 6e-2        # LayerNorm output, |mu| ~ 3
    assert _err(out["std"] / torch.from_numpy(g["std"]).to(out["std"].device), torch.ones(2, 128)) < 5e-2


def test_smplx_to_feats_vs_oracle_and_round_trip(env):
    """infer_ldm.py:459-464 (axis-angle -> matrix -> 6D | trans) and its inverse at infer_ldm.py:168-173."""
    orc, eng = env["orc"], env["eng"]
    gen = torch.Generator().manual_seed(3)
    poses = 0.6 * torch.randn(2, 300, 55, 3, generator=gen)
    poses[0, :4] = 0.0                                   # identity rotations: the small-angle branch
    poses[0, 4, :5] *= 1e-7
    trans = torch.randn(2, 300, 3, generator=gen)
    feats = eng.smplx_to_feats(poses, trans).cpu()
    ref = torch.cat([orc.axis_angle_to_rotation_6d(poses).reshape(2, 300, 330), trans], -1)
    assert feats.shape == (2, 300, 333)
    assert torch.equal(feats[..., 330:], trans)
    assert _err(feats, ref) < 1e-6
    assert _err(feats, torch.cat([orc.axis_angle_to_rotation_6d(poses.double()).reshape(2, 300, 330), trans.double()], -1)) < 1e-6
    # the decode-side epilogue inverts it as a rotation (the candidate-selection quaternion may return the
    # |aa| > pi representative of the same rotation), and value-wise with the legacy quaternion (q_w >= 0)
    back, _ = orc.feats_to_smplx(feats, "p3d")
    assert _err(orc.axis_angle_to_matrix(back.double()), orc.axis_angle_to_matrix(poses.double())) < 1e-5
    back_l, _ = orc.feats_to_smplx(feats.double(), "legacy")
    sel = torch.linalg.vector_norm(poses, dim=-1) < 3.0
    assert _err(back_l[sel], poses[sel]) < 1e-4


def test_encode_decode_round_trip_shapes_and_determinism(env):
    """edit_gesture's data path (infer_ldm.py:459-468): motion -> latent -> motion; deterministic and length-stable."""
    eng = env["eng"]
    g = np.load(GOLDEN / "vae_encode.npz")
    feats = torch.from_numpy(g["feats"].astype(np.float32))
    a = eng.vae_encode(feats, None, "fp32")
    b = eng.vae_encode(feats, None, "fp32")
    assert torch.equal(a["mu"], b["mu"]) and torch.equal(a["std"], b["std"])
    rec = eng.vae_decode(a["latent"], None, "fp32", return_feats=True)
    assert rec["feats"].shape == (2, 300, 333) and bool(torch.isfinite(rec["feats"]).all())


@pytest.mark.parametrize("prec", PARITY)
def test_diffusion_backward_end_to_end_fp32(env, prec):
    """BASELINE config 1 shape: 1 clip, DDIM-50, explicit x_T -> SMPL-X poses; per-joint L2 < 1e-4."""
    from amuse_amd import scheduler as sch
    orc, eng, Wd, Wp = env["orc"], env["eng"], env["Wd"], env["Wp"]
    gen = torch.Generator().manual_seed(2024)
    c, e, s, x = (torch.randn(1, n, generator=gen) for n in (256, 256, 256, 128))
    eng.set_schedule(sch.ddim_table())
    out = eng.diffusion_backward(c, e, s, prec, x_init=x)
    ref = orc.diffusion_backward(Wd, Wp, orc.DDIM(), c, e, s, x)
    assert out["poses"].shape == (1, 300, 55, 3) and out["trans"].shape == (1, 300, 3)
    assert _err(out["latents"], ref["latents"]) < 1e-4
    assert _err(out["trans"], ref["trans"]) < 1e-4
    # 50 chaotic fp32 steps leave ~2e-5 on the latents; the 6D -> axis-angle conversion then divides by
    # the Gram-Schmidt pivots (random weights: down to 1e-3).  So: the bulk must meet the 1e-4 per-joint
    # bar, and the tail must be explained by conditioning (teacher-forced decode is held to 1e-4 on every
    # joint in test_vae_decode_fp32_vs_reference_golden).
    d = torch.linalg.vector_norm(out["poses"].cpu() - ref["poses"], dim=-1)
    pivot, margin = _conditioning(orc, ref["feats"])
    assert float(d.median()) < 3e-5
    assert float((d < 1e-4).float().mean()) > 0.97
    assert float((d * pivot.clamp(max=1.0))[margin > 1e-3].max()) < 5e-4
    R1, R2 = orc.axis_angle_to_matrix(out["poses"].cpu().double()), orc.axis_angle_to_matrix(ref["poses"].double())
    assert float(((R1 - R2).abs().amax(dim=(-1, -2)) * pivot.clamp(max=1.0)).max()) < 5e-4
    # emotion edit = swap which vector is passed as z_emo (trainer.py:1055-1066): changes the output
    e2 = torch.randn(1, 256, generator=gen)
    out2 = eng.diffusion_backward(c, e2, s, prec, x_init=x)
    assert _err(out2["poses"], out["poses"]) > 1e-3


def test_full_size_batch_properties(env):
    """BASELINE config 3 size (256 clips, DDPM-1000, bf16): finite, deterministic, and a clip's result depends on the batch
    only through its slot inside its workgroup tile (256 clips run two per tile): a tile-aligned sub-batch with the same
    clips per tile reproduces its rows bitwise."""
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    gen = torch.Generator().manual_seed(3)
    c, e, s = (torch.randn(256, 256, generator=gen) for _ in range(3))
    eng.set_schedule(sch.ddpm_table())
    a = eng.diffusion_backward(c, e, s, "bf16", seed=2024)
    b = eng.diffusion_backward(c, e, s, "bf16", seed=2024)
    assert torch.isfinite(a["poses"]).all() and torch.isfinite(a["trans"]).all()
    assert torch.equal(a["poses"], b["poses"])
    from amuse_amd.shard import job_plan
    g = job_plan(256)["clips_per_group"]
    assert g == 2
    assert job_plan(256)["decode_path"] == "clip" and job_plan(6)["decode_path"] == "staged"   # ("clip" = the fused kernel in this mode; the fp32x mode's per-clip decoder)
    eng.set_clips_per_group(g)
    eng.set_decode_path(job_plan(256)["decode_path"])   # the job's decode kernels, not the 6-clip launch's own choice
    try:
        sub = eng.diffusion_backward(c[198:204], e[198:204], s[198:204], "bf16", seed=2024, clip_index0=198)
    finally:
        eng.set_clips_per_group(0)
        eng.set_decode_path("auto")
    assert torch.equal(sub["latents"], a["latents"][198:204])
    assert torch.equal(sub["poses"], a["poses"][198:204])


@pytest.mark.parametrize("prec", PARITY)
def test_full_size_batch_values_against_the_oracle(env, prec):
    """VALUE parity at the exact launch shape bench.py times (BASELINE config 3: 256 clips, DDPM-1000, two clips per tile on 128 workgroups, the job's
    decode kernels), in both parity modes: four clips of the 256-clip launch - first, second (slot 1 of tile 0), one mid-batch, last - against the CPU oracle
    run on those clips alone with the SAME counter-based noise (keyed by the global clip index).  Latents <= 3e-5 relative after 1000 steps (the single-clip
    bar), decoded features <= 1e-4 + the latent difference carried through.  (The bf16 launch of the same shape: test_full_size_batch_properties.)"""
    from amuse_amd import scheduler as sch
    orc, eng, Wd, Wp = env["orc"], env["eng"], env["Wd"], env["Wp"]
    j = _ddpm1000_jobs(orc, Wd)
    c, e, s = j["full"]
    pick = j["pick"]
    eng.set_schedule(sch.ddpm_table())
    out = eng.diffusion_backward(c, e, s, prec, seed=2024)
    ref = j["ref"][:4]
    scale = float(ref.abs().max())
    lat = out["latents"][pick].cpu()
    assert _err(lat, ref) < 3e-5 * scale, (_err(lat, ref), scale)
    # the decode of the launch, on the launch's own latents (teacher-forced: the 1e-4 pose bar is a statement about the decoder + conversion)
    f_ref = orc.vae_decode(Wp, lat)
    f = eng.vae_decode(out["latents"], None, prec, return_feats=True)["feats"][pick].cpu()
    assert _err(f, f_ref) < 2e-5 * max(1.0, float(f_ref.abs().max()))
    p_ref, _ = orc.feats_to_smplx(f_ref)
    d = torch.linalg.vector_norm(out["poses"][pick].cpu() - p_ref, dim=-1)
    assert float(d.median()) < 3e-5 and float((d < 1e-4).float().mean()) > 0.97, (float(d.median()), float(d.max()))


def test_job_level_tiling_makes_shards_bitwise(env):
    """amuse_amd/shard.py: clips per tile chosen from the job's TOTAL clip count + shards aligned to it => the shards of a
    300-clip job (three clips per tile) reproduce the single-launch result bitwise, fp32 and bf16."""
    from amuse_amd import scheduler as sch
    from amuse_amd.shard import job_plan, shard_range
    eng = env["eng"]
    gen = torch.Generator().manual_seed(17)
    B = 300
    c, e, s = (torch.randn(B, 256, generator=gen) for _ in range(3))
    eng.set_schedule(sch.ddim_table())
    g = job_plan(B)["clips_per_group"]
    assert g == 3
    for prec in ("fp32", "bf16", "fp32x", "fp16"):
        full = eng.diffusion_backward(c, e, s, prec, seed=7)          # auto: ceil(300 / 128) = 3 clips per tile
        eng.set_clips_per_group(g)
        try:
            parts = []
            for rank in range(4):
                lo, hi = shard_range(B, rank, 4, align=g)
                assert lo % g == 0
                parts.append(eng.diffusion_backward(c[lo:hi], e[lo:hi], s[lo:hi], prec, seed=7, clip_index0=lo)["latents"])
        finally:
            eng.set_clips_per_group(0)
        assert torch.equal(torch.cat(parts), full["latents"]), prec


def test_error_conventions(env):
    from amuse_amd import _lib
    eng = env["eng"]
    with pytest.raises(_lib.AmuseHipError):
        eng.vae_decode(torch.zeros(2, 128), [300, 0], "fp32")
    with pytest.raises(ValueError):
        eng.denoise_step(torch.zeros(2, 64), 1, torch.zeros(2, 256), None, None)
