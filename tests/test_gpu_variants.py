"""GPU parity of the Denoiser VARIANTS (include/amuse_hip.h AMUSE_ARCH_DEC / _ENC_POSE / _DEC_POSE) through the C ABI, against the
goldens of the reference's own Denoiser class built with arch = "trans_dec" and / or diffusion_only = true
(tests/golden/denoiser_variants.npz) and against the oracle pinned to them (tests/test_oracle_variants.py).
Reference: models/latent_diffusion/denoiser.py:64-66,116-131,174-204; utils/cross_attention.py:195-234,297-345.

Bars: the fp32 bars of tests/test_gpu_parity.py, unchanged - teacher-forced eps_hat <= 1e-5 (fp32 and fp32x), DDIM-50 state <= 1e-4;
the 16-bit modes are held to whole-network bounds (|eps| ~ 3) as there."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

PARITY = ("fp32", "fp32x")
VARIANTS = (("trans_dec", False), ("trans_enc", True), ("trans_dec", True))
ROWS = slice(0, 300, 6)


def tag_of(arch, pose):
    return f"{arch}{'_pose' if pose else ''}"


def _err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def g():
    return np.load(GOLDEN / "denoiser_variants.npz")


@pytest.fixture(scope="module", params=VARIANTS, ids=[tag_of(*v) for v in VARIANTS])
def env(request):
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    from oracle import amuse_oracle as orc
    arch, pose = request.param
    wd = wts.make_denoiser_weights(0, arch, pose)
    eng = HipEngine(wd, None if pose else wts.make_prior_weights(0), "cuda:0", arch=arch, diffusion_only=pose)
    yield {"eng": eng, "W": orc.to_torch(wd), "orc": orc, "arch": arch, "pose": pose, "tag": tag_of(arch, pose)}
    eng.close()


def inputs(g, pose):
    con, emo, sty = (g[k] for k in ("con", "emo", "sty"))
    return con, emo, sty, (g["x_pose"].astype(np.float32) if pose else g["x_lat"])


def cut(e, pose):
    e = e.detach().cpu().numpy() if isinstance(e, torch.Tensor) else np.asarray(e)
    return e[:, ROWS] if pose else e


@pytest.mark.parametrize("prec", PARITY)
def test_variant_eps_vs_reference_golden(env, g, prec):
    eng, pose, tag = env["eng"], env["pose"], env["tag"]
    con, emo, sty, x = inputs(g, pose)
    for t in (981, 501, 1):
        eps = eng.denoise_step(x, t, con, emo, sty, prec)
        assert _err(cut(eps, pose), g[f"{tag}/eps_t{t}"]) < 1e-5, t
    assert _err(cut(eng.denoise_step(x, 501, con, None, sty, prec), pose), g[f"{tag}/eps_t501_noemo"]) < 1e-5    # 3 memory / 3 prefix tokens
    assert _err(cut(eng.denoise_step(x, 501, con, None, None, prec), pose), g[f"{tag}/eps_t501_consolo"]) < 1e-5  # 2
    if pose:
        lens = [int(v) for v in g["lengths_ragged"]]
        e = eng.denoise_step(x, 501, con, emo, sty, prec, lengths=lens)
        assert _err(cut(e, pose), g[f"{tag}/eps_t501_ragged"]) < 1e-5
        assert torch.all(e[1, lens[1]:] == 0) and torch.any(e[1, lens[1] - 1] != 0)
        with pytest.raises(Exception):   # lengths_to_mask sizes the mask by max(lengths): the reference's indexing needs one full clip
            eng.denoise_step(x, 501, con, emo, sty, prec, lengths=[200, 173])


def test_pose_step_fp32x_on_the_row_kernel_without_split_k(env, g):
    """diffusion_only + trans_enc in fp32x with stages 1..8 on k_vae_rows8x (what a call of >= 64 clips takes; pinned here): the reference goldens at the
    1e-5 bar with 4 / 3 / 2 prefix tokens (S = 304 / 303 / 302) and ragged lengths, another kernel than the four-wave one, a DDIM-50 sampling run, and a
    clip's result independent of the batch around it."""
    if env["arch"] != "trans_enc" or not env["pose"]:
        pytest.skip("the diffusion_only + trans_enc variant")
    eng, tag = env["eng"], env["tag"]
    con, emo, sty, x = inputs(g, True)
    try:
        eng.set_decode_path("staged")
        four = eng.denoise_step(x, 501, con, emo, sty, "fp32x")
        eng.set_decode_path("fused")
        for t in (981, 501, 1):
            assert _err(cut(eng.denoise_step(x, t, con, emo, sty, "fp32x"), True), g[f"{tag}/eps_t{t}"]) < 1e-5, t
        e = eng.denoise_step(x, 501, con, emo, sty, "fp32x")
        assert not torch.equal(e, four) and _err(e, four) < 1e-5
        assert _err(cut(eng.denoise_step(x, 501, con, None, sty, "fp32x"), True), g[f"{tag}/eps_t501_noemo"]) < 1e-5
        assert _err(cut(eng.denoise_step(x, 501, con, None, None, "fp32x"), True), g[f"{tag}/eps_t501_consolo"]) < 1e-5
        lens = [int(v) for v in g["lengths_ragged"]]
        assert _err(cut(eng.denoise_step(x, 501, con, emo, sty, "fp32x", lengths=lens), True), g[f"{tag}/eps_t501_ragged"]) < 1e-5
        n = x.shape[0]
        rep = lambda a: np.concatenate([a] * 24, 0)                      # 48+ clips: ten-wave workgroups across clip boundaries
        big = eng.denoise_step(rep(x), 501, rep(con), rep(emo), rep(sty), "fp32x")
        assert torch.equal(big[:n], e) and torch.equal(big[-n:], e)
        # ... and with the blocks between the first and the last row stage on the per-clip kernel (k_vae_fusedx.hip k_den_fusedx; "clip" pins it, AUTO takes it where the
        # clips fill rounds of the chip): the same goldens at the same bar, 4 / 3 / 2 prefix tokens, ragged lengths, a clip's bits independent of its launch
        eng.set_decode_path("clip")
        for t in (981, 501, 1):
            assert _err(cut(eng.denoise_step(x, t, con, emo, sty, "fp32x"), True), g[f"{tag}/eps_t{t}"]) < 1e-5, t
        ec = eng.denoise_step(x, 501, con, emo, sty, "fp32x")
        assert not torch.equal(ec, e) and _err(ec, e) < 1e-5
        assert _err(cut(eng.denoise_step(x, 501, con, None, sty, "fp32x"), True), g[f"{tag}/eps_t501_noemo"]) < 1e-5
        assert _err(cut(eng.denoise_step(x, 501, con, None, None, "fp32x"), True), g[f"{tag}/eps_t501_consolo"]) < 1e-5
        assert _err(cut(eng.denoise_step(x, 501, con, emo, sty, "fp32x", lengths=lens), True), g[f"{tag}/eps_t501_ragged"]) < 1e-5
        rep2 = lambda a: np.concatenate([a] * 130, 0)                    # 260+ clips: a second round of workgroups
        big = eng.denoise_step(rep2(x), 501, rep2(con), rep2(emo), rep2(sty), "fp32x")
        assert torch.equal(big[:n], ec) and torch.equal(big[-n:], ec)
        eng.set_decode_path("auto")                                      # AUTO at 256 clips = the per-clip kernel
        auto = eng.denoise_step(rep2(x)[:256], 501, rep2(con)[:256], rep2(emo)[:256], rep2(sty)[:256], "fp32x")
        assert torch.equal(auto[:n], ec)
    finally:
        eng.set_decode_path("auto")


@pytest.mark.parametrize("prec", PARITY)
def test_trans_dec_taps_vs_reference_golden(env, g, prec):
    if env["arch"] != "trans_dec" or env["pose"]:
        pytest.skip("taps exist for the latent trans_dec kernel")
    eng, tag = env["eng"], env["tag"]
    con, emo, sty, x = inputs(g, False)
    eps, tap = eng.denoise_step(x, 981, con, emo, sty, prec, taps=True)
    tap = tap.cpu().numpy()
    B = x.shape[0]
    assert _err(tap[0, :B], g[f"{tag}/tap981/tokens"][:, 0]) < 1e-6            # x_t + query_pos.pe[0]
    assert _err(tap[1, :B], g[f"{tag}/tap981/decoder.layers.0"][:, 0]) < 1e-5
    assert _err(tap[9, :B], g[f"{tag}/tap981/decoder.layers.8"][:, 0]) < 1e-5
    assert _err(tap[10, :B], eps) == 0.0


@pytest.mark.parametrize("prec", PARITY)
def test_variant_per_clip_timesteps_vs_oracle(env, g, prec):
    """LatentDiffusionModel.diffusion_forward's call pattern (ldm.py:75-97): one timestep per clip."""
    eng, orc, W, arch, pose, tag = (env[k] for k in ("eng", "orc", "W", "arch", "pose", "tag"))
    con, emo, sty, x = inputs(g, pose)
    ts = [int(v) for v in g["timesteps_batch"]]
    noise = torch.randn(*x.shape, generator=torch.Generator().manual_seed(5))
    out = eng.diffusion_forward(x, noise, ts, con, emo, sty, prec)
    ac = orc.SchedulerBase().alphas_cumprod
    sh = (-1,) + (1,) * (x.ndim - 1)
    noisy = ac[ts].sqrt().reshape(sh) * torch.from_numpy(x) + (1 - ac[ts]).sqrt().reshape(sh) * noise
    assert _err(out["noisy_latents"], noisy) < 1e-6
    ref = orc.denoiser_forward_variant(W, noisy, ts, *(torch.from_numpy(v) for v in (con, emo, sty)), arch, pose)
    assert _err(out["noise_pred"], ref) < 1e-5
    # and against the reference module itself where the noisy input is the golden's x (noise = 0 at sqrt_ab = 1 is not reachable:
    # the golden's eps_batch_t pins the oracle on the CPU - tests/test_oracle_variants.py)


@pytest.mark.parametrize("prec", PARITY)
def test_variant_ddim50_trajectory_vs_reference_golden(env, g, prec):
    from amuse_amd import scheduler as sch
    eng, pose, tag = env["eng"], env["pose"], env["tag"]
    con, emo, sty, x = inputs(g, pose)
    eng.set_schedule(sch.ddim_table())
    lat, traj = eng.sample(con, emo, sty, prec, x_init=x, return_traj=True)
    for n in (10, 50):
        assert _err(cut(traj[n - 1], pose), g[f"{tag}/x_after_{n}"]) < 1e-4, n
    assert torch.equal(lat, traj[-1])


def test_variant_sixteen_bit_modes(env, g):
    """bf16 / fp16 operands: whole-network bounds against the reference's goldens (|eps| ~ 3) - the 16-bit bars of
    tests/test_gpu_parity.py / test_gpu_fp16.py."""
    eng, pose, tag = env["eng"], env["pose"], env["tag"]
    con, emo, sty, x = inputs(g, pose)
    for prec, bar in (("bf16", 8e-2), ("fp16", 1.5e-2)):
        for t in (981, 1):
            e = eng.denoise_step(x, t, con, emo, sty, prec)
            assert torch.isfinite(e).all()
            assert _err(cut(e, pose), g[f"{tag}/eps_t{t}"]) < bar, (prec, t)


@pytest.mark.parametrize("prec", ("fp32", "bf16"))
def test_variant_ddpm_in_kernel_noise_shards_and_restatement(env, prec):
    """Ancestral sampling on a strided DDPM grid: (i) in-kernel counter noise == the same noise passed explicitly, bitwise;
    (ii) a job split into shards (clip_index0 offsets) == the job in one launch, bitwise; (iii) fp32: the loop vs the oracle's."""
    from amuse_amd import scheduler as sch
    eng, orc, W, arch, pose = (env[k] for k in ("eng", "orc", "W", "arch", "pose"))
    B, T = (5, 4) if pose else (21, 20)
    gq = torch.Generator().manual_seed(11)
    con, emo, sty = (torch.randn(B, 256, generator=gq) for _ in range(3))
    eng.set_schedule(sch.ddpm_table(T))
    full = eng.sample(con, emo, sty, prec, seed=77, clip_index0=40)
    x0 = eng.counter_normal(77, 40, B, 0, 0)
    nz = torch.stack([eng.counter_normal(77, 40, B, s, 1) for s in range(T)])
    expl = eng.sample(con, emo, sty, prec, x_init=x0, step_noise=nz)
    assert torch.equal(full, expl)
    k = 2 if pose else 16
    parts = torch.cat([eng.sample(con[:k], emo[:k], sty[:k], prec, seed=77, clip_index0=40),
                       eng.sample(con[k:], emo[k:], sty[k:], prec, seed=77, clip_index0=40 + k)])
    assert torch.equal(full, parts)
    ref0 = orc.counter_normal(77, np.arange(40, 40 + B), 0, 0, nfeat=int(np.prod(x0.shape[1:])))
    assert _err(x0.reshape(B, -1), ref0) < 5e-5   # hardware log2 / sin / cos vs libm
    if prec == "fp32":
        sched = orc.DDPM(T)
        nb = 2
        ref = orc.sample_variant(W, sched, con[:nb], emo[:nb], sty[:nb], x0[:nb].cpu(), arch, pose, step_noise=nz[:, :nb].cpu())
        assert _err(expl[:nb], ref) < 2e-4 * float(ref.abs().max())


def test_pose_step_fp32x_per_clip_kernel_samples_like_the_launches(env):
    """diffusion_only + trans_enc, fp32x, the whole step on the per-clip kernel (k_den_fusedx; "clip" pins it): the scheduler update inside it - DDIM with clipping off / on the
    schedule's terms, and ancestral DDPM with in-kernel counter noise - against the row / attention launches ("fused" pins them) on the same clips: x_0 within 1e-4; in-kernel
    noise == the same noise passed explicitly, bitwise; a job cut into shards (clip_index0) == the job in one launch, bitwise; ragged lengths through denoise_step."""
    if env["arch"] != "trans_enc" or not env["pose"]:
        pytest.skip("the diffusion_only + trans_enc variant")
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    B = 5
    gq = torch.Generator().manual_seed(13)
    con, emo, sty = (torch.randn(B, 256, generator=gq) for _ in range(3))
    try:
        for table in (sch.ddim_table(6), sch.ddpm_table(4)):
            eng.set_schedule(table)
            eng.set_decode_path("fused")
            want = eng.sample(con, emo, sty, "fp32x", seed=77, clip_index0=40)
            eng.set_decode_path("clip")
            got = eng.sample(con, emo, sty, "fp32x", seed=77, clip_index0=40)
            assert bool(torch.isfinite(got).all()) and _err(got, want) < 1e-4 * max(1.0, float(want.abs().max())) and not torch.equal(got, want)
            parts = [eng.sample(con[a:b], emo[a:b], sty[a:b], "fp32x", seed=77, clip_index0=40 + a) for a, b in ((0, 2), (2, 5))]
            assert torch.equal(torch.cat(parts), got)
        T = 4
        x0 = eng.counter_normal(77, 40, B, 0, 0)
        nz = torch.stack([eng.counter_normal(77, 40, B, s, 1) for s in range(T)])
        assert torch.equal(eng.sample(con, emo, sty, "fp32x", x_init=x0, step_noise=nz), got)     # (the DDPM-4 run above)
    finally:
        eng.set_decode_path("auto")


def test_pose_variant_diffusion_backward_converts_the_sampled_features(env):
    """diffusion_only: no VAE decode (infer_ldm.py:165) - the sampled [300][333] state goes through 6D -> axis-angle (:168-173)."""
    from amuse_amd import scheduler as sch
    eng, orc, pose = env["eng"], env["orc"], env["pose"]
    if not pose:
        pytest.skip("latent variants decode through MotionPrior (tests below)")
    gq = torch.Generator().manual_seed(3)
    con, emo, sty = (torch.randn(2, 256, generator=gq) for _ in range(3))
    eng.set_schedule(sch.ddim_table(3, num_train_timesteps=1000))
    out = eng.diffusion_backward(con, emo, sty, "fp32", seed=5)
    feats = out["latents"].cpu()
    poses, trans = orc.feats_to_smplx(feats.double())
    assert torch.equal(out["trans"].cpu(), feats[..., -3:])
    # as rotations (the axis-angle representation has the 2 pi ambiguity the candidate selection leaves): geodesic distance
    Rg, Rr = orc.axis_angle_to_matrix(out["poses"].cpu().double()), orc.axis_angle_to_matrix(poses)
    cosang = ((Rg.transpose(-1, -2) @ Rr).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    ang = torch.acos(cosang.clamp(-1, 1))
    assert float(ang.median()) < 1e-5 and float(ang.quantile(0.99)) < 1e-3
    with pytest.raises(Exception):
        eng.vae_decode(torch.zeros(1, 128))   # built without MotionPrior weights


def test_trans_dec_latent_end_to_end_decodes_through_the_prior(env):
    from amuse_amd import scheduler as sch
    eng, orc, W, arch, pose = (env[k] for k in ("eng", "orc", "W", "arch", "pose"))
    if pose:
        pytest.skip("pose-space variants never decode")
    from amuse_amd import weights as wts
    Wp = orc.to_torch(wts.make_prior_weights(0))
    gq = torch.Generator().manual_seed(9)
    con, emo, sty = (torch.randn(3, 256, generator=gq) for _ in range(3))
    x0 = torch.randn(3, 128, generator=gq)
    eng.set_schedule(sch.ddim_table())
    for prec in PARITY:
        out = eng.diffusion_backward(con, emo, sty, prec, x_init=x0)
        ref = orc.sample_variant(W, orc.DDIM(), con, emo, sty, x0, arch, pose)
        assert _err(out["latents"], ref) < 1e-4
        feats = orc.vae_decode(Wp, out["latents"].cpu())
        assert _err(out["trans"], feats[..., -3:]) < 2e-5


def test_variant_weight_update_equals_fresh_context(env, g):
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    eng, arch, pose = env["eng"], env["arch"], env["pose"]
    con, emo, sty, x = inputs(g, pose)
    w1 = wts.make_denoiser_weights(1, arch, pose)
    from amuse_amd import scheduler as sch
    fresh = HipEngine(w1, None if pose else wts.make_prior_weights(0), "cuda:0", arch=arch, diffusion_only=pose)
    try:
        fresh.set_schedule(sch.ddim_table())   # (installs torch's timestep frequencies, as every earlier call on `eng` did)
        eng.set_schedule(sch.ddim_table())
        eng.update_weights(denoiser_sd=w1)
        for prec in ("fp32", "fp32x", "bf16", "fp16"):
            assert torch.equal(eng.denoise_step(x, 501, con, emo, sty, prec), fresh.denoise_step(x, 501, con, emo, sty, prec)), prec
    finally:
        fresh.close()
        eng.update_weights(denoiser_sd=wts.make_denoiser_weights(0, arch, pose))


@pytest.mark.parametrize("arch,pose", VARIANTS)
def test_host_mirror_setup_reads_the_denoiser_variant_from_the_config(tmp_path, arch, pose):
    """PretrainedLPDM_v1.setup with configs/<arch>.json "arch_denoiser" naming a variant (infer_ldm.py:66-73 builds Denoiser from
    it): the checkpoint's entry count is asserted against THAT state dict (infer_ldm.py:103), diffusion_backward returns the
    reference's two keys; diffusion_only samples the feature sequence itself (the reference's loop cannot: see the mirror)."""
    import json
    from test_gpu_host_mirror import _mini_config
    from amuse_amd import checkpoint as ckpt, weights as wts
    from amuse_amd.infer_ldm import PretrainedLPDM_v1
    from oracle import amuse_oracle as orc
    cfg, processed, model_dir = _mini_config(tmp_path)
    p = processed.parents[1] / "configs/diff_latent_v2.json"
    ldm = json.load(open(p))
    ldm["arch_denoiser"] = {"nfeats": 201, "latent_dim": [1, 128], "ff_size": 512, "num_layers": 9, "num_heads": 4, "dropout": 0.1,
                            "arch": arch, "normalize_before": False, "activation": "gelu", "position_embedding": "learned",
                            "cond_dim": 256, "freq_shift": 0, "ablation_skip_connection": True, "pe_type": "mld",
                            "flip_sin_to_cos": True, "return_intermediate_dec": False, "diffusion_only": pose}
    json.dump(ldm, open(p, "w"))
    wd, wp = wts.make_denoiser_weights(0, arch, pose), wts.make_prior_weights(0)
    ckpt.save_reference_format(model_dir, wd, wp, epoch=6000, total=0.0123)
    m = PretrainedLPDM_v1(base_prior=None, audio_encoder=lambda wave: (torch.zeros(1, 256),) * 3)
    assert m.setup(cfg, "cuda:0", processed, None, False, verbose=False, diffonly=pose) == 6000
    assert (m.arch, m.diffusion_only) == (arch, pose) and m.latent_dim == ([300, 333] if pose else [1, 128])
    gen = torch.Generator().manual_seed(11)
    con, emo, sty = (torch.randn(2, 256, generator=gen) for _ in range(3))
    x = torch.randn(2, *m.latent_dim, generator=gen)[:, 0] if not pose else torch.randn(2, 300, 333, generator=gen)
    out = m.diffusion_backward(2, con, emo, sty, x_init=x, return_latents=True)
    assert set(out) == {"poses", "trans", "latents"} and out["poses"].shape == (2, 300, 55, 3) and out["trans"].shape == (2, 300, 3)
    ref = orc.sample_variant(orc.to_torch(wd), orc.DDIM(), con, emo, sty, x, arch, pose)
    assert _err(out["latents"], ref) < 1e-4
    # a checkpoint of another variant fails the count assertion, as the reference's loader does
    ckpt.save_reference_format(model_dir, wts.make_denoiser_weights(0), wp, epoch=6000, total=0.0123)
    with pytest.raises(AssertionError):
        PretrainedLPDM_v1(audio_encoder=lambda wave: None).setup(cfg, "cuda:0", processed, None, False, diffonly=pose)
    ldm["arch_denoiser"]["num_layers"] = 7
    json.dump(ldm, open(p, "w"))
    with pytest.raises(NotImplementedError):
        PretrainedLPDM_v1(audio_encoder=lambda wave: None).setup(cfg, "cuda:0", processed, None, False, diffonly=pose)


def test_fused_pose_step_kernel_vs_staged_and_golden(g):
    """k_den_fused (one persistent workgroup per clip, bf16 / fp16 operands; chosen from 64 clips up or pinned) against the staged
    kernels of the same mode (same arithmetic, other summation order), the reference's goldens (16-bit bars), the ragged-length
    zeroing, and its own invariants: batch position, in-kernel noise == explicit noise, shards - bitwise."""
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine
    arch, pose, tag = "trans_enc", True, "trans_enc_pose"
    eng = HipEngine(wts.make_denoiser_weights(0, arch, pose), None, "cuda:0", arch=arch, diffusion_only=pose)
    try:
        con, emo, sty, x = inputs(g, pose)
        for prec, bar, close in (("bf16", 8e-2, 6e-2), ("fp16", 1.5e-2, 1e-2)):
            eng.set_decode_path("staged")
            es = eng.denoise_step(x, 501, con, emo, sty, prec)
            eng.set_decode_path("fused")
            ef = eng.denoise_step(x, 501, con, emo, sty, prec)
            assert torch.isfinite(ef).all()
            assert _err(cut(ef, pose), g[f"{tag}/eps_t501"]) < bar, prec
            assert _err(ef, es) < close and float((ef - es).abs().mean()) < close / 8, prec
            e3 = eng.denoise_step(x, 501, con, None, sty, prec)                  # S = 303
            assert _err(cut(e3, pose), g[f"{tag}/eps_t501_noemo"]) < bar
            e2 = eng.denoise_step(x, 501, con, None, None, prec)                 # S = 302
            assert _err(cut(e2, pose), g[f"{tag}/eps_t501_consolo"]) < bar
            lens = [int(v) for v in g["lengths_ragged"]]
            er = eng.denoise_step(x, 501, con, emo, sty, prec, lengths=lens)
            assert torch.all(er[1, lens[1]:] == 0) and torch.equal(er[1, :lens[1]], ef[1, :lens[1]]) and torch.equal(er[0], ef[0])
            # clips are independent of their batch position
            x5 = np.concatenate([x[1:], x, x[:1]]); c5 = lambda v: np.concatenate([v[1:], v, v[:1]])
            e5 = eng.denoise_step(x5, 501, c5(con), c5(emo), c5(sty), prec)
            assert torch.equal(e5[1], ef[0]) and torch.equal(e5[0], ef[1]) and torch.equal(e5[3], ef[0])
        # sampling: DDIM-10 fused vs staged drift, and the DDPM invariants on the fused kernel
        B = 3
        gq = torch.Generator().manual_seed(21)
        cc, ce, cs = (torch.randn(B, 256, generator=gq) for _ in range(3))
        x0 = torch.randn(B, 300, 333, generator=gq)
        eng.set_schedule(sch.ddim_table(10))
        eng.set_decode_path("staged")
        ls = eng.sample(cc, ce, cs, "fp16", x_init=x0)
        l32 = eng.sample(cc, ce, cs, "fp32", x_init=x0)
        eng.set_decode_path("fused")
        lf = eng.sample(cc, ce, cs, "fp16", x_init=x0)
        assert float((lf - l32).abs().max()) < 2 * max(float((ls - l32).abs().max()), 1e-3)   # no worse than the staged fp16 mode's drift
        eng.set_schedule(sch.ddpm_table(4))
        full = eng.sample(cc, ce, cs, "bf16", seed=9, clip_index0=100)
        nz = torch.stack([eng.counter_normal(9, 100, B, s_, 1) for s_ in range(4)])
        assert torch.equal(full, eng.sample(cc, ce, cs, "bf16", x_init=eng.counter_normal(9, 100, B, 0, 0), step_noise=nz))
        parts = torch.cat([eng.sample(cc[:1], ce[:1], cs[:1], "bf16", seed=9, clip_index0=100),
                           eng.sample(cc[1:], ce[1:], cs[1:], "bf16", seed=9, clip_index0=101)])
        assert torch.equal(full, parts)
    finally:
        eng.close()
