"""GPU: the fp16 throughput mode (AMUSE_PREC_F16) - the bf16 mode's kernels (8-wave sampler k_sampler8.hip, fused decoder
k_vae_fused.hip) built for fp16 MFMA operands: same speed and bytes, 11 significand bits instead of 8.  Checked like the bf16
mode - per block on the kernel's own inputs against the oracle emulating ITS roundings - and gated against the fp32 mode at a
fraction of the bf16 mode's bounds."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


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
    yield {"eng": eng, "Wd": orc.to_torch(wd), "Wp": orc.to_torch(wp), "orc": orc, "wd": wd, "wp": wp}
    eng.set_decode_path("auto")
    eng.close()


def test_fp16_denoise_step_blockwise_and_vs_golden(env):
    orc, eng, Wd = env["orc"], env["eng"], env["Wd"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_t"))
    eps, tap = eng.denoise_step(x, 981, con, emo, sty, "fp16", taps=True)
    # whole network against the reference's own modules: an eighth of the bf16 mode's error (2.4e-2 -> measured ~3e-3)
    for t in (981, 501, 1):
        assert _err(eng.denoise_step(x, t, con, emo, sty, "fp16"), g[f"eps_t{t}"]) < 1e-2, t
    assert _err(eps, orc.denoiser_forward(Wd, x, 981, con, emo, sty, fp16=True)) < 1e-2
    # per block, teacher-forced on the kernel's own block inputs, against the oracle rounding the same operands to fp16
    tp = tap.cpu()
    ops = orc.Ops(True, poly_gelu=True, fp16=True)
    names = [f"encoder.input_blocks.{i}" for i in range(4)] + ["encoder.middle_block"] + [f"encoder.output_blocks.{i}" for i in range(4)]
    errs, meds = [], []
    for b, n in enumerate(names):
        xin = tp[b, :5][None]
        if b >= 5:
            sk = tp[1 + (8 - b), :5][None]
            xin = ops.lin(torch.cat([xin, sk], -1), Wd[f"encoder.linear_blocks.{b - 5}.weight"], Wd[f"encoder.linear_blocks.{b - 5}.bias"])
        ref = orc.enc_block(ops, xin, Wd, n)
        errs.append(_err(tp[b + 1, :5][None], ref))
        meds.append(float((tp[b + 1, :5][None] - ref).abs().median()))
    print("fp16 per-block max", ["%.1e" % v for v in errs], "median", ["%.1e" % v for v in meds])
    # an operand whose fp32 value sits on an fp16 rounding boundary may round the other way in the kernel (other summation
    # order): that moves a block output by O(2^-12 |x|) ~ 1e-4..1e-3 (bf16 mode: 1e-3..2e-2, bar 2e-2); everything else agrees
    # to fp32 summation order
    assert max(errs) < 1.5e-3, errs
    assert sorted(meds)[4] < 1e-6 and max(meds) < 2e-4, meds   # (a flip in a block's first GEMM moves its whole row: median 1e-5)
    # token dropping, every tiling
    try:
        for G in (0, 1, 2, 3):
            eng.set_clips_per_group(G)
            assert _err(eng.denoise_step(x, 501, con, None, sty, "fp16"), g["eps_t501_noemo"]) < 1e-2
            assert _err(eng.denoise_step(x, 501, con, None, None, "fp16"), g["eps_t501_consolo"]) < 1e-2
    finally:
        eng.set_clips_per_group(0)
    # really another kernel than the bf16 one
    assert not torch.equal(eps, eng.denoise_step(x, 981, con, emo, sty, "bf16"))


def test_fp16_mode_gated_against_fp32_mode(env):
    """Same inputs + noise through fp32, fp16 and bf16, 64 clips: the fp16 mode's drift is a fraction of the bf16 mode's
    (operand rounding 2^-12 against 2^-9) - gated at a quarter of the bf16 gates of test_gpu_configs.py."""
    from amuse_amd import scheduler as sch
    orc, eng = env["orc"], env["eng"]
    g = torch.Generator().manual_seed(2024)
    B = 64
    c, e, s, x = (torch.randn(B, n, generator=g) for n in (256, 256, 256, 128))

    def geodesic_deg(lat_ref, lat, prec):
        pa = eng.vae_decode(lat_ref, None, "fp32")["poses"].cpu()
        pb = eng.vae_decode(lat, None, prec)["poses"].cpu()
        Ra, Rb = orc.axis_angle_to_matrix(pa.double()), orc.axis_angle_to_matrix(pb.double())
        return torch.acos(((Ra.transpose(-1, -2) @ Rb).diagonal(dim1=-2, dim2=-1).sum(-1) - 1).div(2).clamp(-1, 1)) * 180 / np.pi

    eng.set_schedule(sch.ddim_table())
    a = eng.sample(c, e, s, "fp32", x_init=x).cpu()
    h = eng.sample(c, e, s, "fp16", x_init=x).cpu()
    b = eng.sample(c, e, s, "bf16", x_init=x).cpu()
    rms_h, rms_b = float((a - h).pow(2).mean().sqrt()), float((a - b).pow(2).mean().sqrt())
    print(f"DDIM-50 latent drift vs fp32: fp16 rms {rms_h:.4f} max {float((a - h).abs().max()):.4f}; bf16 rms {rms_b:.4f}")
    assert rms_h < 0.02 and float((a - h).abs().max()) < 0.115      # bf16 gates: 0.08 / 0.46
    assert rms_h < 0.35 * rms_b
    ang_h, ang_b = geodesic_deg(a, h, "fp16"), geodesic_deg(a, b, "bf16")
    p99 = lambda t: float(t.flatten().kthvalue(int(t.numel() * 0.99)).values)
    print(f"DDIM-50 pose geodesic vs fp32 (deg): fp16 median {float(ang_h.median()):.3f} p99 {p99(ang_h):.2f}; bf16 median {float(ang_b.median()):.3f} p99 {p99(ang_b):.2f}")
    assert float(ang_h.median()) < 1.8 and p99(ang_h) < 15.0           # bf16 gates: 7.2 / 61
    assert float(ang_h.median()) < 0.5 * float(ang_b.median())
    tab = sch.ddpm_table()
    eng.set_schedule(tab)
    nz = torch.randn(tab.n_steps, B, 128, generator=g)
    a = eng.sample(c, e, s, "fp32", x_init=x, step_noise=nz).cpu()
    h = eng.sample(c, e, s, "fp16", x_init=x, step_noise=nz).cpu()
    assert float((a - h).pow(2).mean().sqrt()) < 0.09                 # bf16 gate: 0.35 (latent rms 32)
    ang = geodesic_deg(a, h, "fp16")
    print(f"DDPM-1000 fp16 vs fp32: latent rms {float((a - h).pow(2).mean().sqrt()):.4f}; pose geodesic median {float(ang.median()):.3f} p99 {p99(ang):.2f}")
    assert float(ang.median()) < 0.25 and p99(ang) < 2.0               # bf16 gates: 1.0 / 8.0


def test_fp16_fused_decode_vs_golden_and_oracle(env):
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    g = np.load(GOLDEN / "vae_decode.npz")
    eng.set_decode_path("fused")                                          # (3 clips would run the staged kernels)
    out = eng.vae_decode(g["z"], None, "fp16", return_feats=True)
    assert out["feats"].shape == (3, 300, 333) and bool(torch.isfinite(out["feats"]).all())
    assert _err(out["feats"], g["feats"]) < 1e-2                          # bf16: 6e-2 (|feats| ~ 3)
    ref = orc.vae_decode(Wp, torch.from_numpy(g["z"]), None, fp16=True)
    d = (out["feats"].cpu() - ref).abs()
    assert float(d.max()) < 8e-3 and float(d.mean()) < 8e-4              # bf16: 5e-2 / 5e-3
    # ragged lengths: masked frames zeroed, valid part independent of the batch
    z = torch.randn(4, 128, generator=torch.Generator().manual_seed(5))
    o = eng.vae_decode(z, [300, 173, 1, 17], "fp16", return_feats=True)
    assert float(o["feats"][1, 173:].abs().max()) == 0.0 and float(o["poses"][2, 1:].abs().max()) == 0.0
    o2 = eng.vae_decode(z[[1]], [173], "fp16", return_feats=True)
    assert torch.equal(o2["feats"][0], o["feats"][1])
    assert not torch.equal(out["feats"], eng.vae_decode(g["z"], None, "bf16", return_feats=True)["feats"])
    eng.set_decode_path("auto")


def test_fp16_staged_decode_and_encode(env):
    """Below 64 clips the fp16 mode decodes on the staged kernels' fp16 instantiations (k_vae.hip k_vae_rows<PREC_F16> /
    k_vae_attn_bf16<PREC_F16>), and MotionPrior.encode runs on them at any size: against the reference goldens at a fraction of
    the bf16 bounds, against the oracle emulating the roundings, and against the fused kernel (another summation order)."""
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    g = np.load(GOLDEN / "vae_decode.npz")
    z = torch.from_numpy(g["z"])
    st = eng.vae_decode(z, None, "fp16", return_feats=True)               # 3 clips: staged
    assert _err(st["feats"], g["feats"]) < 1e-2                            # bf16: 6e-2
    ref = orc.vae_decode(Wp, z, None, fp16=True)
    d = (st["feats"].cpu() - ref).abs()
    assert float(d.max()) < 8e-3 and float(d.mean()) < 8e-4
    eng.set_decode_path("fused")
    fu = eng.vae_decode(z, None, "fp16", return_feats=True)
    eng.set_decode_path("auto")
    assert not torch.equal(st["feats"], fu["feats"]) and _err(st["feats"], fu["feats"]) < 8e-3
    o = eng.vae_decode(z, [300, 41, 7], "fp16", return_feats=True)         # ragged, staged
    assert float(o["feats"][1, 41:].abs().max()) == 0.0 and float(o["poses"][2, 7:].abs().max()) == 0.0
    ge = np.load(GOLDEN / "vae_encode.npz")
    feats = torch.from_numpy(ge["feats"].astype(np.float32))
    out = eng.vae_encode(feats, None, "fp16")
    assert _err(out["mu"], ge["mu"]) < 1e-2                                # bf16: 6e-2 (|mu| ~ 3)
    assert _err(out["std"] / torch.from_numpy(ge["std"]).to(out["std"].device), torch.ones(2, 128)) < 1e-2   # bf16: 5e-2
    mu_ref, std_ref = orc.vae_encode(Wp, feats, None, fp16=True)
    assert _err(out["mu"], mu_ref) < 8e-3
    assert not torch.equal(out["mu"], eng.vae_encode(feats, None, "bf16")["mu"])


def test_fp16_shards_noise_and_updates_are_bitwise(env):
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine
    eng = env["eng"]
    gen = torch.Generator().manual_seed(9)
    c, e, s = (torch.randn(8, 256, generator=gen) for _ in range(3))
    eng.set_schedule(sch.ddpm_table(50))
    full = eng.sample(c, e, s, "fp16", seed=2024, clip_index0=16)
    a = eng.sample(c[:4], e[:4], s[:4], "fp16", seed=2024, clip_index0=16)
    b = eng.sample(c[4:], e[4:], s[4:], "fp16", seed=2024, clip_index0=20)
    assert torch.equal(full, torch.cat([a, b])) and bool(torch.isfinite(full).all())
    x0 = eng.counter_normal(2024, 16, 8, 0, 0)
    nz = torch.stack([eng.counter_normal(2024, 16, 8, st, 1) for st in range(50)])
    assert torch.equal(full, eng.sample(c, e, s, "fp16", x_init=x0, step_noise=nz))
    # weight updates (host and device path) == a fresh context on the new weights
    w1 = (wts.make_denoiser_weights(1), wts.make_prior_weights(1))
    fresh, upd = HipEngine(*w1), HipEngine(env["wd"], env["wp"])
    for en in (fresh, upd):
        en.set_schedule(sch.ddim_table())
    x = torch.randn(8, 128, generator=gen)
    before = upd.sample(c, e, s, "fp16", x_init=x)
    upd.update_weights(*w1, what=16)
    la, lb = upd.sample(c, e, s, "fp16", x_init=x), fresh.sample(c, e, s, "fp16", x_init=x)
    assert torch.equal(la, lb) and not torch.equal(la, before)
    assert torch.equal(upd.vae_decode(la, None, "fp16")["poses"], fresh.vae_decode(lb, None, "fp16")["poses"])
    from amuse_amd.engine import flatten_state_dict
    f0 = (torch.from_numpy(flatten_state_dict(env["wd"], wts.denoiser_param_spec())).cuda(),
          torch.from_numpy(flatten_state_dict(env["wp"], wts.prior_param_spec())).cuda())
    upd.update_weights_device(*f0, what=16)
    assert torch.equal(upd.sample(c, e, s, "fp16", x_init=x), before)
    fresh.close(); upd.close()


def test_fp16_mode_range_headroom(env):
    """fp16 operands overflow at 65504 where bf16 does not (amuse_hip.h): condition embeddings 30 x the usual scale (token magnitudes in
    the hundreds - LayerNorm brings every later operand back to O(1)) still sample and decode to finite poses, and stay close to the
    bf16 mode's result on the same inputs (both are roundings of the same fp32 computation)."""
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    g = torch.Generator().manual_seed(77)
    c, e, s, x = (30.0 * torch.randn(8, n, generator=g) for n in (256, 256, 256, 128))
    x = x / 30.0
    eng.set_schedule(sch.ddim_table())
    h = eng.diffusion_backward(c, e, s, "fp16", x_init=x)
    b = eng.diffusion_backward(c, e, s, "bf16", x_init=x)
    f = eng.diffusion_backward(c, e, s, "fp32", x_init=x)
    assert bool(torch.isfinite(h["poses"]).all()) and bool(torch.isfinite(h["latents"]).all())
    dh = float((h["latents"] - f["latents"]).pow(2).mean().sqrt())
    db = float((b["latents"] - f["latents"]).pow(2).mean().sqrt())
    assert dh < db, (dh, db)      # closer to the fp32 mode than bf16 is, also at this scale


def test_fp16_overflow_surfaces_as_non_finite_never_clips(env):
    """The price of the fp16 mode is range: a GEMM operand beyond 65504 becomes inf in the fp16 pack (v_cvt_pk_f16_f32 rounds to
    infinity, it does not saturate) and the outputs go non-finite - loudly - where the bf16 mode carries the same inputs.  Through the
    C ABI: the teacher-forced step, the sampling loop and the decode.  (amuse_hip.h AMUSE_PREC_F16; the policy line of INTEGRATION.md:
    fp32x to match the reference, fp16 for throughput, bf16 only for activations beyond fp16 range.)"""
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    g = torch.Generator().manual_seed(5)
    c, e, s, x = (torch.randn(4, n, generator=g) for n in (256, 256, 256, 128))
    big = x.clone()
    big[1, 7] = 3.0e5                       # one latent feature of clip 1 beyond fp16 range: the row is an operand of block 0's in_proj
    h = eng.denoise_step(big, 501, c, e, s, "fp16")
    assert not bool(torch.isfinite(h[1]).all()), "an fp16 overflow must not be clipped silently"
    assert bool(torch.isfinite(h[[0, 2, 3]]).all())             # clips are independent: the others are untouched
    assert torch.equal(h[[0, 2, 3]], eng.denoise_step(x, 501, c, e, s, "fp16")[[0, 2, 3]])
    assert bool(torch.isfinite(eng.denoise_step(big, 501, c, e, s, "bf16")).all())      # bf16 carries it
    assert bool(torch.isfinite(eng.denoise_step(big, 501, c, e, s, "fp32x")[[0, 2, 3]]).all())
    eng.set_schedule(sch.ddim_table(5))
    lat = eng.sample(c, e, s, "fp16", x_init=big)
    assert not bool(torch.isfinite(lat[1]).all()) and bool(torch.isfinite(lat[[0, 2, 3]]).all())
    zbig = torch.randn(64, 128, generator=g)
    zbig[3] *= 1.0e6                        # the one-token memory of clip 3: its cross-attention constant overflows the first operand pack
    for path in ("staged", "fused"):
        eng.set_decode_path(path)
        out = eng.vae_decode(zbig, None, "fp16", return_feats=True)["feats"]
        eng.set_decode_path("auto")
        assert bool(torch.isfinite(out[:3]).all()) and bool(torch.isfinite(out[4:]).all()), path
        # (LayerNorm of a row dominated by one huge constant is still finite: what must never happen is a silently clipped value)
        if bool(torch.isfinite(out[3]).all()):
            ref = eng.vae_decode(zbig, None, "fp32", return_feats=True)["feats"][3]
            assert float((out[3] - ref).abs().max()) < 0.5, path
