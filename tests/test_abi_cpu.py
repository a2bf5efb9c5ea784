"""CPU: the C-ABI library loads and exports every symbol include/amuse_hip.h declares; host-side
tables and packing helpers.  No compute calls (no GPU here)."""
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def _declared_symbols():
    src = (REPO / "include/amuse_hip.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amuse_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from amuse_amd import _lib
    lib = _lib.load()
    decl = _declared_symbols()
    assert len(decl) >= 11
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in amuse_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == decl
    hdr = (REPO / "include/amuse_hip.h").read_text()
    assert lib.amuse_abi_version() == int(re.search(r"#define AMUSE_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION == 5


def test_f16_split_of_the_fp32x_mode_matches_numpy():
    """The host packer's hi / lo split (amuse_api.hip f2h / h2f, used for AMUSE_PREC_F32X's weight stream) against numpy's
    IEEE float16 conversion: normal range, subnormal results, ties, overflow, signed zeros - bit for bit - and the split
    keeps 22 significand bits."""
    import ctypes as C
    from amuse_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    w = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1.0, 0.05, 1e-3, 1e-5, 3e-7, 100.0, 3e4)]
                       + [np.array([0.0, -0.0, 1.0, 1.00048828125, 1.00146484375, 65504.0, 65519.9, 65520.0, -7e4, 6.1035e-5, 5.96e-8,
                                    2.98e-8, 2.9802325e-8, 8.94e-8, 1e-9], np.float32)])
    hi, lo = np.zeros(w.size, np.uint16), np.zeros(w.size, np.uint16)
    f, u = C.POINTER(C.c_float), C.POINTER(C.c_uint16)
    assert lib.amuse_debug_f16_split(w.ctypes.data_as(f), w.size, hi.ctypes.data_as(u), lo.ctypes.data_as(u)) == 0
    with np.errstate(over="ignore"):
        ref_hi = w.astype(np.float16)
        fin = np.isfinite(ref_hi)
        ref_lo = (w[fin] - ref_hi[fin].astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi, ref_hi.view(np.uint16))
    assert np.array_equal(lo[fin], ref_lo.view(np.uint16))
    both = hi.view(np.float16)[fin].astype(np.float64) + lo.view(np.float16)[fin].astype(np.float64)
    # 22 significand bits while the lo piece is a normal fp16 number (|w| >= 2^-2); below that it is subnormal and the
    # split is exact to half a unit of 2^-24 - MI355X's MFMA keeps subnormal fp16 operands (tools/probes/f16_denorm_probe.hip)
    assert np.all(np.abs(both - w[fin]) <= np.maximum(2.0 ** -22 * np.abs(w[fin]), 2.0 ** -25))


def test_create_rejects_bad_sizes_without_touching_the_gpu():
    import ctypes as C
    from amuse_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 4)()
    assert not lib.amuse_create(0, buf, 4, buf, 4)
    assert b"parameter count mismatch" in lib.amuse_last_error()
    assert not lib.amuse_create(0, None, 0, None, 0)


def test_engine_refuses_cpu_device():
    from amuse_amd import _lib, weights as wts
    from amuse_amd.engine import HipEngine
    with pytest.raises(_lib.AmuseHipError):
        HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0), device="cpu")


def test_schedule_tables_match_oracle_restatement():
    """amuse_amd.scheduler tables applied with the kernel's update formula == oracle step functions."""
    import torch
    from amuse_amd import scheduler as sch
    from oracle import amuse_oracle as orc

    def apply(row, x, eps, z):
        sb, sa, c0, cx, ce, sg, clip = (np.float32(v) for v in row[:7])
        x0 = (x - sb * eps) / sa
        if clip > 0:
            x0 = np.clip(x0, -clip, clip)
        nx = c0 * x0
        if cx != 0:
            nx = nx + cx * x
        if ce != 0:
            nx = nx + ce * eps
        if sg != 0:
            nx = nx + sg * z
        return nx

    rng = np.random.default_rng(0)
    x, eps, z = (rng.standard_normal(64).astype(np.float32) for _ in range(3))
    for tab, o in ((sch.ddim_table(), orc.DDIM()), (sch.ddpm_table(), orc.DDPM()), (sch.ddpm_table(100), orc.DDPM(100)),
                   (sch.ddim_table(eta=0.5), orc.DDIM(eta=0.5))):
        assert list(tab.timesteps) == o.timesteps
        for i in (0, 1, len(o.timesteps) // 2, len(o.timesteps) - 2, len(o.timesteps) - 1):
            t = o.timesteps[i]
            ref = o.step(torch.from_numpy(eps), t, torch.from_numpy(x), torch.from_numpy(z)).numpy()
            got = apply(tab.coef[i], x, eps, z)
            assert np.abs(ref - got).max() < 2e-6 * max(1.0, np.abs(ref).max()), (tab.kind, i)
            assert bool(tab.needs_noise()[i]) == o.needs_noise(t)
    with pytest.raises(ValueError):
        sch.ddim_table(1000)   # would index alphas_cumprod[1000]
    assert np.array_equal(sch.timestep_freqs(), orc.timestep_freqs().numpy())


def test_flatten_state_dict_checks_keys_and_shapes():
    from amuse_amd import weights as wts
    from amuse_amd.engine import flatten_state_dict
    w = wts.make_denoiser_weights(0)
    flat = flatten_state_dict(w, wts.denoiser_param_spec())
    assert flat.size == 2192384 and flat.dtype == np.float32
    bad = dict(w)
    bad.pop("encoder.norm.bias")
    with pytest.raises(KeyError):
        flatten_state_dict(bad, wts.denoiser_param_spec())
    bad = dict(w)
    bad["encoder.norm.bias"] = np.zeros(7, np.float32)
    with pytest.raises(ValueError):
        flatten_state_dict(bad, wts.denoiser_param_spec())


def test_schedule_known_answers():
    """Known-answer checks that do not go through the build's own restatement: the public constants of the
    scaled-linear (Stable-Diffusion) beta schedule the reference configures (configs/diff_latent_v2.json:48-66), the DDIM
    timestep grid, and the published closed forms - Ho et al. 2020 eq. 6-7 (posterior mean / variance, diffusers'
    `fixed_small`) and Song et al. 2021 eq. 12 with sigma = 0 - evaluated in float64 from first principles."""
    from amuse_amd import scheduler as sch
    ac = sch.alphas_cumprod()
    assert ac.dtype == np.float32 and ac.shape == (1000,)
    assert abs(float(ac[0]) - 0.99915) < 1e-6            # 1 - beta_start
    assert abs(float(ac[999]) - 0.0046601) < 2e-7        # SD's alphas_cumprod[-1] = 0.00466
    beta = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    assert abs(beta[0] - 0.00085) < 1e-15 and abs(beta[-1] - 0.012) < 1e-15
    ab = np.cumprod(1.0 - beta)
    assert np.abs(ac - ab).max() < 1e-6 and (np.abs(ac - ab) / ab).max() < 2e-6   # fp32 cumprod, as diffusers computes it
    # DDIM-50 grid of the inference scheduler: steps_offset 1, leading spacing -> 981, 961, ..., 1 (infer_ldm.py:143-144)
    d = sch.ddim_table()
    assert list(d.timesteps) == list(range(981, 0, -20)) and d.init_noise_sigma == 1.0
    abp = lambda t: ab[t] if t >= 0 else ab[0]            # set_alpha_to_one False: final_alpha_cumprod = alphas_cumprod[0]
    for i, t in enumerate(d.timesteps):
        sb, sa, c0, cx, ce, sg, clip, _ = d.coef[i]
        tp = int(t) - 20
        assert abs(sa - ab[t] ** 0.5) < 2e-6 and abs(sb - (1 - ab[t]) ** 0.5) < 2e-6
        assert abs(c0 - abp(tp) ** 0.5) < 2e-6 and abs(ce - (1 - abp(tp)) ** 0.5) < 2e-6     # eq. 12, sigma = 0
        assert cx == 0.0 and sg == 0.0 and clip == 1.0                                        # diffusers' clip_sample default
    # DDPM-1000 ancestral sampler: x_{t-1} = mu~(x_t, x0) + sigma_t z, Ho et al. eq. 6-7
    p = sch.ddpm_table()
    assert list(p.timesteps) == list(range(999, -1, -1))
    for i in (0, 1, 500, 998, 999):
        t = int(p.timesteps[i])
        sb, sa, c0, cx, ce, sg, clip, _ = p.coef[i]
        ab_t, ab_p = ab[t], (ab[t - 1] if t > 0 else 1.0)
        b_t = 1.0 - ab_t / ab_p
        assert abs(b_t - beta[t]) < 1e-12
        # diffusers evaluates these on fp32 tensors: 1 - abar_t cancels near t = 0 (1 - 0.9983 carries 3.5e-5 relative)
        tol = 2e-6 + 2.5e-7 / (1 - ab_t)
        assert abs(c0 - ab_p ** 0.5 * b_t / (1 - ab_t)) < tol                                 # eq. 7, x0 coefficient
        assert abs(cx - (1 - b_t) ** 0.5 * (1 - ab_p) / (1 - ab_t)) < tol                     # eq. 7, x_t coefficient
        var = (1 - ab_p) / (1 - ab_t) * b_t                                                   # beta~_t ("fixed_small")
        assert (sg == 0.0) if t == 0 else abs(sg - max(var, 1e-20) ** 0.5) < tol
        assert ce == 0.0 and clip == 0.0
    # consistency: if eps_hat is the noise that produced x_t from x0, a noiseless DDIM step lands on the x0 / eps ray
    rng = np.random.default_rng(1)
    x0 = np.clip(rng.standard_normal(16), -1, 1)
    eps = rng.standard_normal(16)
    i = 25
    t = int(d.timesteps[i])
    xt = ab[t] ** 0.5 * x0 + (1 - ab[t]) ** 0.5 * eps
    sb, sa, c0, cx, ce, sg, clip, _ = d.coef[i].astype(np.float64)
    nx = c0 * np.clip((xt - sb * eps) / sa, -clip, clip) + ce * eps
    assert np.abs(nx - (ab[t - 20] ** 0.5 * x0 + (1 - ab[t - 20]) ** 0.5 * eps)).max() < 1e-5


def test_training_entry_points_validate_their_arguments_without_touching_the_gpu():
    """The amuse_train_* family (csrc/k_train.hip, k_train_attn.hip) rejects bad shapes / missing pointers with AMUSE_EINVAL and a message before any HIP call."""
    import ctypes as C
    from amuse_amd import _lib
    lib = _lib.load()
    err = lambda: lib.amuse_last_error().decode()
    assert lib.amuse_train_ws_floats() >= 8 * 128 * 1024
    assert lib.amuse_train_set_lane(2) != 0 and "lane" in err() and lib.amuse_train_set_lane(-1) != 0          # two scratch lanes per device: 0 and 1
    assert lib.amuse_train_set_lane(1) == 0 and lib.amuse_train_set_lane(0) == 0
    one = 0x1000                                              # (a non-null address that is never dereferenced: every call below fails in its checks)
    assert lib.amuse_train_ln_fwd(None, None, None, one, one, 0.1, 1, 1, 16, one, None, None, None) != 0 and "must be given" in err()
    assert lib.amuse_train_ln_fwd(None, one, None, one, one, 0.1, 1, 1, 0, one, None, None, None) != 0 and "rows" in err()
    assert lib.amuse_train_ln_fwd(None, one, None, one, one, 1.0, 1, 1, 16, one, None, None, None) != 0 and "dropout" in err()
    assert lib.amuse_train_ln_bwd(one, None, one, one, one, 0.0, 1, 1, 16, None, one, None, None, None, None, None) != 0 and "NULL" in err()
    assert lib.amuse_train_bias_gelu_drop_fwd(one, one, 0.0, 1, 1, 16, 510, one, None) != 0 and "multiple of 4" in err()
    assert lib.amuse_train_bias_gelu_drop_bwd(one, one, one, 0.0, 1, 1, 16, 2048, one, one, one, None) != 0
    assert lib.amuse_train_colsum(one, 16, 6, one, one, None) != 0
    assert lib.amuse_train_linear_fwd(one, one, None, 16, 128, 0, one, None) != 0 and "N 0" in err()      # (any width >= 1 is taken since round 6: N = 333 included)
    assert lib.amuse_train_linear_bwd(one, one, one, 16, 128, 2048, one, one, one, 0, one, None) != 0 and "up to 1024" in err()
    assert lib.amuse_train_linear_bwd(one, one, one, 16, 128, 384, one, one, one, 0, None, None) != 0 and "workspace" in err()
    assert lib.amuse_train_adamw(one, one, one, one, 0, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, None) != 0
    assert lib.amuse_train_adamw(one, one, one, one, 8, 1e-4, 0.9, 0.999, 1e-8, 0.01, 0, None) != 0 and "step" in err()
    assert lib.amuse_train_attn_fwd(one, 2, 305, 0.0, 1, 1, one, one, None, None) != 0 and "1..304" in err()
    assert lib.amuse_train_attn_fwd(None, 2, 300, 0.0, 1, 1, one, one, None, None) != 0
    assert lib.amuse_train_attn_bwd(one, one, one, None, 2, 300, 0.0, 1, 1, one, None) != 0
    L = _lib.TrainLayer()
    assert lib.amuse_train_layer_fwd(None, None) != 0 and "NULL" in err()
    L.rows, L.B, L.S, L.H, L.ff = 600, 2, 299, 4, 512
    assert lib.amuse_train_layer_fwd(C.byref(L), None) != 0 and "rows" in err()
    L.S = 300
    assert lib.amuse_train_layer_fwd(C.byref(L), None) != 0 and "required pointer" in err()
    L.H = 3
    assert lib.amuse_train_layer_bwd(C.byref(L), None) != 0 and "heads" in err()

