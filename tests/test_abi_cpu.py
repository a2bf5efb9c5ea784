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
    assert lib.amuse_abi_version() == 1


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
