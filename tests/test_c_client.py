"""The drop-in boundary from plain C: tests/c_client/client.c (C99, -pedantic -Werror, no torch / Python / HIP headers) is built
against include/amuse_hip.h and libamuse_hip.so.  CPU: the error conventions without a GPU.  GPU: a whole job - create, schedule,
diffusion_backward - from files, bitwise what the Python host mirror gets through the same ABI."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def _build(tmp_path) -> Path:
    from amuse_amd import _lib
    _lib.load()                                                   # builds nothing; fails loudly if the library is missing
    exe = tmp_path / "amuse_c_client"
    lib_dir = REPO / "amuse_amd"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", str(REPO / "include"),
           str(REPO / "tests" / "c_client" / "client.c"), "-o", str(exe), "-L", str(lib_dir), "-lamuse_hip",
           "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_client_builds_and_sees_the_error_conventions(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe), "abi"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ABI_OK 5", (r.stdout, r.stderr)
    assert subprocess.run([str(exe)], capture_output=True).returncode == 2
    # the launch plan from plain C == the Python mirror's (both are calls of amuse_plan; neither restates the rule)
    from amuse_amd import _lib
    for clips, prec, tokens in ((1, 0, 5), (64, 1, 5), (200, 2, 4), (256, 2, 5), (300, 3, 3), (4096, 2, 5)):
        r = subprocess.run([str(exe), "plan", str(clips), str(prec), str(tokens)], capture_output=True, text=True, timeout=60)
        p = _lib.plan(clips, prec, tokens)
        want = f"PLAN {p['clips_per_group']} " + " ".join(str(_lib.DECODE_PATHS.index(p[k])) for k in ("decode_path", "encode_path", "step_path"))
        assert r.returncode == 0 and r.stdout.strip() == want, (r.stdout, r.stderr, want)
    assert subprocess.run([str(exe), "plan", "0", "0", "5"], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_c_client_job_equals_the_python_host_mirror(tmp_path):
    import torch
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine, flatten_state_dict
    exe = _build(tmp_path)
    wd, wp = wts.make_denoiser_weights(3), wts.make_prior_weights(3)
    B, seed = 5, 77
    table = sch.ddim_table()
    flatten_state_dict(wd, wts.denoiser_param_spec()).tofile(tmp_path / "den.f32")
    flatten_state_dict(wp, wts.prior_param_spec()).tofile(tmp_path / "prior.f32")
    with open(tmp_path / "sched.bin", "wb") as f:
        f.write(np.int32(table.n_steps).tobytes())
        f.write(np.ascontiguousarray(table.timesteps, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(table.coef, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(sch.timestep_freqs(), dtype=np.float32).tobytes())
    cond = torch.randn(3, B, 256, generator=torch.Generator().manual_seed(5))
    cond.numpy().tofile(tmp_path / "cond.f32")
    r = subprocess.run([str(exe), "run", str(tmp_path), str(B), str(seed)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ))
    assert r.returncode == 0 and r.stdout.strip() == f"RUN_OK B={B} T=50", (r.stdout, r.stderr)
    eng = HipEngine(wd, wp)
    eng.set_schedule(table)
    out = eng.diffusion_backward(cond[0], cond[1], cond[2], "bf16", "p3d", seed=seed, clip_index0=0)
    for name, key, shape in (("out_latents.f32", "latents", (B, 128)), ("out_poses.f32", "poses", (B, 300, 55, 3)),
                             ("out_trans.f32", "trans", (B, 300, 3))):
        got = np.fromfile(tmp_path / name, dtype=np.float32).reshape(shape)
        assert np.isfinite(got).all() and np.array_equal(got, out[key].cpu().numpy()), name
    eng.close()
