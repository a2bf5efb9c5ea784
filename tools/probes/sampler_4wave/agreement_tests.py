"""Shelved with the 4-wave bf16 / fp32x sampler instantiations (round 6; see README.md in this directory): the two GPU tests that held the
4-wave kernel's bf16 and fp32x instantiations against the 8-wave kernels the modes run on.  They passed on every round-1..5 tree (profiles/r05_gpu_pytest_tail.txt).
To run them again: apply restore_4wave.patch (re-adds the instantiations, their weight streams and the AMUSE_SAMPLE_WAVES=4 switch), rebuild, and paste these two
functions back into tests/test_gpu_parity.py (they use that module's `env` fixture, GOLDEN and _err)."""
import numpy as np


def test_bf16_kernels_4_and_8_waves_agree(env):
    """The 4-wave bf16 kernel (AMUSE_SAMPLE_WAVES=4, kept for A/B measurements) and the 8-wave one compute the same
    network with different summation orders and GELU evaluations: teacher-forced eps_hat and a DDIM-50 run stay within
    the whole-network bf16 tolerance of each other."""
    import os, subprocess, sys, tempfile
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    tr = np.load(GOLDEN / "ddim50_traj.npz")
    con, emo, sty, x = (g[k] for k in ("con", "emo", "sty", "x_t"))
    eps8 = eng.denoise_step(x, 981, con, emo, sty, "bf16").cpu().numpy()
    eng.set_schedule(sch.ddim_table())
    lat8 = eng.sample(tr["con"], tr["emo"], tr["sty"], "bf16", x_init=tr["x_T"]).cpu().numpy()
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "from amuse_amd import weights as wts, scheduler as sch\n"
        "from amuse_amd.engine import HipEngine\n"
        "eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))\n"
        "g = np.load(sys.argv[1] + '/tests/golden/denoiser_steps.npz'); tr = np.load(sys.argv[1] + '/tests/golden/ddim50_traj.npz')\n"
        "eps = eng.denoise_step(g['x_t'], 981, g['con'], g['emo'], g['sty'], 'bf16').cpu().numpy()\n"
        "eng.set_schedule(sch.ddim_table())\n"
        "lat = eng.sample(tr['con'], tr['emo'], tr['sty'], 'bf16', x_init=tr['x_T']).cpu().numpy()\n"
        "np.savez(sys.argv[2], eps=eps, lat=lat)\n")
    repo = str(GOLDEN.parents[1])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w4.npz")
        subprocess.run([sys.executable, "-c", code, repo, out], check=True, env=dict(os.environ, AMUSE_SAMPLE_WAVES="4"),
                       timeout=600)
        w4 = np.load(out)
        assert not np.array_equal(w4["eps"], eps8)            # really two kernels
        assert _err(w4["eps"], eps8) < 5e-2
        assert _err(w4["lat"], lat8) < 0.3                     # bf16 drift class over 50 steps (latent rms 0.53)


def test_fp32x_kernels_4_and_8_waves_agree(env):
    """fp32x runs on the 8-wave role-split kernel (k_sampler8x.hip); the 4-wave kernel's PREC_F16X2 instantiation (k_sampler.hip,
    AMUSE_SAMPLE_WAVES=4) computes the same network with another summation order: both hold the parity bars, so they agree
    with each other at that level - and they really are two kernels."""
    import os, subprocess, sys, tempfile
    from amuse_amd import scheduler as sch
    eng = env["eng"]
    g = np.load(GOLDEN / "denoiser_steps.npz")
    tr = np.load(GOLDEN / "ddim50_traj.npz")
    eps8 = eng.denoise_step(g["x_t"], 981, g["con"], g["emo"], g["sty"], "fp32x").cpu().numpy()
    eng.set_schedule(sch.ddim_table())
    lat8 = eng.sample(tr["con"], tr["emo"], tr["sty"], "fp32x", x_init=tr["x_T"]).cpu().numpy()
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "from amuse_amd import weights as wts, scheduler as sch\n"
        "from amuse_amd.engine import HipEngine\n"
        "eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))\n"
        "g = np.load(sys.argv[1] + '/tests/golden/denoiser_steps.npz'); tr = np.load(sys.argv[1] + '/tests/golden/ddim50_traj.npz')\n"
        "eps = eng.denoise_step(g['x_t'], 981, g['con'], g['emo'], g['sty'], 'fp32x').cpu().numpy()\n"
        "eng.set_schedule(sch.ddim_table())\n"
        "lat = eng.sample(tr['con'], tr['emo'], tr['sty'], 'fp32x', x_init=tr['x_T']).cpu().numpy()\n"
        "np.savez(sys.argv[2], eps=eps, lat=lat)\n")
    repo = str(GOLDEN.parents[1])
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "w4.npz")
        subprocess.run([sys.executable, "-c", code, repo, out], check=True, env=dict(os.environ, AMUSE_SAMPLE_WAVES="4"), timeout=600)
        w4 = np.load(out)
        assert not np.array_equal(w4["eps"], eps8)
        assert _err(w4["eps"], eps8) < 2e-5 and _err(w4["eps"], g["eps_t981"]) < 1e-5
        assert _err(w4["lat"], lat8) < 1e-4 and _err(w4["lat"], tr["x_after_50"]) < 1e-4


