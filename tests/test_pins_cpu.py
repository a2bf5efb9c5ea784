"""CPU: the third-party arithmetic of the path - pytorch3d's matrix_to_axis_angle and diffusers' DDPM / DDIM updates - pinned
against material the REFERENCE TREE itself holds (fixtures written by oracle/gen_golden.py --pins-only):

  * tests/golden/ref_poses.npz: the `poses` of the three committed sample outputs viz_dump/test/**/*_motion_smplx.npz, i.e.
    outputs of the deployed pytorch3d.matrix_to_axis_angle (infer_ldm.py:171-172).
  * tests/golden/sched_ref.npz: tables and steps of the reference tree's own GaussianDiffusion / SpacedDiffusion
    (models/diffusion/utils/mdm_gaussian_diffusion.py:198-278,323-366,528-533,690,895-940; mdm_respace.py:64-87).

Both the oracle (oracle/amuse_oracle.py) and the product's host tables (amuse_amd/scheduler.py - what the HIP kernels consume)
are checked.  The coefficient formula of the kernels' update (include/amuse_hip.h, amuse_schedule) is applied here in numpy
float32; the -m gpu counterpart (tests/test_gpu_pins.py) runs the same fixtures through the C ABI.
"""
import numpy as np
import pytest
import torch

from amuse_amd import scheduler as sch
from oracle import amuse_oracle as orc
from conftest import GOLDEN

# float32 (diffusers: torch-fp32 cumprod, 1 - abar in fp32) against the reference tree's fp64 tables: bound on every
# coefficient.  The largest gaps sit at t <= 2 where 1 - abar ~ 1e-3 carries 2^-24 / 1e-3 ~ 6e-5 of relative rounding.
TABLE_RTOL = 3e-5


@pytest.fixture(scope="module")
def ref():
    return np.load(GOLDEN / "sched_ref.npz")


@pytest.fixture(scope="module")
def poses():
    z = np.load(GOLDEN / "ref_poses.npz")
    return {k: z[k] for k in z.files}


# ------------------------------------------------------------------------------------------------ A15
def test_ref_poses_shape_and_beyond_pi_population(poses):
    assert len(poses) == 3 and all(v.shape == (300, 55, 3) and v.dtype == np.float32 for v in poses.values())
    allp = np.concatenate([v.reshape(-1, 3) for v in poses.values()])
    assert allp.shape[0] == 49500
    assert int((np.linalg.norm(allp, axis=-1) > np.pi).sum()) == 10


def test_oracle_p3d_axis_angle_reproduces_the_committed_outputs(poses):
    """matrix_to_axis_angle(axis_angle_to_matrix(p), "p3d") == p for every joint the reference wrote, the |aa| > pi ones
    included; the vendored ("legacy") variant misses exactly those."""
    for name, p in poses.items():
        aa = torch.from_numpy(p).reshape(-1, 3)
        m = orc.axis_angle_to_matrix(aa.double()).float()       # the rotation itself, free of fp32 composition error
        back = orc.matrix_to_axis_angle(m, "p3d")
        err = torch.linalg.vector_norm(back - aa, dim=-1)
        assert float(err.max()) <= 1e-5, (name, float(err.max()))
        legacy = orc.matrix_to_axis_angle(m, "legacy")
        miss = torch.linalg.vector_norm(legacy - aa, dim=-1) > 1e-3
        big = torch.linalg.vector_norm(aa, dim=-1) > np.pi
        assert torch.equal(miss, big), name


def test_oracle_p3d_fp32_matrix_path(poses):
    """Same, through the float32 axis_angle_to_matrix the product's own round trip uses (amuse_smplx_to_feats)."""
    for name, p in poses.items():
        aa = torch.from_numpy(p).reshape(-1, 3)
        back = orc.matrix_to_axis_angle(orc.axis_angle_to_matrix(aa), "p3d")
        assert float(torch.linalg.vector_norm(back - aa, dim=-1).max()) <= 2e-5, name


# ------------------------------------------------------------------------------------------------ A11
def test_alphas_cumprod_fp32_vs_reference_fp64(ref):
    ac = sch.alphas_cumprod()
    assert np.abs(ac / ref["alphas_cumprod"] - 1).max() < 5e-6
    assert np.abs(orc.SchedulerBase().alphas_cumprod.numpy() / ref["alphas_cumprod"] - 1).max() < 5e-6


def test_ddpm_table_matches_reference_posterior(ref):
    """amuse_amd.scheduler.ddpm_table rows [sb, sa, c0, cx, ce, sigma, clip, 0] against the reference object's arrays:
    x0 = (x - sb eps) / sa  <->  sqrt_recip x - sqrt_recipm1 eps;  c0 / cx <-> posterior_mean_coef1 / 2;
    sigma <-> exp(0.5 posterior_log_variance_clipped) for t > 0, no noise at t = 0 (p_sample's nonzero_mask)."""
    tab = sch.ddpm_table()
    assert list(tab.timesteps) == list(range(999, -1, -1))
    c = tab.coef[::-1].astype(np.float64)          # row t
    np.testing.assert_allclose(1.0 / c[:, 1], ref["sqrt_recip_alphas_cumprod"], rtol=TABLE_RTOL)
    np.testing.assert_allclose(c[:, 0] / c[:, 1], ref["sqrt_recipm1_alphas_cumprod"], rtol=TABLE_RTOL)
    # c0 carries diffusers' fp32 "1 - abar_t / abar_{t-1}" (= beta_t >= 8.5e-4 by cancellation): 2^-24 / 8.5e-4 ~ 7e-5
    np.testing.assert_allclose(c[:, 2], ref["posterior_mean_coef1"], rtol=1e-4, atol=0)
    assert np.abs(c[:, 2] - ref["posterior_mean_coef1"]).max() < 2e-5
    np.testing.assert_allclose(c[1:, 3], ref["posterior_mean_coef2"][1:], rtol=TABLE_RTOL)
    assert c[0, 3] == 0.0 and ref["posterior_mean_coef2"][0] == 0.0
    np.testing.assert_allclose(c[1:, 5], np.exp(0.5 * ref["posterior_log_variance_clipped"][1:]), rtol=1e-4)
    np.testing.assert_allclose(c[1:, 5] ** 2, ref["posterior_variance"][1:], rtol=2e-4)
    assert c[0, 5] == 0.0 and np.all(c[:, 4] == 0) and np.all(c[:, 6] == 0)


def kernel_update(row, x, eps, z):
    """The update the HIP kernels apply per schedule row (include/amuse_hip.h amuse_schedule), in float32."""
    sb, sa, c0, cx, ce, sigma, clip, _ = (np.float32(v) for v in row)
    x0 = (x - sb * eps) / sa
    if clip > 0:
        x0 = np.clip(x0, -clip, clip)
    out = c0 * x0 + cx * x + ce * eps
    if sigma != 0:
        out = out + sigma * z
    return out.astype(np.float32)


def test_ddpm_single_step_known_answers(ref):
    x, eps, nz = ref["kat_x"], ref["kat_eps"], ref["kat_noise"]
    tab = sch.ddpm_table()
    o = orc.DDPM()
    for t in ref["kat_t"]:
        want = ref[f"kat_ddpm_t{t}/sample"]
        scale = np.abs(want).max()
        got_o = o.step(torch.from_numpy(eps), int(t), torch.from_numpy(x), torch.from_numpy(nz)).numpy()
        got_k = kernel_update(tab.coef[999 - t], x, eps, nz)
        # t = 999: x0 = (x - sb eps) / sa with sa = 0.068 amplifies the fp32 cancellation; relative to the output scale
        assert np.abs(got_o - want).max() / scale < 2e-5, t
        assert np.abs(got_k - want).max() / scale < 2e-5, t


def test_add_noise_matches_reference_q_sample(ref):
    ac = sch.alphas_cumprod()
    t = ref["q_sample_t"]
    got = np.sqrt(ac[t])[:, None] * ref["kat_x"] + np.sqrt(1 - ac[t])[:, None] * ref["kat_noise"]
    assert np.abs(got - ref["q_sample"]).max() < 1e-5
    # the oracle's diffusion_forward uses the same two factors
    oc = orc.SchedulerBase().alphas_cumprod
    tt = torch.from_numpy(t)
    got_o = oc[tt].sqrt()[:, None] * torch.from_numpy(ref["kat_x"]) + (1 - oc[tt]).sqrt()[:, None] * torch.from_numpy(ref["kat_noise"])
    assert np.abs(got_o.numpy() - ref["q_sample"]).max() < 1e-5


def ddpm_noise(ref, B=2):
    g = torch.Generator().manual_seed(int(ref["traj_noise_seed"]))
    return torch.stack([torch.randn(B, 128, generator=g) for _ in range(1000)])     # loop order t = 999 .. 0


def test_ddpm_1000_step_trajectory(ref):
    """Scheduler-only ancestral trajectory (eps_hat == a constant vector), 1000 steps: oracle and kernel formula."""
    e = ref["traj_eps_const"]
    nz = ddpm_noise(ref)
    tab, o = sch.ddpm_table(), orc.DDPM()
    xo = torch.from_numpy(ref["traj_x_T"].copy())
    xk = ref["traj_x_T"].copy()
    eb = np.broadcast_to(e, xk.shape)
    for i, t in enumerate(range(999, -1, -1)):
        xo = o.step(torch.from_numpy(eb.copy()), t, xo, nz[i])
        xk = kernel_update(tab.coef[i], xk, eb, nz[i].numpy())
        if t in (900, 500, 100, 0):
            want = ref[f"traj_ddpm_after_t{t}"]
            assert np.abs(xo.numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max()), t
            assert np.abs(xk - want).max() < 1e-4 * max(1.0, np.abs(want).max()), t


# ------------------------------------------------------------------------------------------------ A10
def test_ddim_timesteps(ref):
    assert list(sch.ddim_table().timesteps) == list(ref["ddim_timesteps"]) == list(range(981, 0, -20))
    assert orc.DDIM().timesteps == list(range(981, 0, -20))


@pytest.mark.parametrize("alpha_to_one,nsteps", [(True, 50), (False, 49)])
def test_ddim_unclipped_trajectory(ref, alpha_to_one, nsteps):
    """eta 0, no clipping: all 50 steps equal the reference tree's for set_alpha_to_one=True; for the reference's own
    set_alpha_to_one=False the first 49 do (the 50th differs only through alpha_bar_prev = abar[0], convention (i))."""
    eps, want = ref["ddim_eps_seq"], ref["ddim_traj_noclip"]
    o = orc.DDIM(set_alpha_to_one=alpha_to_one, clip_sample=False)
    tab = sch.ddim_table(set_alpha_to_one=alpha_to_one, clip_sample=False)
    xo, xk = torch.from_numpy(ref["ddim_x_T"].copy()), ref["ddim_x_T"].copy()
    for i, t in enumerate(o.timesteps[:nsteps]):
        xo = o.step(torch.from_numpy(eps[i]), t, xo)
        xk = kernel_update(tab.coef[i], xk, eps[i], None)
        tol = 2e-5 * max(1.0, np.abs(want[i]).max())
        assert np.abs(xo.numpy() - want[i]).max() < tol, i
        assert np.abs(xk - want[i]).max() < tol, i


def test_ddim_last_step_convention_is_the_only_difference(ref):
    """set_alpha_to_one False vs True: identical tables except the last row's sqrt(abar_prev) / sqrt(1 - abar_prev)."""
    a, b = sch.ddim_table(set_alpha_to_one=False), sch.ddim_table(set_alpha_to_one=True)
    assert np.array_equal(a.coef[:-1], b.coef[:-1])
    ac0 = ref["alphas_cumprod"][0]
    np.testing.assert_allclose(a.coef[-1, 2], np.sqrt(ac0), rtol=TABLE_RTOL)
    np.testing.assert_allclose(a.coef[-1, 4], np.sqrt(1 - ac0), rtol=1e-4)
    assert b.coef[-1, 2] == 1.0 and b.coef[-1, 4] == 0.0


def test_ddim_clipped_trajectory_rederived_eps(ref):
    """clip_denoised=True in the reference tree's ddim_sample == diffusers' clip_sample + use_clipped_model_output=True:
    pins the clamp of x0 at +-1 and the re-derivation; 50 steps with set_alpha_to_one=True."""
    eps, want = ref["ddim_eps_seq"], ref["ddim_traj_clip"]
    o = orc.DDIM(set_alpha_to_one=True, clip_sample=True, use_clipped_model_output=True)
    x = torch.from_numpy(ref["ddim_x_T"].copy())
    clipped_any = False
    for i, t in enumerate(o.timesteps):
        a = o.alphas_cumprod[t]
        clipped_any |= bool((((x - (1 - a) ** 0.5 * torch.from_numpy(eps[i])) / a ** 0.5).abs() > 1).any())
        x = o.step(torch.from_numpy(eps[i]), t, x)
        assert np.abs(x.numpy() - want[i]).max() < 2e-5 * max(1.0, np.abs(want[i]).max()), i
    assert clipped_any and np.abs(ref["ddim_traj_clip"] - ref["ddim_traj_noclip"]).max() > 1e-2


def test_ddim_shipped_convention_decomposes_into_the_pinned_pieces(ref):
    """The reference's configuration (clip_sample=True, use_clipped_model_output=False): x0 is the clamped value the clipped
    run pins, the direction term carries the raw eps_hat with the coefficient the unclipped run pins.  Checked row-wise:
    the shipped table's (sb, sa, c0, ce) equal the unclipped table's, and the clip range is 1."""
    a, b = sch.ddim_table(), sch.ddim_table(clip_sample=False)
    assert np.array_equal(a.coef[:, :6], b.coef[:, :6])
    assert np.all(a.coef[:, 6] == 1.0) and np.all(b.coef[:, 6] == 0.0)
    o = orc.DDIM()
    x, e = torch.from_numpy(ref["ddim_x_T"].copy()) * 3, torch.from_numpy(ref["ddim_eps_seq"][0])
    got_o = o.step(e, 981, x).numpy()
    got_k = kernel_update(a.coef[0], x.numpy(), e.numpy(), None)
    assert np.abs(got_o - got_k).max() < 1e-6


def ddim_eta_noise(ref, B=2):
    """What th.randn_like(x) returned inside the reference tree's ddim_sample at step i (oracle/gen_golden.py seeds torch's generator with seed0 + i in front of it)."""
    out = []
    for i in range(50):
        torch.manual_seed(int(ref["ddim_eta_seed0"]) + i)
        out.append(torch.randn(B, 128))
    return torch.stack(out)


@pytest.mark.parametrize("alpha_to_one,nsteps", [(True, 50), (False, 49)])
def test_ddim_eta_trajectory(ref, alpha_to_one, nsteps):
    """eta = 0.5 (stochastic DDIM; the reference forwards the configured eta to scheduler.step): sigma_t = eta sqrt((1 - abar_prev) / (1 - abar_t)) sqrt(1 - abar_t / abar_prev),
    direction coefficient sqrt(1 - abar_prev - sigma_t^2), + sigma_t z - oracle and the product's table against SpacedDiffusion.ddim_sample."""
    eta = float(ref["ddim_eta"])
    e = np.broadcast_to(ref["traj_eps_const"], (2, 128)).copy()
    nz = ddim_eta_noise(ref)
    want = ref["ddim_traj_const_eta"]
    o = orc.DDIM(set_alpha_to_one=alpha_to_one, clip_sample=False, eta=eta)
    tab = sch.ddim_table(set_alpha_to_one=alpha_to_one, clip_sample=False, eta=eta)
    assert tab.needs_noise()[:49].all()
    xo, xk = torch.from_numpy(ref["traj_x_T"].copy()), ref["traj_x_T"].copy()
    for i, t in enumerate(o.timesteps[:nsteps]):
        xo = o.step(torch.from_numpy(e), t, xo, nz[i])
        xk = kernel_update(tab.coef[i], xk, e, nz[i].numpy())
        tol = 2e-5 * max(1.0, np.abs(want[i]).max())
        assert np.abs(xo.numpy() - want[i]).max() < tol, i
        assert np.abs(xk - want[i]).max() < tol, i
    assert np.abs(want - ref["ddim_traj_const_noclip"]).max() > 1.0          # (the noise term is not a rounding effect)
