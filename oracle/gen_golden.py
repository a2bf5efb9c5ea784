"""Generate tests/golden/*.npz from the REFERENCE's own modules (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
Needs /root/reference (read-only).  The reference never travels to the GPU box; only the vectors
written here do.  Inputs: the build's deterministic weights (amuse_amd.weights, seed 0) and seeded
torch.Generator draws.  Outputs are produced by:
  * models.latent_diffusion.denoiser.Denoiser          (denoiser.py:16-204)
  * models.latent_diffusion.vae.MotionPrior.decode      (vae.py:216-278)
  * models/diffusion/utils/rotation_conversions.py      (vendored pytorch3d snapshot)
imported through the package shim described in SURVEY.md section 8c (the reference's package
__init__ chain imports seaborn/timm/... which are not installed).
The DDIM-50 trajectory fixture drives the reference Denoiser with the oracle's restated scheduler
(diffusers is not installed) - it pins the network under iteration, not diffusers.
Two further fixtures pin the third-party arithmetic from material the reference tree itself holds
(`--pins-only` writes just these):
  * tests/golden/ref_poses.npz  - the `poses` arrays of the three committed sample outputs
    viz_dump/test/**/*_motion_smplx.npz: outputs of the DEPLOYED pytorch3d.matrix_to_axis_angle
    (infer_ldm.py:171-172), 49,500 joints, 10 of them with |aa| > pi.
  * tests/golden/sched_ref.npz  - posterior tables, single-step known answers, a DDPM-1000 and two DDIM-50
    scheduler-only trajectories computed by the reference's own GaussianDiffusion / SpacedDiffusion
    (models/diffusion/utils/mdm_gaussian_diffusion.py:198-278,343-366,528-549,895-940; mdm_respace.py:64-87) on the
    scaled_linear betas of configs/diff_latent_v2.json.
"""
import importlib.util
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(REPO))
sys.dont_write_bytecode = True

from amuse_amd import weights as wts  # noqa: E402
from oracle import amuse_oracle as orc  # noqa: E402


def _shim():
    for name, path in (("models", REF / "models"),
                       ("models.latent_diffusion", REF / "models/latent_diffusion"),
                       ("models.latent_diffusion.utils", REF / "models/latent_diffusion/utils")):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m
    from models.latent_diffusion.utils.position_encoding_layer import PositionalEncoding
    sys.modules["models.latent_diffusion.utils"].PositionalEncoding = PositionalEncoding
    from models.latent_diffusion.denoiser import Denoiser
    from models.latent_diffusion.vae import MotionPrior
    spec = importlib.util.spec_from_file_location("ref_rot", REF / "models/diffusion/utils/rotation_conversions.py")
    rot = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rot)
    return Denoiser, MotionPrior, rot


def build_variant(Denoiser, arch, diffusion_only):
    """Denoiser(arch_denoiser of configs/diff_latent_v2.json with `arch` / `diffusion_only` overridden), as
    PretrainedLPDM_v1.setup / LatentDiffusionModel.setup build it (infer_ldm.py:66-73, ldm.py:58-63)."""
    base = json.load(open(REF / "configs/base_new.json"))
    dcfg = dict(json.load(open(REF / "configs/diff_latent_v2.json"))["arch_denoiser"])
    dcfg["smplx_data"] = base["TRAIN_PARAM"]["latent_diffusion"]["smplx_data"]
    dcfg["smplx_rep"] = base["TRAIN_PARAM"]["latent_diffusion"]["smplx_rep"]
    dcfg["arch"], dcfg["diffusion_only"] = arch, diffusion_only
    return Denoiser(dcfg).eval()


POSE_ROWS = slice(0, 300, 6)   # frames of the pose-space outputs kept in the fixture (every output row depends on every input)


def gen_variants(out, Denoiser):
    """tests/golden/denoiser_variants.npz: teacher-forced eps_hat of the three Denoiser variants the shipped configuration
    does not reach - trans_dec (latent), diffusion_only + trans_enc (S = 304), diffusion_only + trans_dec - on the build's
    deterministic weights: t = 981 / 501 / 1, ragged lengths, token dropping, per-sample timesteps, a DDIM-50 trajectory
    (reference Denoiser + the restated scheduler).  Pose-space outputs are stored on every 6th frame."""
    spec, d = {}, {}
    g = torch.Generator().manual_seed(4202)
    B = 2
    con, emo, sty = (torch.randn(B, 256, generator=g) for _ in range(3))
    x_lat = torch.randn(B, 1, 128, generator=g)
    x_pose = torch.randn(B, 300, 333, generator=g).half().float()   # stored as float16: round first
    d.update(con=con.numpy(), emo=emo.numpy(), sty=sty.numpy(), x_lat=x_lat[:, 0].numpy(), x_pose=x_pose.numpy().astype(np.float16))
    sched = orc.DDIM()
    for arch, pose in (("trans_dec", False), ("trans_enc", True), ("trans_dec", True)):
        tag = f"{arch}{'_pose' if pose else ''}"
        den = build_variant(Denoiser, arch, pose)
        spec[tag] = {k: list(v.shape) for k, v in den.state_dict().items()}
        load_weights(den, wts.make_denoiser_weights(0, arch, pose))
        x = x_pose if pose else x_lat
        cut = (lambda e: e[:, POSE_ROWS].numpy()) if pose else (lambda e: e[:, 0].numpy())
        kw = dict(con_hidden=con[:, None], emo_hidden=emo[:, None], sty_hidden=sty[:, None], lengths=[300] * B)
        taps, hooks = {}, []
        if arch == "trans_dec":
            def tap(name):
                def fn(_m, _i, o):
                    taps[name] = o.permute(1, 0, 2).clone().numpy()
                return fn
            hooks = [den.query_pos.register_forward_hook(tap("tokens")), den.mem_pos.register_forward_hook(tap("memory")),
                     den.decoder.layers[0].register_forward_hook(tap("decoder.layers.0")),
                     den.decoder.layers[8].register_forward_hook(tap("decoder.layers.8"))]
        for t in (981, 501, 1):
            d[f"{tag}/eps_t{t}"] = cut(den(sample=x, timestep=torch.tensor(t), **kw)[0])
            if t == 981:
                for k, v in taps.items():
                    d[f"{tag}/tap981/{k}"] = v[:, POSE_ROWS] if (pose and k != "memory") else v
        for h in hooks:
            h.remove()
        d[f"{tag}/eps_t501_noemo"] = cut(den(sample=x, timestep=torch.tensor(501), **dict(kw, emo_hidden=None))[0])
        d[f"{tag}/eps_t501_consolo"] = cut(den(sample=x, timestep=torch.tensor(501), **dict(kw, emo_hidden=None, sty_hidden=None))[0])
        ts = torch.tensor([7, 640])
        d[f"{tag}/eps_batch_t"] = cut(den(sample=x, timestep=ts, **kw)[0])
        if pose:   # sample[~mask.T] = 0 (denoiser.py:187,199); the padded frames stay attended keys
            d[f"{tag}/eps_t501_ragged"] = cut(den(sample=x, timestep=torch.tensor(501), **dict(kw, lengths=[300, 173]))[0])
        # DDIM-50: reference Denoiser + restated scheduler, explicit x_T
        xx = x[:, 0].clone() if not pose else x.clone()
        for i, t in enumerate(sched.timesteps):
            eps = den(sample=xx[:, None] if not pose else xx, timestep=torch.tensor(t), **kw)[0]
            xx = sched.step(eps[:, 0] if not pose else eps, t, xx)
            if (i + 1) in (10, 50):
                d[f"{tag}/x_after_{i + 1}"] = xx[:, POSE_ROWS].numpy().copy() if pose else xx.numpy().copy()
    d["timesteps_batch"] = np.array([7, 640])
    d["lengths_ragged"] = np.array([300, 173])
    np.savez_compressed(out / "denoiser_variants.npz", **d)
    json.dump(spec, open(out / "state_dict_spec_variants.json", "w"), indent=0)


def _shim_mdm():
    """models.diffusion.utils.{mdm_gaussian_diffusion, mdm_respace} through the same empty-package shim
    (the package __init__ chain imports seaborn / timm; these two files need numpy, torch and einops only)."""
    for name, path in (("models", REF / "models"), ("models.diffusion", REF / "models/diffusion"),
                       ("models.diffusion.utils", REF / "models/diffusion/utils")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m
    from models.diffusion.utils import mdm_gaussian_diffusion as gdm
    from models.diffusion.utils import mdm_respace as rsp
    return gdm, rsp


def gen_ref_poses(out):
    """tests/golden/ref_poses.npz: data the reference's tests/ samples hold - the axis-angle `poses` the deployed
    pytorch3d wrote (infer_ldm.py:171-172 -> trainer.py:524-526 -> visualizer.py:344-364)."""
    d = {}
    for p in sorted((REF / "viz_dump/test").rglob("*_motion_smplx.npz")):
        z_ = np.load(p, allow_pickle=True)
        key = p.name.split("_motion_smplx")[0]
        assert z_["poses"].dtype == np.float32 and z_["poses"].shape == (300, 55, 3)
        d[key] = z_["poses"]
    assert len(d) == 3
    np.savez_compressed(out / "ref_poses.npz", **d)


def gen_sched_ref(out):
    """tests/golden/sched_ref.npz from the reference tree's own Gaussian diffusion code.

    GaussianDiffusion(betas = scaled_linear fp64, EPSILON, FIXED_SMALL): its posterior (mdm_gaussian_diffusion.py:343-366)
    IS diffusers' DDPMScheduler.step with variance_type fixed_small at 1000 inference steps, its q_sample (:323-341) IS
    add_noise.  p_sample / p_mean_variance were rewritten upstream around dicts of pose streams and no longer run on a
    plain tensor, so a step is composed here from the member functions they call - _predict_xstart_from_eps (:528-533),
    q_posterior_mean_variance (:343-366) - and p_sample's own last line (:690): mean + (t != 0) exp(0.5 logvar) noise.
    DDIM: SpacedDiffusion on {1, 21, ..., 981} (mdm_respace.py:64-87) + ddim_sample (:895-940, eta = 0), with
    p_mean_variance overridden to the EPSILON branch of the original (:506-511) because the upstream body is the
    dict version.  Residual differences to diffusers 0.17.1, stated: (i) the last step's alpha_bar_prev is 1.0 here
    (diffusers: set_alpha_to_one=True; the reference passes False -> abar[0]), (ii) with clip_denoised=True this code
    re-derives eps from the clipped x0 (diffusers: use_clipped_model_output=True; the reference leaves it False)."""
    gdm, rsp = _shim_mdm()
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2      # diff_latent_v2.json:48-66
    kw = dict(betas=betas, model_mean_type=gdm.ModelMeanType.EPSILON, model_var_type=gdm.ModelVarType.FIXED_SMALL,
              loss_type=gdm.LossType.MSE)
    gd = gdm.GaussianDiffusion(**kw)
    d = {"betas": betas, "alphas_cumprod": gd.alphas_cumprod, "posterior_mean_coef1": gd.posterior_mean_coef1,
         "posterior_mean_coef2": gd.posterior_mean_coef2, "posterior_variance": gd.posterior_variance,
         "posterior_log_variance_clipped": gd.posterior_log_variance_clipped,
         "sqrt_recip_alphas_cumprod": gd.sqrt_recip_alphas_cumprod, "sqrt_recipm1_alphas_cumprod": gd.sqrt_recipm1_alphas_cumprod}

    def ddpm_step(x, t, eps, noise):
        tt = torch.full((x.shape[0],), t, dtype=torch.long)
        x0 = gd._predict_xstart_from_eps(x, tt, eps)
        mean, _, logvar = gd.q_posterior_mean_variance(x0, x, tt)
        return x0, mean, mean + float(t != 0) * torch.exp(0.5 * logvar) * noise

    g = torch.Generator().manual_seed(1105)
    B = 4
    x, eps, noise = (torch.randn(B, 128, generator=g) for _ in range(3))
    d.update(kat_x=x.numpy(), kat_eps=eps.numpy(), kat_noise=noise.numpy(), kat_t=np.array([999, 500, 37, 1, 0]))
    for t in (999, 500, 37, 1, 0):
        x0, mean, smp = ddpm_step(x, t, eps, noise)
        d[f"kat_ddpm_t{t}/x0"], d[f"kat_ddpm_t{t}/mean"], d[f"kat_ddpm_t{t}/sample"] = x0.numpy(), mean.numpy(), smp.numpy()
    # add_noise (the only DDPMScheduler call the reference makes, ldm.py:84) == q_sample, per-sample timesteps
    ts = torch.tensor([7, 640, 999, 0])
    d["q_sample_t"], d["q_sample"] = ts.numpy(), gd.q_sample(x, ts, noise=noise).numpy()

    # DDPM-1000 scheduler-only trajectory: eps_hat == a constant vector (what a Denoiser whose final LayerNorm has weight 0
    # and bias e produces), ancestral noise from torch.Generator(seed) drawn step by step in loop order -> stored checkpoints
    B2 = 2
    gt = torch.Generator().manual_seed(77)
    e_const = torch.randn(128, generator=gt)
    xt = torch.randn(B2, 128, generator=gt)
    d.update(traj_eps_const=e_const.numpy(), traj_x_T=xt.numpy().copy(), traj_noise_seed=np.array(78))
    gn = torch.Generator().manual_seed(78)
    for t in range(999, -1, -1):
        nz = torch.randn(B2, 128, generator=gn)
        xt = ddpm_step(xt, t, e_const[None].expand(B2, -1), nz)[2]
        if t in (900, 500, 100, 0):
            d[f"traj_ddpm_after_t{t}"] = xt.numpy().copy()

    class EpsOnly(rsp.SpacedDiffusion):
        def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
            x0 = self._predict_xstart_from_eps(x_t=x, t=t, eps=model(x, t))
            return {"pred_xstart": x0.clamp(-1, 1) if clip_denoised else x0}

    sd = EpsOnly(use_timesteps=set(range(1, 1000, 20)), **kw)
    assert sd.timestep_map == list(range(1, 1000, 20))
    ge = torch.Generator().manual_seed(79)
    eps_seq = torch.randn(50, B2, 128, generator=ge)      # a different eps_hat per step, same for both runs
    x_T = torch.randn(B2, 128, generator=ge)
    d.update(ddim_eps_seq=eps_seq.numpy(), ddim_x_T=x_T.numpy(), ddim_timesteps=np.array(sd.timestep_map[::-1]))
    for clip in (False, True):
        xx, tr = x_T.clone(), []
        for i, idx in enumerate(range(49, -1, -1)):
            o = sd.ddim_sample(lambda _x, _t: eps_seq[i], xx, torch.full((B2,), idx, dtype=torch.long),
                               clip_denoised=clip, eta=0.0)
            xx = o["sample"]
            tr.append(xx.numpy().copy())
        d[f"ddim_traj_{'clip' if clip else 'noclip'}"] = np.stack(tr)
    # the same with eps_hat == the constant vector of the DDPM trajectory above (what the GPU test can drive through the kernels)
    xx, tr = torch.from_numpy(d["traj_x_T"]).clone(), []
    for idx in range(49, -1, -1):
        xx = sd.ddim_sample(lambda _x, _t: e_const[None].expand(B2, -1), xx, torch.full((B2,), idx, dtype=torch.long),
                            clip_denoised=False, eta=0.0)["sample"]
        tr.append(xx.numpy().copy())
    d["ddim_traj_const_noclip"] = np.stack(tr)
    # eta > 0 (the reference forwards configs/diff_latent_v2.json "eta" to scheduler.step, infer_ldm.py:142-147,160-161): ddim_sample draws its own noise with
    # th.randn_like(x) - made reproducible by seeding torch's generator in front of every step (the test re-draws the same (2, 128) normals)
    d["ddim_eta"] = np.array(0.5)
    d["ddim_eta_seed0"] = np.array(4100)
    xx, tr = torch.from_numpy(d["traj_x_T"]).clone(), []
    for i, idx in enumerate(range(49, -1, -1)):
        torch.manual_seed(4100 + i)
        xx = sd.ddim_sample(lambda _x, _t: e_const[None].expand(B2, -1), xx, torch.full((B2,), idx, dtype=torch.long),
                            clip_denoised=False, eta=0.5)["sample"]
        tr.append(xx.numpy().copy())
    d["ddim_traj_const_eta"] = np.stack(tr)
    np.savez_compressed(out / "sched_ref.npz", **d)


def build_reference():
    Denoiser, MotionPrior, rot = _shim()
    base = json.load(open(REF / "configs/base_new.json"))
    ldm_cfg = json.load(open(REF / "configs/diff_latent_v2.json"))
    prior_cfg = json.load(open(REF / "configs/prior_emotional_fing.json"))
    # infer_gesture.yaml overrides (scripts/overrides/infer_gesture.yaml): smplx_rep 6D etc. already the defaults
    dcfg = dict(ldm_cfg["arch_denoiser"])
    dcfg["smplx_data"] = base["TRAIN_PARAM"]["latent_diffusion"]["smplx_data"]  # infer_ldm.py:69
    dcfg["smplx_rep"] = base["TRAIN_PARAM"]["latent_diffusion"]["smplx_rep"]    # infer_ldm.py:72
    den = Denoiser(dcfg).eval()
    prior = MotionPrior()
    prior.setup(None, base, prior_cfg=prior_cfg)
    prior.eval()
    return den, prior, rot


def load_weights(module, w):
    sd = module.state_dict()
    assert list(sd.keys()) == list(w.keys()), "state-dict key list/order mismatch vs amuse_amd.weights spec"
    for k, v in w.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    for p in module.parameters():
        p.requires_grad = False


def gen_wellcond(out):
    """tests/golden/wellcond.npz: the whole path of infer_ldm.py:130-178 on a SECOND weight draw (seed 1) whose decoder emits
    well-conditioned 6D rotations (weights.make_wellcond_prior_weights), produced by the reference's own modules:
      reference Denoiser x DDIM-50 (the restated scheduler of the shipped configuration: clip_sample, eta 0, steps_offset 1)
      -> reference MotionPrior.decode -> vendored rotation_6d_to_matrix -> the p3d matrix_to_axis_angle that tests/golden/ref_poses.npz pins.
    Two jobs: `full` (B = 4: content, emotion, style) and `noemo` (B = 2: z_emo = None, the S = 4 token-dropping path of
    denoiser.py:159-171).  Also the per-kernel goldens of that draw: teacher-forced eps_hat at t = 981 / 501 / 1,
    MotionPrior.decode with ragged lengths, MotionPrior.encode (mu / std, full + ragged)."""
    den, prior, rot = build_reference()
    wd, wp = wts.make_denoiser_weights(1), wts.make_wellcond_prior_weights(1)
    load_weights(den, wd)
    load_weights(prior, wp)
    g = torch.Generator().manual_seed(6001)
    B = 4
    con, emo, sty = (torch.randn(B, 256, generator=g) for _ in range(3))
    x_T = torch.randn(B, 128, generator=g)
    d = {"con": con.numpy(), "emo": emo.numpy(), "sty": sty.numpy(), "x_T": x_T.numpy()}
    for t in (981, 501, 1):
        d[f"eps_t{t}"] = den(sample=x_T[:, None], timestep=torch.tensor(t), con_hidden=con[:, None], emo_hidden=emo[:, None],
                             sty_hidden=sty[:, None], lengths=[300] * B)[0][:, 0].numpy()
    sched = orc.DDIM()

    def job(tag, sl, use_emo):
        n = sl.stop - sl.start
        x = x_T[sl].clone() * sched.init_noise_sigma
        for i, t in enumerate(sched.timesteps):
            eps = den(sample=x[:, None], timestep=torch.tensor(t), con_hidden=con[sl, None],
                      emo_hidden=emo[sl, None] if use_emo else None, sty_hidden=sty[sl, None], lengths=[300] * n)[0][:, 0]
            x = sched.step(eps, t, x)
            if i + 1 == 10:
                d[f"{tag}/x_after_10"] = x.numpy().copy()
        feats = prior.decode(x[None], [300] * n)                       # PretrainedVAE.get_motion (infer_pretrained_vae.py)
        rot6d = feats[..., :-3].reshape(n, 300, 55, 6)                 # infer_ldm.py:168-170
        mat = rot.rotation_6d_to_matrix(rot6d)
        poses = orc.matrix_to_axis_angle(mat, "p3d")                   # the deployed pytorch3d's function, pinned by ref_poses.npz
        # conditioning facts of the fixture: the Gram-Schmidt pivots (|a1|, |a2 - (b1.a2) b1|) and the largest rotation angle
        a1, a2 = rot6d[..., :3], rot6d[..., 3:]
        b1 = torch.nn.functional.normalize(a1, dim=-1)
        piv = torch.minimum(a1.norm(dim=-1), (a2 - (b1 * a2).sum(-1, keepdim=True) * b1).norm(dim=-1))
        print(f"wellcond/{tag}: min pivot {float(piv.min()):.3f}, max |aa| {float(poses.norm(dim=-1).max()):.3f} rad, "
              f"latent |x| max {float(x.abs().max()):.3f}")
        # the candidate selection of matrix_to_quaternion is discontinuous where two |q| candidates tie (q <-> -q: the same rotation
        # written with |aa| on the other side of pi): the fixture must sit away from those ties, by far more than the 1e-4 bar
        tr = torch.stack([1 + mat[..., 0, 0] + mat[..., 1, 1] + mat[..., 2, 2], 1 + mat[..., 0, 0] - mat[..., 1, 1] - mat[..., 2, 2],
                          1 - mat[..., 0, 0] + mat[..., 1, 1] - mat[..., 2, 2], 1 - mat[..., 0, 0] - mat[..., 1, 1] + mat[..., 2, 2]], -1)
        top = tr.clamp(min=0).sqrt().topk(2, dim=-1).values
        tie = float((top[..., 0] - top[..., 1]).min())
        print(f"wellcond/{tag}: closest |q| tie {tie:.2e}, joints with |aa| > pi: {int((poses.norm(dim=-1) > np.pi).sum())}")
        assert float(piv.min()) >= 0.5 and tie > 2e-3, "not well-conditioned"
        d[f"{tag}/tie_margin"] = np.array(tie)
        d[f"{tag}/latents"], d[f"{tag}/feats"], d[f"{tag}/poses"] = x.numpy().copy(), feats.numpy(), poses.numpy()
        d[f"{tag}/min_pivot"] = np.array(float(piv.min()))
        return x

    lat = job("full", slice(0, 4), True)
    job("noemo", slice(0, 2), False)
    # per-kernel goldens of the second draw: ragged decode, encode
    d["feats_ragged"] = prior.decode(lat[None, :2], [300, 173]).numpy()
    d["lengths_ragged"] = np.array([300, 173])
    fe = (0.5 * torch.randn(2, 300, 333, generator=g)).half().float()
    _, dist = prior.encode(fe, [300, 300])
    _, dist_r = prior.encode(fe, [300, 211])
    d.update(enc_feats=fe.numpy().astype(np.float16), mu=dist.loc[0].numpy(), std=dist.scale[0].numpy(),
             mu_ragged=dist_r.loc[0].numpy(), std_ragged=dist_r.scale[0].numpy(), enc_lengths_ragged=np.array([300, 211]))
    np.savez_compressed(out / "wellcond.npz", **d)


def main():
    out = REPO / "tests/golden"
    out.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    if "--pins-only" in sys.argv:       # only tests/golden/ref_poses.npz + sched_ref.npz
        gen_ref_poses(out)
        gen_sched_ref(out)
        return
    if "--wellcond" in sys.argv:        # only tests/golden/wellcond.npz (second weight draw, well-conditioned decoder)
        gen_wellcond(out)
        return
    if "--variants-only" in sys.argv:   # only tests/golden/denoiser_variants.npz + state_dict_spec_variants.json
        gen_variants(out, _shim()[0])
        return
    den, prior, rot = build_reference()
    gen_variants(out, type(den))

    # ---- state-dict spec straight from the reference modules
    spec = {"denoiser": {k: list(v.shape) for k, v in den.state_dict().items()},
            "prior": {k: list(v.shape) for k, v in prior.state_dict().items()}}
    json.dump(spec, open(out / "state_dict_spec.json", "w"), indent=0)
    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    load_weights(den, wd)
    load_weights(prior, wp)
    print("denoiser params", sum(p.numel() for p in den.parameters()),
          "prior params", sum(p.numel() for p in prior.parameters()))

    g = torch.Generator().manual_seed(2024)
    B = 3
    con = torch.randn(B, 256, generator=g)
    emo = torch.randn(B, 256, generator=g)
    sty = torch.randn(B, 256, generator=g)
    x_t = torch.randn(B, 1, 128, generator=g)
    lengths = [300] * B

    # ---- Denoiser: teacher-forced eps_hat at three timesteps, module-level taps at t = 981
    taps = {}
    hooks = []
    def tap(name, seq_first=True):
        def fn(_m, _i, o):
            taps[name] = (o.permute(1, 0, 2) if (seq_first and o.dim() == 3) else o).clone().numpy()
        return fn
    hooks.append(den.time_embedding.register_forward_hook(tap("time_embedding", False)))
    for n in ("con", "emo", "sty"):
        hooks.append(getattr(den, f"emb_proj_{n}").register_forward_hook(tap(f"emb_proj_{n}")))
    hooks.append(den.query_pos.register_forward_hook(tap("tokens")))
    hooks.append(den.encoder.input_blocks[0].register_forward_hook(tap("encoder.input_blocks.0")))
    hooks.append(den.encoder.middle_block.register_forward_hook(tap("encoder.middle_block")))
    hooks.append(den.encoder.output_blocks[3].register_forward_hook(tap("encoder.output_blocks.3")))
    d = {"con": con.numpy(), "emo": emo.numpy(), "sty": sty.numpy(), "x_t": x_t[:, 0].numpy()}
    for t in (981, 501, 1):
        eps = den(sample=x_t, timestep=torch.tensor(t), con_hidden=con[:, None], emo_hidden=emo[:, None],
                  sty_hidden=sty[:, None], lengths=lengths)[0]
        d[f"eps_t{t}"] = eps[:, 0].numpy()
        if t == 981:
            for k, v in taps.items():
                d[f"tap981/{k}"] = v
    for h in hooks:
        h.remove()
    # token-dropping variants (emo / sty None -> S = 4 / 3; denoiser.py:159-171)
    d["eps_t501_noemo"] = den(sample=x_t, timestep=torch.tensor(501), con_hidden=con[:, None], emo_hidden=None,
                              sty_hidden=sty[:, None], lengths=lengths)[0][:, 0].numpy()
    d["eps_t501_consolo"] = den(sample=x_t, timestep=torch.tensor(501), con_hidden=con[:, None], emo_hidden=None,
                                sty_hidden=None, lengths=lengths)[0][:, 0].numpy()
    # per-sample timesteps, as LatentDiffusionModel.diffusion_forward calls the denoiser (ldm.py:75-97)
    ts_batch = torch.tensor([7, 640, 999])
    d["timesteps_batch"] = ts_batch.numpy()
    d["eps_batch_t"] = den(sample=x_t, timestep=ts_batch, con_hidden=con[:, None], emo_hidden=emo[:, None],
                           sty_hidden=sty[:, None], lengths=lengths)[0][:, 0].numpy()
    np.savez_compressed(out / "denoiser_steps.npz", **d)

    # ---- DDIM-50 trajectory: reference Denoiser + restated scheduler, explicit x_T, B = 2
    sched = orc.DDIM()
    assert sched.timesteps[0] == 981 and sched.timesteps[-1] == 1 and len(sched.timesteps) == 50
    B2 = 2
    x = torch.randn(B2, 128, generator=g)
    tr = {"x_T": x.numpy().copy(), "con": con[:B2].numpy(), "emo": emo[:B2].numpy(), "sty": sty[:B2].numpy()}
    for i, t in enumerate(sched.timesteps):
        eps = den(sample=x[:, None], timestep=torch.tensor(t), con_hidden=con[:B2, None], emo_hidden=emo[:B2, None],
                  sty_hidden=sty[:B2, None], lengths=[300] * B2)[0][:, 0]
        x = sched.step(eps, t, x)
        if (i + 1) % 10 == 0:
            tr[f"x_after_{i + 1}"] = x.numpy().copy()
    np.savez_compressed(out / "ddim50_traj.npz", **tr)

    # ---- MotionPrior.decode on the DDIM result and on a fresh latent
    z = torch.stack([x[0], x[1], torch.randn(128, generator=g)])  # (3,128)
    dtaps = {}
    hk = [prior.decoder.input_blocks[0].register_forward_hook(
              lambda _m, _i, o: dtaps.__setitem__("decoder.input_blocks.0", o.permute(1, 0, 2).clone().numpy())),
          prior.decoder.output_blocks[3].register_forward_hook(
              lambda _m, _i, o: dtaps.__setitem__("decoder.output_blocks.3", o.permute(1, 0, 2).clone().numpy()))]
    feats = prior.decode(z[None], [300] * 3)  # (3,300,333)
    for h in hk:
        h.remove()
    feats_ragged = prior.decode(z[None, :2], [300, 173])  # padded keys masked, padded frames zeroed
    np.savez_compressed(out / "vae_decode.npz", z=z.numpy(), feats=feats.numpy(),
                        feats_ragged=feats_ragged.numpy(), lengths_ragged=np.array([300, 173]),
                        **{f"tap/{k}": v[:, ::25].copy() for k, v in dtaps.items()})

    # ---- MotionPrior.encode: distribution parameters for full-length and ragged inputs (vae.py:154-214)
    fe = (0.5 * torch.randn(2, 300, 333, generator=g)).half().float()   # stored as float16: round first
    _, dist = prior.encode(fe, [300, 300])
    _, dist_r = prior.encode(fe, [300, 211])
    np.savez_compressed(out / "vae_encode.npz", feats=fe.numpy().astype(np.float16), mu=dist.loc[0].numpy(),
                        std=dist.scale[0].numpy(), mu_ragged=dist_r.loc[0].numpy(), std_ragged=dist_r.scale[0].numpy(),
                        lengths_ragged=np.array([300, 211]))

    # ---- rotation conversions from the vendored pytorch3d snapshot
    d6 = torch.randn(2000, 6, generator=g)
    d6[:50] *= 1e-3
    d6[50:60, 3:] = d6[50:60, :3] * 2.0 + 1e-4 * torch.randn(10, 3, generator=g)  # nearly parallel a1,a2
    mat = rot.rotation_6d_to_matrix(d6)
    # near-pi rotations: axis-angle with |aa| in [pi-1e-3, pi]
    ax = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=-1)
    ang = torch.cat([torch.full((100,), np.pi) - 1e-3 * torch.rand(100, generator=g), 3.0 * torch.rand(100, generator=g)])
    mat_pi = rot.axis_angle_to_matrix(ax * ang[:, None])
    allm = torch.cat([mat, mat_pi])
    np.savez_compressed(out / "rotation.npz", d6=d6.numpy(), mat=mat.numpy(), mat_extra=mat_pi.numpy(),
                        quat_legacy=rot.matrix_to_quaternion(allm).numpy(),
                        aa_legacy=rot.matrix_to_axis_angle(allm).numpy(),
                        aa2mat=rot.axis_angle_to_matrix(rot.matrix_to_axis_angle(allm)).numpy())

    # ---- layout facts of the committed sample outputs (viz_dump/test/**/*_motion_smplx.npz)
    lay = {}
    for p in sorted((REF / "viz_dump/test").rglob("*_motion_smplx.npz")):
        z_ = np.load(p, allow_pickle=True)
        poses = z_["poses"]
        lay[str(p.relative_to(REF))] = {
            "fields": {k: [str(z_[k].dtype), list(z_[k].shape)] for k in z_.files},
            "lower_body_constant": bool(np.all(poses[:, orc.LOWER_BODY] == poses[0, orc.LOWER_BODY])),
            "trans_zero": bool(np.all(z_["trans"] == 0)),
            "max_aa_norm": float(np.linalg.norm(poses, axis=-1).max()),
            "mocap_frame_rate": float(z_["mocap_frame_rate"]),
        }
    json.dump(lay, open(out / "npz_layout.json", "w"), indent=1)
    # ---- gender / betas the reference's own writer put into those files (visualizer.py:357-362 <- ldm_evals.py:67-71)
    bet = {}
    for p in sorted((REF / "viz_dump/test").rglob("*_motion_smplx.npz")):
        z_ = np.load(p, allow_pickle=True)
        actor = p.name.split("_")[0]
        assert actor not in bet or np.array_equal(bet[actor], z_["betas"])
        bet[actor] = z_["betas"]
        bet[actor + "_gender"] = z_["gender"]
    np.savez_compressed(out / "sample_npz_betas.npz", **bet)
    gen_ref_poses(out)
    gen_sched_ref(out)
    gen_wellcond(out)
    for f in sorted(out.iterdir()):
        print(f.name, os.path.getsize(f))


if __name__ == "__main__":
    main()
