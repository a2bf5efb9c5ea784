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


def main():
    out = REPO / "tests/golden"
    out.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
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
    for f in sorted(out.iterdir()):
        print(f.name, os.path.getsize(f))


if __name__ == "__main__":
    main()
