import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
from oracle import amuse_oracle as orc
wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
Wd, Wp = orc.to_torch(wd), orc.to_torch(wp)
eng = HipEngine(wd, wp)

def masks(feats):
    d6 = feats[..., :-3].reshape(*feats.shape[:-1], 55, 6).double()
    a1, a2 = d6[..., :3], d6[..., 3:]
    n1 = a1.norm(dim=-1); b1 = a1 / n1[..., None]
    n2 = (a2 - (b1 * a2).sum(-1, keepdim=True) * b1).norm(dim=-1)
    m = orc.rotation_6d_to_matrix(d6)
    qa = torch.stack([1 + m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2], 1 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2],
                      1 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2], 1 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2]], -1).clamp(min=0).sqrt()
    top = qa.topk(2, dim=-1).values
    return n1, n2, top[..., 0] - top[..., 1]

g = np.load(REPO / "tests/golden/vae_decode.npz")
o = eng.vae_decode(g["z"], None, "fp32", return_feats=True)
feats = o["feats"].cpu()
n1, n2, margin = masks(feats)
ref, _ = orc.feats_to_smplx(feats, "p3d")
d = torch.linalg.vector_norm(o["poses"].cpu() - ref, dim=-1)
print("same-feats fp32 oracle: max", float(d.max()), "p99.9", float(d.flatten().kthvalue(int(d.numel()*0.999)).values))
for thr in (0.05, 0.1, 0.3):
    wc = (n1 > thr) & (n2 > thr) & (margin > 1e-3)
    print(" thr", thr, "frac", float(wc.float().mean()), "max d", float(d[wc].max()))
print("min n1", float(n1.min()), "min n2", float(n2.min()))

gen = torch.Generator().manual_seed(2024)
c, e, s, x = (torch.randn(1, n, generator=gen) for n in (256, 256, 256, 128))
eng.set_schedule(sch.ddim_table())
out = eng.diffusion_backward(c, e, s, "fp32", x_init=x)
r = orc.diffusion_backward(Wd, Wp, orc.DDIM(), c, e, s, x)
print("lat err", float((out["latents"].cpu() - r["latents"]).abs().max()), "feats? trans err", float((out["trans"].cpu() - r["trans"]).abs().max()))
n1, n2, margin = masks(r["feats"])
d = torch.linalg.vector_norm(out["poses"].cpu() - r["poses"], dim=-1)
print("e2e: max", float(d.max()), "median", float(d.median()), "p99", float(d.flatten().kthvalue(int(d.numel()*0.99)).values))
for thr in (0.05, 0.1, 0.3, 0.5):
    wc = (n1 > thr) & (n2 > thr) & (margin > 1e-3)
    print(" thr", thr, "frac", float(wc.float().mean()), "max d", float(d[wc].max()))
R1, R2 = orc.axis_angle_to_matrix(out["poses"].cpu().double()), orc.axis_angle_to_matrix(r["poses"].double())
dr = (R1 - R2).abs().amax(dim=(-1, -2))
print("rot err max", float(dr.max()))
for thr in (0.05, 0.1, 0.3):
    wc = (n1 > thr) & (n2 > thr)
    print(" thr", thr, "rot max", float(dr[wc].max()))
