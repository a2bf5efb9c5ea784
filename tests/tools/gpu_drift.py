"""bf16 throughput mode vs fp32 parity mode on identical inputs + noise (the analogue of BASELINE.md section 2's
fp32-vs-bf16-autocast drift rows), and fp32 DDPM-1000 vs the CPU oracle for one clip."""
import sys, json
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
from oracle import amuse_oracle as orc
wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
eng = HipEngine(wd, wp)
g = torch.Generator().manual_seed(2024)
B = 64
c, e, s, x = (torch.randn(B, n, generator=g) for n in (256, 256, 256, 128))
res = {}
for name, tab in (("ddim50", sch.ddim_table()), ("ddpm1000", sch.ddpm_table())):
    eng.set_schedule(tab)
    nz = torch.randn(tab.n_steps, B, 128, generator=g) if name == "ddpm1000" else None
    a = eng.sample(c, e, s, "fp32", x_init=x, step_noise=nz).cpu()
    b = eng.sample(c, e, s, "bf16", x_init=x, step_noise=nz).cpu()
    res[name] = {"latent_rms": float(a.pow(2).mean().sqrt()), "bf16_vs_fp32_max": float((a - b).abs().max()),
                 "bf16_vs_fp32_rms": float((a - b).pow(2).mean().sqrt())}
    pa = eng.vae_decode(a, None, "fp32")["poses"].cpu(); pb = eng.vae_decode(b, None, "bf16")["poses"].cpu()
    Ra, Rb = orc.axis_angle_to_matrix(pa.double()), orc.axis_angle_to_matrix(pb.double())
    ang = torch.acos(((Ra.transpose(-1, -2) @ Rb).diagonal(dim1=-2, dim2=-1).sum(-1) - 1).div(2).clamp(-1, 1))
    res[name]["pose_geodesic_deg_median"] = float(ang.median() * 180 / np.pi)
    res[name]["pose_geodesic_deg_p99"] = float(ang.flatten().kthvalue(int(ang.numel() * 0.99)).values * 180 / np.pi)
    print(name, res[name])
# fp32 DDPM-1000, one clip, explicit noise, vs the oracle
Wd = orc.to_torch(wd)
tab = sch.ddpm_table(); eng.set_schedule(tab)
nz = torch.randn(1000, 1, 128, generator=g)
lat = eng.sample(c[:1], e[:1], s[:1], "fp32", x_init=x[:1], step_noise=nz).cpu()
ref = orc.sample_latents(Wd, orc.DDPM(), c[:1], e[:1], s[:1], x[:1], nz)
res["ddpm1000_fp32_vs_oracle"] = {"max_abs": float((lat - ref).abs().max()), "ref_rms": float(ref.pow(2).mean().sqrt())}
print(res["ddpm1000_fp32_vs_oracle"])
json.dump(res, open(REPO / "gpurun_out/drift.json", "w"), indent=1)
