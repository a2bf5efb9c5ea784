"""Smallest possible launches of the sampling kernel first (1 clip, teacher-forced single step), then parity spot checks."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
g = np.load(REPO / "tests/golden/denoiser_steps.npz")
for prec in ("fp32", "bf16"):
    eps = eng.denoise_step(g["x_t"], 981, g["con"], g["emo"], g["sty"], prec)
    torch.cuda.synchronize()
    print(prec, "denoise_step max err vs golden", float(np.abs(eps.cpu().numpy() - g["eps_t981"]).max()), flush=True)
eng.set_schedule(sch.ddim_table(50))
tr = np.load(REPO / "tests/golden/ddim50_traj.npz")
lat = eng.sample(tr["con"], tr["emo"], tr["sty"], "fp32", x_init=tr["x_T"])
torch.cuda.synchronize()
print("ddim50 fp32 err", float(np.abs(lat.cpu().numpy() - tr["x_after_50"]).max()), flush=True)
eng.set_schedule(sch.ddpm_table(1000))
gen = torch.Generator().manual_seed(1)
for B in (1, 256, 768):
    c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
    for prec in ("bf16", "fp32"):
        eng.sample(c, e, s, prec, seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lat = eng.sample(c, e, s, prec, seed=1)
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} {prec}: {e0.elapsed_time(e1):.2f} ms per 1000 steps; finite={bool(torch.isfinite(lat).all())}", flush=True)
