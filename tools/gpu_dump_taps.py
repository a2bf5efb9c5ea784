import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
eng = HipEngine(wd, wp)
g = np.load(REPO / "tests/golden/denoiser_steps.npz")
out = {}
for prec in ("fp32", "bf16"):
    eps, tap = eng.denoise_step(g["x_t"], 981, g["con"], g["emo"], g["sty"], prec, taps=True)
    out[f"eps_{prec}"] = eps.cpu().numpy(); out[f"tap_{prec}"] = tap.cpu().numpy()
vg = np.load(REPO / "tests/golden/vae_decode.npz")
for prec in ("fp32", "bf16"):
    o = eng.vae_decode(vg["z"][:1], None, prec, return_feats=True)
    out[f"feats_{prec}"] = o["feats"].cpu().numpy()[:, ::10]
np.savez_compressed(REPO / "gpurun_out/taps.npz", **out)
print("dumped")
