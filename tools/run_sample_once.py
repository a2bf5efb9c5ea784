"""Minimal driver for counter collection: N launches of the sampling kernel (+ one decode) at bench shape."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_schedule(sch.ddpm_table(1000))
g = torch.Generator().manual_seed(1)
c, e, s = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
for _ in range(n):
    lat = eng.sample(c, e, s, prec, seed=1)
eng.vae_decode(lat, None, prec)
torch.cuda.synchronize()
print("done")
