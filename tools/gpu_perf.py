"""Timing sweep of the sampling kernel and the decode on a GPU box."""
import sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
eng = HipEngine(wd, wp)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
eng.set_schedule(sch.ddpm_table(T))
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
gen = torch.Generator().manual_seed(2)
for B, Gs in ((1, (1,)), (32, (1,)), (256, (1, 2, 3)), (768, (3,)), (1536, (3,)), (4096, (3,))):
    c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
    for prec in ("bf16", "fp32"):
        for G in Gs:
            eng.set_clips_per_group(G)
            dt = timeit(lambda: eng.sample(c, e, s, prec, seed=1))
            print(f"sample  B={B:5d} G={G} {prec}: {dt*1e3:8.2f} ms  {dt/T*1e6:7.2f} us/step  {B*300/dt:12.0f} frames/s(sampling only)")
    eng.set_clips_per_group(0)
    z = torch.randn(B, 128, generator=gen).cuda()
    for prec in ("bf16", "fp32"):
        dt = timeit(lambda: eng.vae_decode(z, None, prec))
        print(f"decode  B={B:5d}     {prec}: {dt*1e3:8.2f} ms  {dt/B*1e6:8.1f} us/clip")
