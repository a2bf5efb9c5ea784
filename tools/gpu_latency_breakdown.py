"""Where a single-clip job's time goes: HIP-event time of amuse_diffusion_backward (B = 1) against its parts (sample, decode) for DDIM-50
and DDPM-1000; run under `rocprofv3 --kernel-trace` the kernel trace gives the gaps between launches.  Usage: python tools/gpu_latency_breakdown.py [precision]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
gen = torch.Generator().manual_seed(3)
c, e, s = (torch.randn(1, 256, generator=gen).cuda() for _ in range(3))
out = {"latents": torch.empty(1, 128, device="cuda"), "poses": torch.empty(1, 300, 55, 3, device="cuda"), "trans": torch.empty(1, 300, 3, device="cuda")}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, n=30):
    ts, ws = [], []
    for i in range(n + 5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record(); fn(); e1.record(); e1.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1)); ws.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); ws.sort()
    return ts[len(ts) // 2], ws[len(ws) // 2]
for name, tab in (("DDIM-50", sch.ddim_table()), ("DDPM-1000", sch.ddpm_table(1000))):
    eng.set_schedule(tab)
    z = eng.sample(c, e, s, prec, seed=1)
    a = timed(lambda: eng.sample(c, e, s, prec, seed=1))
    b = timed(lambda: eng.vae_decode(z, None, prec))
    d = timed(lambda: eng.diffusion_backward(c, e, s, prec, seed=1, out=out))
    print(f"{name} {prec}: sample {a[0]:.3f} ms (wall {a[1]:.3f})   decode {b[0]:.3f} (wall {b[1]:.3f})   diffusion_backward {d[0]:.3f} (wall {d[1]:.3f})   "
          f"unaccounted {d[0] - a[0] - b[0]:+.3f} ms", flush=True)
