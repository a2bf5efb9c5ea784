"""Sampling time (1000-step DDPM, bf16): tile kernel (k_sample8) vs wide kernel (k_sample_wide) per clip count."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
T = 1000
eng.set_schedule(sch.ddpm_table(T))
Bs = [int(v) for v in sys.argv[1:]] or [256, 768, 1024, 1536, 3072, 6144, 12288]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
FLOP = 19_120_640
for B in Bs:
    g = torch.Generator().manual_seed(B)
    c, e, s = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
    row = []
    for path in ("tile", "wide"):
        eng.set_sampler_path(path)
        ts = []
        for i in range(3):
            e0.record(); eng.sample(c, e, s, "bf16", seed=1); e1.record(); e1.synchronize()
            if i >= 1:
                ts.append(e0.elapsed_time(e1))
        row.append(min(ts))
    f = lambda ms: f"{ms:9.2f} ms {B * 300 / ms / 1e3:7.2f} M frames/s {B * T * FLOP / ms / 1e9:7.1f} TFLOP/s ({B * T * FLOP / ms / 1e9 / 25:4.1f} %)"
    print(f"B={B:6d}  tile {f(row[0])}   wide {f(row[1])}", flush=True)
