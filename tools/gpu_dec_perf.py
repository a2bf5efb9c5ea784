"""Sampling time of the trans_dec (latent) denoiser variant - k_sample_dec, the whole T-step loop in one launch, 16 clips per tile.
Usage: python tools/gpu_dec_perf.py [clips ...]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import scheduler as sch, weights as wts  # noqa: E402
from amuse_amd.engine import HipEngine  # noqa: E402

FLOP_STEP = 9 * (4 * 2 * 128 * 128 + 2 * 2 * 128 * 512 + 4 * 2 * 2 * 4 * 32)   # per clip: v, out, q, out2 projections + FFN + 4-key attention
eng = HipEngine(wts.make_denoiser_weights(0, "trans_dec"), wts.make_prior_weights(0), "cuda:0", arch="trans_dec")
for T, table in ((1000, sch.ddpm_table(1000)), (50, sch.ddim_table())):
    eng.set_schedule(table)
    for B in [int(a) for a in sys.argv[1:]] or [1, 16, 256, 4096]:
        g = torch.Generator().manual_seed(0)
        con, emo, sty = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
        for prec in ("bf16", "fp16", "fp32x", "fp32"):
            eng.sample(con, emo, sty, prec, seed=1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record(); eng.sample(con, emo, sty, prec, seed=1); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = min(ts)
            print(f"T={T:4d} B={B:5d} {prec:5s}: {ms:8.3f} ms = {ms / T * 1e3:7.2f} us/step, {B * 300 / ms * 1e3 / 1e6:8.3f} M frames/s (sampling only), "
                  f"{B * T * FLOP_STEP / ms / 1e9:7.1f} TFLOP/s", flush=True)
eng.close()
