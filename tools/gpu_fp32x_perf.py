"""Parity modes side by side on one box: ms per job and us per step of the sampling loop in fp32 (v_mfma_f32_16x16x4_f32),
fp32x (split-fp16 operands, 3 x v_mfma_f32_16x16x32_f16 per product) and bf16, DDPM-1000 and DDIM-50, plus the
teacher-forced eps_hat error of each against the reference's golden vectors."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
g = np.load(REPO / "tests/golden/denoiser_steps.npz")
for prec in ("fp32", "fp32x", "bf16"):
    errs = [float(np.abs(eng.denoise_step(g["x_t"], t, g["con"], g["emo"], g["sty"], prec).cpu().numpy() - g[f"eps_t{t}"]).max()) for t in (981, 501, 1)]
    print(f"eps_hat vs reference golden, {prec:6s}: " + " ".join(f"{e:.2e}" for e in errs), flush=True)
clips = [int(x) for x in sys.argv[1:]] or [1, 256, 768]
for name, table in (("ddpm1000", sch.ddpm_table()), ("ddim50", sch.ddim_table())):
    eng.set_schedule(table)
    T = len(table.timesteps)
    for B in clips:
        gen = torch.Generator().manual_seed(1)
        c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
        for prec in ("fp32", "fp32x", "bf16"):
            eng.sample(c, e, s, prec, seed=1); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); eng.sample(c, e, s, prec, seed=1); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = min(ts)
            print(f"{name} B={B:4d} {prec:6s}: {ms:9.3f} ms  {ms / T * 1e3:7.2f} us/step  {B * 300 / ms * 1e3:12.0f} frames/s (sampling only)", flush=True)
