"""What the reference's OWN form of the headline job costs on this GPU: PyTorch-ROCm eager modules in a Python denoising loop.

The reference cannot travel to the GPU box, so this runs the build's torch twins of its modules (amuse_amd/nn_modules.py: `Denoiser`, `MotionPrior` - the reference's
state-dict keys, pinned to the reference modules' golden vectors by tests/test_train_cpu.py) on torch's stock layers (AMUSE_TRAIN_FUSED=0: nn.MultiheadAttention,
F.layer_norm, F.gelu, torch's GEMMs - no kernel of this library), in eval mode, through the loop the reference's `diffusion_backward` runs (infer_ldm.py:116-161):
T x (Denoiser.forward + scheduler update), then `MotionPrior.decode`.  The scheduler update is ONE fused expression per step (amuse_amd/scheduler.py's row), cheaper
than diffusers' `step()`; the 6D -> axis-angle conversion is left out.  Both favour this baseline.  fp32 (the reference's arithmetic) and bf16 autocast.

  python tools/gpu_torch_eager_baseline.py [clips]      -> one line per (sampler, precision): ms per job, SMPL-X frames/s
"""
import json
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

WHAT = ("torch twins of the reference modules (amuse_amd/nn_modules.py, pinned to the reference's goldens) on torch's stock layers, eval mode, eager: Python loop of "
        "T x (Denoiser.forward + one fused scheduler expression) + MotionPrior.decode; no axis-angle conversion")


def measure(B: int, dev, samplers=("DDPM-1000", "DDIM-50"), precisions=("fp32", "bf16 autocast"), repeats=2):
    """[{sampler, clips, precision, ms_per_job, frames_per_s}] - the twins run on torch's own layers for the duration of the call (AMUSE_TRAIN_FUSED=0, restored after)."""
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.nn_modules import Denoiser, MotionPrior, load_numpy_state
    from amuse_amd.train_gesture import TrainModeInnerSampler
    prev = os.environ.get("AMUSE_TRAIN_FUSED")
    os.environ["AMUSE_TRAIN_FUSED"] = "0"
    try:
        g = torch.Generator().manual_seed(0)
        den = load_numpy_state(Denoiser(dropout=0.1), wts.make_denoiser_weights(0)).to(dev).eval()
        prior = load_numpy_state(MotionPrior(dropout=0.1), wts.make_prior_weights(0)).to(dev).eval()
        con, emo, sty = (torch.randn(B, 256, generator=g).to(dev) for _ in range(3))
        update = TrainModeInnerSampler.scheduler_update

        @torch.no_grad()
        def job(table, steps=None):
            coef = torch.as_tensor(table.coef)
            x = torch.randn(B, 128, device=dev) * table.init_noise_sigma
            ts = table.timesteps if steps is None else table.timesteps[:steps]
            for i, t in enumerate(ts):
                eps = den(x[:, None], int(t), con, emo, sty)[0][:, 0]
                z = torch.randn_like(x) if float(coef[i, 5]) != 0 else None
                x = update(coef[i], x, eps, z)
            return prior.decode(x[None].float(), [300] * B)

        def timed(table, autocast):
            ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else torch.autocast("cuda", enabled=False)
            with ctx:
                job(table, steps=20)                # warm-up: kernels loaded, allocator settled
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                out = job(table)
                torch.cuda.synchronize(dev)
                dt = time.perf_counter() - t0
            assert bool(torch.isfinite(out).all())
            return dt

        res = []
        for name in samplers:
            table = sch.ddpm_table() if name == "DDPM-1000" else sch.ddim_table()
            for prec in precisions:
                dt = min(timed(table, prec != "fp32") for _ in range(repeats))
                res.append({"sampler": name, "clips": B, "precision": prec, "ms_per_job": round(dt * 1e3, 1), "frames_per_s": round(B * 300 / dt, 1)})
        return res
    finally:
        if prev is None:
            os.environ.pop("AMUSE_TRAIN_FUSED", None)
        else:
            os.environ["AMUSE_TRAIN_FUSED"] = prev


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    for r in measure(B, torch.device("cuda", 0)):
        print(json.dumps(dict(r, what=WHAT)), flush=True)
