"""Per-step time of the pose-space (diffusion_only) denoiser variants on the GPU: staged kernels vs the fused per-clip kernel.
Usage: python tools/gpu_den_perf.py [clips ...]   (default 64 256)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import scheduler as sch, weights as wts  # noqa: E402
from amuse_amd.engine import HipEngine  # noqa: E402

FLOP_STEP = lambda S: 9 * S * 393216 + 4 * S * 65536 + 9 * 4 * (2 * 2 * S * S * 32) + 2 * 2 * 300 * 333 * 128   # noqa: E731
ATTN_STEP = lambda S: 9 * 4 * (2 * 2 * S * S * 32)   # noqa: E731

clips = [int(a) for a in sys.argv[1:]] or [64, 256]
for arch in ("trans_enc", "trans_dec"):
    eng = HipEngine(wts.make_denoiser_weights(0, arch, True), None, "cuda:0", arch=arch, diffusion_only=True)
    T = 10
    eng.set_schedule(sch.ddim_table(T))
    for B in clips:
        g = torch.Generator().manual_seed(0)
        con, emo, sty = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
        for prec in ("bf16", "fp16", "fp32x", "fp32"):
            for path in (("staged", "fused") if (arch == "trans_enc" and prec in ("bf16", "fp16", "fp32x")) else ("staged",)):   # (fp32x "fused" = stages 1..8 on k_vae_rows8x)
                eng.set_decode_path(path)
                eng.sample(con, emo, sty, prec, seed=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 3 if prec in ("bf16", "fp16") else 1
                for _ in range(n):
                    eng.sample(con, emo, sty, prec, seed=1)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n / T * 1e3
                S = 304 if arch == "trans_enc" else 300
                print(f"{arch:9s} B={B:4d} {prec:5s} {path:6s}: {ms:8.3f} ms/step  {B * FLOP_STEP(S) / ms / 1e9:7.1f} TFLOP/s "
                      f"(DDIM-50 job {50 * ms:7.1f} ms = {B * 300 / (50 * ms) * 1e3 / 1e6:6.2f} M frames/s)", flush=True)
    eng.close()
