"""Decode (MotionPrior.decode + 6D -> axis-angle) timing, staged kernels vs the fused per-clip kernel, HIP events."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine

eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
Bs = [int(v) for v in sys.argv[1:]] or [1, 8, 16, 24, 32, 64, 128, 256, 512, 768]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in Bs:
    z = torch.randn(B, 128, generator=torch.Generator().manual_seed(B)).cuda()
    row = []
    for path in ("staged", "fused"):
        eng.set_decode_path(path)
        ts = []
        for i in range(7):
            e0.record(); eng.vae_decode(z, None, "bf16"); e1.record(); e1.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        row.append(min(ts))
    fl = B * 1.76e9
    print(f"B={B:4d}  staged {row[0]:8.3f} ms ({fl / row[0] / 1e9:7.1f} TFLOP/s)   fused {row[1]:8.3f} ms ({fl / row[1] / 1e9:7.1f} TFLOP/s"
          f" = {fl / row[1] / 1e9 / 2500 * 100:4.1f} % of bf16 MFMA peak)", flush=True)
