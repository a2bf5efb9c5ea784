"""Time amuse_vae_encode / amuse_vae_decode (HIP events) at a few batch sizes.  Usage: python tools/gpu_encode_perf.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine

eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0), "cuda:0")
for B in (1, 64, 256, 512):
    feats = 0.5 * torch.randn(B, 300, 333, device="cuda:0")
    z = torch.randn(B, 128, device="cuda:0")
    for prec in ("bf16", "fp32"):
        for name, fn in (("encode", lambda: eng.vae_encode(feats, None, prec)), ("decode", lambda: eng.vae_decode(z, None, prec))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"B={B:4d} {prec} {name}: {ms:8.3f} ms  ({B * 300 / ms * 1e3 / 1e6:7.2f} M frames/s)", flush=True)
