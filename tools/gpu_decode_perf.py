"""Decode (MotionPrior.decode + 6D -> axis-angle) timing, HIP events: bf16 staged kernels vs the fused per-clip kernel, and the
two parity modes on the staged kernels (fp32: v_mfma_f32_16x16x4_f32; fp32x: split-fp16 operands), with and without taps."""
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
    eng.set_decode_path("auto")
    for prec in ("fp32", "fp32x"):
        ts = []
        for i in range(4):
            e0.record(); eng.vae_decode(z, None, prec); e1.record(); e1.synchronize()
            if i >= 1:
                ts.append(e0.elapsed_time(e1))
        row.append(min(ts))
    eng.set_decode_path("fused")   # the tapped instantiation of the fused kernel computes the same bits with another register allocation
    ts = []
    for i in range(5):
        e0.record(); eng.vae_decode(z, None, "bf16", return_taps=True); e1.record(); e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    row.append(min(ts))
    ts = []
    for i in range(7):   # the fused kernel's fp16 build (the fp16 mode decodes on it at any batch size)
        e0.record(); eng.vae_decode(z, None, "fp16"); e1.record(); e1.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    row.append(min(ts))
    fl = B * 1.76e9
    print(f"B={B:4d}  staged {row[0]:8.3f} ms ({fl / row[0] / 1e9:7.1f} TFLOP/s)   fused {row[1]:8.3f} ms ({fl / row[1] / 1e9:7.1f} TFLOP/s"
          f" = {fl / row[1] / 1e9 / 2500 * 100:4.1f} % of bf16 MFMA peak)   fused+taps {row[4]:8.3f} ms   fused fp16 {row[5]:8.3f} ms   fp32 {row[2]:8.3f} ms   fp32x {row[3]:8.3f} ms", flush=True)
