"""Throughput of the audio front-end (fbank + 3 x AST) on one GPU.  Usage: python tools/gpu_audio_perf.py [B ...]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import audio_weights as aw
from amuse_amd.audio import AudioEngine

eng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS))
FLOP = 3 * 12 * (2 * 1214 * 768 * (2304 + 768 + 2 * 3072) + 4 * 1214 * 1214 * 768)   # per clip, GEMMs + attention
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 32, 64]:
    w = 0.1 * torch.randn(B, 160000, device="cuda:0")
    for _ in range(2):
        eng.features(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    e0.record()
    for _ in range(n):
        eng.features(w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"B={B:3d}: {ms:9.2f} ms  = {ms / B:7.2f} ms/clip, {B * FLOP / ms / 1e9:7.1f} TFLOP/s", flush=True)
