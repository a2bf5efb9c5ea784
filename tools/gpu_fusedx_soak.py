"""Determinism soak of the fp32x per-clip kernels (k_vae_fusedx / k_den_fusedx: LDS-DMA rings with counted waits - a protocol slip would show as run-to-run differences):
N decodes of 256 + 37 ragged clips and N Denoiser steps of 256 clips, every output compared bitwise with the first.  Usage: python tools/gpu_fusedx_soak.py [N]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_decode_path("clip")
g = torch.Generator().manual_seed(11)
z = torch.randn(256, 128, generator=g).cuda()
zr = torch.randn(37, 128, generator=g).cuda()
lens = [300 - (53 * i) % 299 for i in range(37)]
ref = ref_r = None
bad = 0
for i in range(N):
    a = eng.vae_decode(z, None, "fp32x", return_feats=True)["feats"]
    b = eng.vae_decode(zr, lens, "fp32x", return_feats=True)["feats"]
    if ref is None:
        ref, ref_r = a.clone(), b.clone()
    elif not (torch.equal(a, ref) and torch.equal(b, ref_r)):
        bad += 1
print(f"decode: {N} x (256 full-length + 37 ragged clips): {bad} runs differ from the first", flush=True)
eng.close()
den = HipEngine(wts.make_denoiser_weights(0, "trans_enc", True), None, "cuda:0", arch="trans_enc", diffusion_only=True)
den.set_decode_path("clip")
con, emo, sty = (torch.randn(256, 256, generator=g).cuda() for _ in range(3))
x = torch.randn(256, 300, 333, generator=g).cuda()
ref = None
bad = 0
for i in range(N):
    e = den.denoise_step(x, 501, con, emo if i % 2 == 0 else None, sty, precision="fp32x")
    if i < 2:
        ref = [e.clone()] if i == 0 else ref + [e.clone()]
    elif not torch.equal(e, ref[i % 2]):
        bad += 1
print(f"denoiser step: {N} x 256 clips (S = 304 / 303 alternating): {bad} runs differ from the first of their kind", flush=True)
