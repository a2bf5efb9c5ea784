"""BASELINE config 5 shape: edit_gesture emotion_control as one batch - 8 source + 8 target waveforms
through the audio front-end, 8 x 8 content / emotion combinations (the emotion embedding of a target swapped in,
infer_ldm.py:392-410) = 64 jobs, sampled with DDIM-50 (the reference's edit path) and DDPM-1000, decoded to SMPL-X."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import audio_weights as aw, weights as wts
from amuse_amd.infer_ldm import PretrainedLPDM_v1

m = PretrainedLPDM_v1.from_state_dicts(wts.make_denoiser_weights(0), wts.make_prior_weights(0), device="cuda:0")
m.precision = "bf16"
m.set_audio_encoders(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS))
gen = torch.Generator().manual_seed(0)
src = 0.1 * torch.randn(8, 160000, generator=gen)
tgt = 0.1 * torch.randn(8, 160000, generator=gen)


def run(sampler):
    m.set_sampler(sampler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    con, emo, sty = m.audio_engine.features(torch.cat([src, tgt]))          # 16 clips
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    z_con = con[:8].repeat_interleave(8, 0)                                   # content i ...
    z_sty = sty[:8].repeat_interleave(8, 0)
    z_emo = emo[8:].repeat(8, 1)                                              # ... with the emotion of target j
    out = m.diffusion_backward(64, z_con, z_emo, z_sty)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert out["poses"].shape == (64, 300, 55, 3) and bool(torch.isfinite(out["poses"]).all())
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


for s in ("ddim", "ddpm"):
    run(s)
    a, b = zip(*(run(s) for _ in range(3)))
    print(f"{s}: audio front-end (16 clips) {min(a):7.2f} ms, sampling + decode (64 jobs) {min(b):7.2f} ms "
          f"-> {64 * 300 / ((min(a) + min(b)) * 1e-3):,.0f} frames/s")
