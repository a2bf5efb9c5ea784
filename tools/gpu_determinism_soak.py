"""Race hunt: the 8-wave sampling kernels (bf16 k_sample8, its fp16 build k_sample8h, fp32x k_sample8x) must reproduce themselves bitwise, launch after launch, for every tiling
(a data race between its alternating wave groups would show up as a flaky mismatch); so must the fused decode kernel and the
audio front-end (DMA rings, hand-counted waits)."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
gen = torch.Generator().manual_seed(11)
bad = 0
for T, tab in ((100, sch.ddpm_table(100)), (50, sch.ddim_table())):
    eng.set_schedule(tab)
    for B, G in ((1, 0), (7, 1), (7, 2), (7, 3), (256, 0), (512, 0), (768, 0), (1000, 3)):
        c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
        eng.set_clips_per_group(G)
        for prec, reps in (("bf16", 24), ("fp16", 12), ("fp32x", 12)):
            ref = eng.sample(c, e, s, prec, seed=5).clone()
            n = 0
            for _ in range(reps):
                out = eng.sample(c, e, s, prec, seed=5)
                n += int(not torch.equal(out, ref))
            bad += n
            print(f"T={T} B={B} G={G} {prec}: {n} mismatching launches of {reps}; finite={bool(torch.isfinite(ref).all())}", flush=True)
eng.set_clips_per_group(0)
# the fused decode kernel and the audio front-end (LDS-DMA rings with hand-counted s_waitcnt: a protocol slip would show up here)
z = torch.randn(300, 128, generator=gen).cuda()
eng.set_decode_path("fused")
ref = eng.vae_decode(z, None, "bf16")["poses"].clone()
n = sum(int(not torch.equal(eng.vae_decode(z, None, "bf16")["poses"], ref)) for _ in range(16))
bad += n
print(f"fused decode, 300 clips: {n} mismatching launches of 16; finite={bool(torch.isfinite(ref).all())}", flush=True)
ref = eng.vae_decode(z, None, "fp16")["poses"].clone()
n = sum(int(not torch.equal(eng.vae_decode(z, None, "fp16")["poses"], ref)) for _ in range(16))
bad += n
print(f"fused decode (fp16 build), 300 clips: {n} mismatching launches of 16; finite={bool(torch.isfinite(ref).all())}", flush=True)
eng.set_decode_path("auto")
# the fp32x decode / encode (staged kernels, split-fp16 fragment images)
ref = eng.vae_decode(z, [300] * 299 + [123], "fp32x")["poses"].clone()
n = sum(int(not torch.equal(eng.vae_decode(z, [300] * 299 + [123], "fp32x")["poses"], ref)) for _ in range(6))
bad += n
print(f"fp32x decode, 300 clips: {n} mismatching launches of 6; finite={bool(torch.isfinite(ref).all())}", flush=True)
# the staged kernels' fp16 instantiations (decode below 64 clips, encode) and the bf16 ones they are built from
zs = torch.randn(37, 128, generator=gen).cuda()
for prec in ("bf16", "fp16"):
    ref = eng.vae_decode(zs, [300] * 36 + [77], prec, return_feats=True)
    enc = eng.vae_encode(ref["feats"], None, prec, eps=torch.zeros(37, 128))
    n = 0
    for _ in range(8):
        o = eng.vae_decode(zs, [300] * 36 + [77], prec, return_feats=True)
        e = eng.vae_encode(ref["feats"], None, prec, eps=torch.zeros(37, 128))
        n += int(not (torch.equal(o["poses"], ref["poses"]) and torch.equal(e["mu"], enc["mu"]) and torch.equal(e["std"], enc["std"])))
    bad += n
    print(f"staged decode + encode, 37 clips, {prec}: {n} mismatching calls of 8; finite={bool(torch.isfinite(ref['poses']).all())}", flush=True)
from amuse_amd import audio_weights as aw
from amuse_amd.audio import AudioEngine
aeng = AudioEngine(*(aw.make_ast_weights(0, k) for k in aw.ENCODERS))
for B in (1, 3, 34):
    w = 0.1 * torch.randn(B, 160000, generator=gen).cuda()
    ref = [t.clone() for t in aeng.features(w)]
    n = 0
    for _ in range(8):
        out = aeng.features(w)
        n += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
    bad += n
    print(f"audio features, {B} clips: {n} mismatching calls of 8; finite={all(bool(torch.isfinite(t).all()) for t in ref)}", flush=True)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
