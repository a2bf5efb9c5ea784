"""Prints the measured error of every parity check whose DESIGN.md row quotes a bound (uses the oracle: lives under tests/)."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, audio_weights as aw
from amuse_amd.engine import HipEngine
from amuse_amd.audio import AudioEngine
from oracle import amuse_oracle as orc, audio_oracle as ao

G = REPO / "tests/golden"
wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
eng = HipEngine(wd, wp)
e = lambda a, b: float((torch.as_tensor(a).cpu().double() - torch.as_tensor(b).cpu().double()).abs().max())
g = np.load(G / "vae_encode.npz")
feats = torch.from_numpy(g["feats"].astype(np.float32))
o = eng.vae_encode(feats, None, "fp32")
print("encode mu vs golden", e(o["mu"], g["mu"]), " std rel", e(o["std"].cpu() / torch.from_numpy(g["std"]), torch.ones(2, 128)))
o = eng.vae_encode(feats, [300, 211], "fp32")
print("encode ragged mu", e(o["mu"], g["mu_ragged"]))
o = eng.vae_encode(feats, None, "bf16")
print("encode bf16 mu", e(o["mu"], g["mu"]))
d = np.load(G / "denoiser_steps.npz")
ts = [int(v) for v in d["timesteps_batch"]]
ac = orc.SchedulerBase().alphas_cumprod
z0 = torch.from_numpy(d["x_t"]) / ac[torch.tensor(ts)].sqrt()[:, None]
o = eng.diffusion_forward(z0, torch.zeros(3, 128), ts, d["con"], d["emo"], d["sty"], "fp32")
print("diffusion_forward eps vs golden", e(o["noise_pred"], d["eps_batch_t"]))
gen = torch.Generator().manual_seed(3)
poses = 0.6 * torch.randn(2, 300, 55, 3, generator=gen)
trans = torch.randn(2, 300, 3, generator=gen)
f = eng.smplx_to_feats(poses, trans).cpu()
print("smplx_to_feats", e(f[..., :330], orc.axis_angle_to_rotation_6d(poses.double()).reshape(2, 300, 330)))
W = {n: aw.make_ast_weights(0, n) for n in aw.ENCODERS}
au = AudioEngine(W["con"], W["emo"], W["sty"])
tt = torch.arange(159744, dtype=torch.float32) / 16000.0
w = (0.2 * torch.sin(2 * np.pi * 220.0 * tt) + 0.05 * torch.randn(159744, generator=gen))[None]
fb = au.fbank(w).cpu()
ref = ao.prepare_fbank(w[0])[None]
print("fbank max / mean", e(fb, ref), float((fb - ref).abs().mean()))
Wt = ao.to_torch(W["emo"])
taps = {}
with torch.no_grad():
    r32 = ao.ast_forward(Wt, ref, True, False, taps)
feat, hid = au.encode("emo", ref, tap_block=11)
print("ast whole-net rel-L2 (block 11)", float((hid.cpu() - taps["block11"]).norm() / taps["block11"].norm()), " feature max err / max", e(feat, r32) / float(r32.abs().max()))
