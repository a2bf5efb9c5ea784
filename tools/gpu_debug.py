"""First-light diagnostics on a GPU box: prints error tables instead of asserting."""
import sys, time
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
from oracle import amuse_oracle as orc

G = REPO / "tests/golden"
def err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()), float(np.abs(b).max())

wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
Wd, Wp = orc.to_torch(wd), orc.to_torch(wp)
t0 = time.time(); eng = HipEngine(wd, wp); print("create s", time.time() - t0)
g = np.load(G / "denoiser_steps.npz")
con, emo, sty, x = (torch.from_numpy(g[k]) for k in ("con", "emo", "sty", "x_t"))

z = eng.counter_normal(2024, 5, 16, 3, 1).cpu().numpy()
print("rng", err(z, orc.counter_normal(2024, np.arange(5, 21), 3, 1)))

for prec in ("fp32", "bf16"):
    for t in (981, 501, 1):
        eps, tap = eng.denoise_step(x, t, con, emo, sty, prec, taps=True)
        torch.cuda.synchronize()
        taps = {}
        ref = orc.denoiser_forward(Wd, x, t, con, emo, sty, emulate_bf16=(prec == "bf16"), taps=taps)
        print(prec, "t", t, "eps vs oracle", err(eps, ref), "vs golden(fp32 ref)", err(eps, g[f"eps_t{t}"]))
        if t == 981:
            tp = tap.cpu().numpy()  # [11][16][128], rows = clip*5 + token (G=1 -> only clip 0)
            names = ["tokens"] + [f"encoder.input_blocks.{i}" for i in range(4)] + ["encoder.middle_block"] + \
                    [f"encoder.output_blocks.{i}" for i in range(4)]
            for i, n in enumerate(names):
                print("   tap", n, err(tp[i, :5], taps[n][0]))
            print("   tap final", err(tp[10, 0], ref[0]))
    e4 = eng.denoise_step(x, 501, con, None, sty, prec)
    e3 = eng.denoise_step(x, 501, con, None, None, prec)
    print(prec, "S=4", err(e4, orc.denoiser_forward(Wd, x, 501, con, None, sty, emulate_bf16=(prec == "bf16"))),
          "S=3", err(e3, orc.denoiser_forward(Wd, x, 501, con, None, None, emulate_bf16=(prec == "bf16"))))

# grouping G = 1..3 with B = 7
gen = torch.Generator().manual_seed(1)
c7, e7, s7, x7 = (torch.randn(7, n, generator=gen) for n in (256, 256, 256, 128))
ref7 = orc.denoiser_forward(Wd, x7, 321, c7, e7, s7)
for G_ in (1, 2, 3):
    eng.set_clips_per_group(G_)
    print("G", G_, err(eng.denoise_step(x7, 321, c7, e7, s7, "fp32"), ref7))
eng.set_clips_per_group(0)

# DDIM-50 trajectory vs golden
tr = np.load(G / "ddim50_traj.npz")
eng.set_schedule(sch.ddim_table())
for prec in ("fp32", "bf16"):
    lat, traj = eng.sample(tr["con"], tr["emo"], tr["sty"], prec, x_init=tr["x_T"], return_traj=True)
    torch.cuda.synchronize()
    for i in (10, 20, 30, 40, 50):
        print(prec, "ddim after", i, err(traj[i - 1], tr[f"x_after_{i}"]))

# VAE decode vs golden
vg = np.load(G / "vae_decode.npz")
for prec in ("fp32", "bf16"):
    out = eng.vae_decode(vg["z"], None, prec, return_feats=True)
    torch.cuda.synchronize()
    print(prec, "vae feats", err(out["feats"], vg["feats"]))
    ref = orc.vae_decode(Wp, torch.from_numpy(vg["z"]), emulate_bf16=(prec == "bf16"))
    print(prec, "vae feats vs oracle(emu)", err(out["feats"], ref))
    pr, trn = orc.feats_to_smplx(out["feats"].cpu())
    print(prec, "poses vs oracle-on-same-feats", err(out["poses"], pr), "trans", err(out["trans"], trn))
    outl = eng.vae_decode(vg["z"], None, prec, quat_mode="legacy")
    prl, _ = orc.feats_to_smplx(out["feats"].cpu(), "legacy")
    print(prec, "legacy poses", err(outl["poses"], prl))
out = eng.vae_decode(vg["z"][:2], [300, 173], "fp32", return_feats=True)
print("ragged", err(out["feats"], vg["feats_ragged"]))

# end-to-end + timing
eng.set_schedule(sch.ddpm_table())
for B in (1, 32, 256):
    gen = torch.Generator().manual_seed(2)
    c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
    for prec in ("fp32", "bf16"):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            o = eng.diffusion_backward(c, e, s, prec, seed=2024)
            torch.cuda.synchronize(); dt = time.time() - t0
        print(f"DDPM-1000 B={B} {prec}: {dt*1e3:.1f} ms -> {B*300/dt:.0f} frames/s; finite={bool(torch.isfinite(o['poses']).all())}")
