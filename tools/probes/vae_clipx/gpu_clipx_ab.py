"""A/B of the fp32x decode paths from 64 clips up: ONE launch per chunk (k_vae_clipx, default) against the launch sequence it replaces (k_vae_rows8x + k_vae_attn_x,
AMUSE_VAE_CLIPX=0) - each in its own process (the switch is read once), same latents: bitwise comparison of features / poses (full and ragged lengths) and timings.
   python tools/gpu_clipx_ab.py [clips ...]"""
import os, subprocess, sys, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    out = []
    for B in [int(x) for x in sys.argv[2:]]:
        g = torch.Generator().manual_seed(B)
        z = torch.randn(B, 128, generator=g).cuda()
        o = eng.vae_decode(z, None, "fp32x", return_feats=True)
        lens = [300 - (7 * i) % 200 for i in range(B)]
        o2 = eng.vae_decode(z, lens, "fp32x", return_feats=True)
        torch.cuda.synchronize()
        h = hashlib.sha256(o["feats"].cpu().numpy().tobytes() + o["poses"].cpu().numpy().tobytes()).hexdigest()[:12]
        h2 = hashlib.sha256(o2["feats"].cpu().numpy().tobytes() + o2["poses"].cpu().numpy().tobytes()).hexdigest()[:12]
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.vae_decode(z, None, "fp32x"); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append(f"B={B}: {min(ts[1:]):.3f} ms  full [{h}] ragged [{h2}] finite {bool(torch.isfinite(o['feats']).all())}")
    print("  ".join(out))
else:
    clips = sys.argv[1:] or ["64", "256", "768"]
    for rnd in range(2):
        for name, env in (("launch sequence (AMUSE_VAE_CLIPX=0)", {"AMUSE_VAE_CLIPX": "0"}), ("k_vae_clipx (one launch)       ", {})):
            r = subprocess.run([sys.executable, __file__, "--child", *clips], env=dict(os.environ, **env), capture_output=True, text=True)
            print(rnd, name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
