"""One fp32x step of the pose-space Denoiser (diffusion_only + trans_enc, S = 304) with its blocks on the per-clip kernel (csrc/k_vae_fusedx.hip k_den_fusedx, amuse_set_decode_path CLIP)
against the row / attention launches (=0): teacher-forced eps_hat of 256 clips compared, and ms per step over a DDIM-10 loop (HIP events).  One process per mode.
Usage: python tools/gpu_den_fusedx_check.py [clips]"""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, str(REPO))
    from amuse_amd import scheduler as sch, weights as wts
    from amuse_amd.engine import HipEngine
    out, B = sys.argv[2], int(sys.argv[3])
    eng = HipEngine(wts.make_denoiser_weights(0, "trans_enc", True), None, "cuda:0", arch="trans_enc", diffusion_only=True)
    eng.set_decode_path(os.environ.get("FX_PATH", "auto"))   # the parent pins the kernel family per child process
    g = torch.Generator().manual_seed(0)
    con, emo, sty = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
    x = torch.randn(B, 300, 333, generator=g).cuda()
    res = {}
    for name, (e, s_) in (("ces", (emo, sty)), ("c", (None, None))):
        eps = eng.denoise_step(x, 501, con, e, s_, precision="fp32x")
        torch.cuda.synchronize()
        res["eps_" + name] = eps[:4].cpu().numpy()
        res["sum_" + name] = np.array([float(eps.double().abs().sum())])
    T = 10
    eng.set_schedule(sch.ddim_table(T))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(4):
        e0.record(); xs = eng.sample(con, emo, sty, "fp32x", seed=1); e1.record(); e1.synchronize()
        if i >= 1:
            ts.append(e0.elapsed_time(e1) / T)
    res["x_final"] = xs[:2].cpu().numpy()
    print(f"  B={B}: {min(ts):.3f} ms per step", flush=True)
    np.savez(out, **res)
else:
    import numpy as np
    B = sys.argv[1] if len(sys.argv) > 1 else "256"
    outs = {}
    for mode in ("staged", "fusedx"):
        env = dict(os.environ, FX_PATH="clip" if mode == "fusedx" else "fused")
        out = f"/tmp/denfx_{mode}.npz"
        print(f"--- {mode}", flush=True)
        r = subprocess.run([sys.executable, __file__, "--child", out, B], env=env, capture_output=True, text=True)
        print(r.stdout.rstrip() or r.stderr[-2000:], flush=True)
        if r.returncode:
            print(r.stderr[-3000:]); sys.exit(1)
        outs[mode] = np.load(out)
    for k in outs["staged"].files:
        a, b = outs["staged"][k], outs["fusedx"][k]
        print(f"{k:12s} {'bitwise equal' if np.array_equal(a, b) else 'max |diff| %.3e (max |staged| %.3e)' % (np.abs(a - b).max(), np.abs(a).max())}", flush=True)
