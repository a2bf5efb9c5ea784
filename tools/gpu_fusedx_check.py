"""fp32x decode as one persistent workgroup per clip (csrc/k_vae_fusedx.hip, amuse_set_decode_path CLIP) against the staged fp32x path (k_vae_rows8x + k_vae_attn_x): the two must
produce the same bits (full-length and ragged clips); ms per decode by HIP events.  Each mode in its own process (the switch is read once).
Usage: python tools/gpu_fusedx_check.py [clips ...]"""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, str(REPO))
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    out = sys.argv[2]
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    eng.set_decode_path(os.environ.get("FX_PATH", "auto"))   # the parent pins the kernel family per child process
    res = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B in [int(v) for v in sys.argv[3:]]:
        z = torch.randn(B, 128, generator=torch.Generator().manual_seed(B)).cuda()
        lens = [300 - (37 * i) % 290 for i in range(B)]
        for name, ln in (("full", None), ("ragged", lens)):
            d = eng.vae_decode(z, ln, "fp32x", return_feats=True)
            torch.cuda.synchronize()
            res[f"{B}_{name}_feats"] = d["feats"][: min(B, 8)].cpu().numpy()
            res[f"{B}_{name}_poses"] = d["poses"][: min(B, 8)].cpu().numpy()
            res[f"{B}_{name}_sum"] = np.array([float(d["feats"].double().abs().sum()), float(d["poses"].double().abs().sum())])
        ts = []
        for i in range(6):
            e0.record(); eng.vae_decode(z, None, "fp32x"); e1.record(); e1.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        print(f"  B={B}: min {min(ts):.3f} ms", flush=True)
    np.savez(out, **res)
else:
    import numpy as np
    Bs = sys.argv[1:] or ["64", "256"]
    outs = {}
    for mode in ("staged", "fusedx"):
        env = dict(os.environ)
        env["FX_PATH"] = "clip" if mode == "fusedx" else "fused"
        out = f"/tmp/fusedx_{mode}.npz"
        print(f"--- {mode}", flush=True)
        r = subprocess.run([sys.executable, __file__, "--child", out, *Bs], env=env, capture_output=True, text=True)
        print(r.stdout.rstrip() or r.stderr[-2000:], flush=True)
        if r.returncode:
            print(r.stderr[-2000:])
            sys.exit(1)
        outs[mode] = np.load(out)
    for k in outs["staged"].files:
        a, b = outs["staged"][k], outs["fusedx"][k]
        same = np.array_equal(a, b)
        print(f"{k:24s} {'bitwise equal' if same else 'max |diff| %.3e (max |staged| %.3e)' % (np.abs(a - b).max(), np.abs(a).max())}", flush=True)
