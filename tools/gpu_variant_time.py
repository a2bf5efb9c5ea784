"""A/B timing of libamuse_hip*.so variants on ONE box: each variant in its own process (AMUSE_HIP_LIB), repeated in
rounds so that clock drift hits all of them alike.  Usage: [AMUSE_VT_PREC=fp32x] python tools/gpu_variant_time.py [clips ...]"""
import os, subprocess, sys, glob
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import weights as wts, scheduler as sch
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    eng.set_schedule(sch.ddpm_table(1000))
    out = []
    prec = os.environ.get("AMUSE_VT_PREC", "bf16")
    for B in [int(x) for x in sys.argv[2:]]:
        gen = torch.Generator().manual_seed(1)
        c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
        eng.sample(c, e, s, prec, seed=1); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.sample(c, e, s, prec, seed=1); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        z = eng.sample(c, e, s, prec, seed=1)
        z = z[0] if isinstance(z, (tuple, list)) else z
        h = int(z.contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff   # (equal words = bitwise-equal variants)
        out.append(f"B={B}: min {min(ts):.2f} med {sorted(ts)[2]:.2f} [{h:08x}]")
    print("  ".join(out))
else:
    clips = sys.argv[1:] or ["1", "256", "768"]
    libs = sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so")))
    for rnd in range(3):
        for lib in libs:
            r = subprocess.run([sys.executable, __file__, "--child", *clips], env=dict(os.environ, AMUSE_HIP_LIB=lib),
                               capture_output=True, text=True)
            print(rnd, os.path.basename(lib).ljust(28), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
