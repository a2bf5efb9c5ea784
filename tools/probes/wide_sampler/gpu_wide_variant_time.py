"""A/B timing of the wide sampling kernel across libamuse_hip*.so variants on ONE box (each variant in its own process,
AMUSE_HIP_LIB; tools/build_variant.sh NAME k_samplerw.hip -DAMUSE_WABL=..).  Usage: python tools/gpu_wide_variant_time.py [clips ...]"""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import weights as wts, scheduler as sch
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    T = 200
    eng.set_schedule(sch.ddpm_table(T))
    eng.set_sampler_path("wide")
    out = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for B in [int(x) for x in sys.argv[2:]]:
        g = torch.Generator().manual_seed(B)
        c, e, s = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
        ts = []
        for i in range(4):
            e0.record(); eng.sample(c, e, s, "bf16", seed=1); e1.record(); torch.cuda.synchronize()
            if i >= 1:
                ts.append(e0.elapsed_time(e1))
        out.append(f"B={B}: {min(ts) / T * 1e3:.1f} us/step ({min(ts) / T * 2.4e6 / 58:.0f} cyc/stage)")
    print("  ".join(out))
else:
    clips = sys.argv[1:] or ["1536", "3072"]
    libs = sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so")))
    for rnd in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, __file__, "--child", *clips], env=dict(os.environ, AMUSE_HIP_LIB=lib),
                               capture_output=True, text=True)
            print(rnd, os.path.basename(lib).ljust(30), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
