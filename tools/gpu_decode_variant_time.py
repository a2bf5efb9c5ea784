"""A/B timing of the fused decode kernel across libamuse_hip*.so variants on ONE box (each variant in its own process,
AMUSE_HIP_LIB), two rounds; the bracketed word is a checksum of the poses (equal = bitwise-equal variants).  Usage: python tools/gpu_decode_variant_time.py [clips ...]"""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    eng.set_decode_path("fused")
    PREC = os.environ.get("AMUSE_DV_PREC", "bf16")   # fp32x: the k_vae_rows8x + k_vae_attn_x pair
    out = []
    for B in [int(x) for x in sys.argv[2:]]:
        z = torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(7):
            e0.record(); eng.vae_decode(z, None, PREC); e1.record(); torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        h = int(eng.vae_decode(z, None, PREC)["poses"].view(torch.int32).to(torch.int64).sum().item()) & 0xffffffff
        out.append(f"B={B}: min {min(ts):.3f} med {sorted(ts)[2]:.3f} ms [{h:08x}]")
    print("  ".join(out))
else:
    clips = sys.argv[1:] or ["1", "256"]
    libs = sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so")))
    for rnd in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, __file__, "--child", *clips], env=dict(os.environ, AMUSE_HIP_LIB=lib),
                               capture_output=True, text=True)
            print(rnd, os.path.basename(lib).ljust(30), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
