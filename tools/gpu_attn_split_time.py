"""Staged decode (bf16 / fp32x, ms) at 32..128 clips across libamuse_hip*.so variants built with other -DAMUSE_ATTN_SPLIT_MAX values (tools/build_variant.sh NAME k_vae.hip -DAMUSE_ATTN_SPLIT_MAX=N): where splitting a (clip, head) pair over five attention workgroups stops paying."""
import sys, os, glob, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
    eng.set_decode_path("staged")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for B in (32, 40, 48, 63, 96, 128):
        z = torch.randn(B, 128, generator=torch.Generator().manual_seed(B)).cuda()
        r = []
        for prec in ("bf16", "fp32x"):
            ts = []
            for i in range(7):
                e0.record(); eng.vae_decode(z, None, prec); e1.record(); e1.synchronize()
                if i >= 2: ts.append(e0.elapsed_time(e1))
            r.append(min(ts))
        out.append(f"B={B}: {r[0]:.3f}/{r[1]:.3f}")
    print("  ".join(out))
else:
    for lib in sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so"))):
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AMUSE_HIP_LIB=lib), capture_output=True, text=True)
        print(os.path.basename(lib).ljust(28), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
