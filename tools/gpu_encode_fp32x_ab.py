import sys, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
for B in (64, 256, 512):
    f = 0.5 * torch.randn(B, 300, 333, generator=torch.Generator().manual_seed(1)).cuda()
    for path in ("staged", "fused"):
        eng.set_decode_path(path)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for i in range(12):
            e0.record(); eng.vae_encode(f, None, "fp32x"); e1.record(); torch.cuda.synchronize()
            if i >= 4: ts.append(e0.elapsed_time(e1))
        print(f"B={B:4d} fp32x encode, rows kernel {'k_vae_rows<f16x2> (4 waves, split-K)' if path == 'staged' else 'k_vae_rows8x<ENC> (stages 1..9)'}: {min(ts):.3f} ms")
