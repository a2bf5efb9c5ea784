"""One decode of 256 clips in each parity mode (the program under `rocprofv3 --kernel-trace --stats`: per-kernel times of the staged
fp32 / fp32x decode)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
z = torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()
for prec in sys.argv[2:] or ["fp32", "fp32x"]:
    for _ in range(3):
        eng.vae_decode(z, None, prec)
torch.cuda.synchronize()
print("done")
