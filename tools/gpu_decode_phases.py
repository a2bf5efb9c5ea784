"""Phase timeline of the fused decode kernel: run with AMUSE_HIP_LIB pointing at a -DAMUSE_FPROF=1 build
(tools/build_variant.sh fprof k_vae_fused.hip -DAMUSE_FPROF=1); the library prints s_memtime deltas (cycles at 100 MHz x ...
see below) of wave 0 / workgroup 0 for decoder blocks 1 and 6 on its third launch."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_decode_path("fused")
z = torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3):
    eng.vae_decode(z, None, "bf16")
torch.cuda.synchronize()
