"""N DDIM steps of the pose-space (diffusion_only + trans_enc, S = 304) denoiser at the bench shape - the program under rocprofv3
(--kernel-trace --stats / --pmc).  Usage: python tools/gpu_den_once.py [clips] [precision] [steps] [staged|fused]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import scheduler as sch, weights as wts  # noqa: E402
from amuse_amd.engine import HipEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 10
path = sys.argv[4] if len(sys.argv) > 4 else "auto"
eng = HipEngine(wts.make_denoiser_weights(0, "trans_enc", True), None, "cuda:0", arch="trans_enc", diffusion_only=True)
eng.set_schedule(sch.ddim_table(T))
eng.set_decode_path(path)
g = torch.Generator().manual_seed(0)
con, emo, sty = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
for _ in range(2):
    x = eng.sample(con, emo, sty, prec, seed=1)
torch.cuda.synchronize()
print("ok", float(x.abs().mean()))
