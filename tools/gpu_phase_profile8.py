"""Phase timeline of one denoising step of the 8-wave bf16 sampling kernel (s_memtime stamps, [8 waves][96])."""
import sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_schedule(sch.ddpm_table(50))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = torch.Generator().manual_seed(2)
c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
st = eng.profile_sample(c, e, s, "bf16", prof_step=3).astype(np.int64).reshape(8, 96)
names = ["pre(skip)", "P1", "C1", "FFN", "C2"]
for w in range(8):
    v = st[w]; n = int((v != 0).sum()); v = v[:n]
    assert n == 1 + 9 * 5 + 1, n
    blocks = v[1:46].reshape(9, 5)
    prev = np.concatenate([[v[0]], blocks[:-1, -1]])
    seg = np.diff(np.concatenate([prev[:, None], blocks], axis=1), axis=1)
    tot = v[-1] - v[0]
    print(f"wave {w}: step {tot} ticks; sums " + "  ".join(f"{nm} {int(x)}" for nm, x in zip(names, seg.sum(axis=0))) + f"  tail {int(v[-1]-v[-2])}")
    if w in (0, 4):
        for b in range(9):
            print("    blk", b, " ".join(f"{int(x):6d}" for x in seg[b]))
