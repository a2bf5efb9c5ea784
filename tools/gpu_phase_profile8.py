"""Phase timeline of one denoising step of the 8-wave sampling kernels (s_memtime stamps, [8 waves][96]): bf16 (k_sample8) or,
with a second argument fp32x, k_sample8x.  Usage: python tools/gpu_phase_profile8.py [clips] [bf16|fp32x]"""
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
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
st = eng.profile_sample(c, e, s, prec, prof_step=3).astype(np.int64).reshape(8, 96)
names = ["pre(skip)", "P1", "C1", "FFN", "C2"]
names10 = ["pre(skip)", "P1", "C1", "F1a", "F1b", "geluA", "F2a", "geluB", "F2b", "C2"]   # k_sample8x stamps inside its FFN half
for w in range(8):
    v = st[w]; n = int((v != 0).sum()); v = v[:n]
    K = 10 if n == 1 + 9 * 10 + 1 else 5
    assert n == 1 + 9 * K + 1, n
    nm = names10 if K == 10 else names
    blocks = v[1:1 + 9 * K].reshape(9, K)
    prev = np.concatenate([[v[0]], blocks[:-1, -1]])
    seg = np.diff(np.concatenate([prev[:, None], blocks], axis=1), axis=1)
    tot = v[-1] - v[0]
    print(f"wave {w}: step {tot} ticks; sums " + "  ".join(f"{n_} {int(x)}" for n_, x in zip(nm, seg.sum(axis=0))) + f"  tail {int(v[-1]-v[-2])}")
    if w in (0, 4):
        for b in range(9):
            print("    blk", b, " ".join(f"{int(x):6d}" for x in seg[b]))
        # absolute times of block 2's stamps relative to the block start of wave 0: who waits for whom
    if w in (0, 4):
        print("    blk 2 stamps relative to its first:", " ".join(str(int(x - blocks[2][0])) for x in blocks[2]), " (abs start", int(blocks[2][0] - st[0][1]), ")")
