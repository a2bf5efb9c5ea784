"""Phase timeline of one denoising step of the 4-wave sampling kernel (s_memtime stamps): the fp32 parity kernel
(the 8-wave kernels of the other modes: tools/gpu_phase_profile8.py)."""
import os
import sys, json
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_schedule(sch.ddpm_table(50))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2:
    eng.set_clips_per_group(int(sys.argv[2]))
precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ("fp32",)
gen = torch.Generator().manual_seed(2)
c, e, s = (torch.randn(B, 256, generator=gen).cuda() for _ in range(3))
names = ["blk_start", "in_proj", "attention", "out_proj", "combine1", "LN1", "linear1", "GELU", "linear2", "combine2", "LN2"]
names = ["blk_start", "in_proj", "attention", "out_proj", "combine1", "LN1", "ffn.q0", "ffn.q1", "ffn.q2", "ffn.q3", "combine2", "LN2"]
names_bf16 = names
res = {}
for prec in precs:   # e.g. "fp32,fp32x"
    st = eng.profile_sample(c, e, s, prec, prof_step=3).astype(np.int64)
    per = {}
    tot = []
    for w in range(4):
        v = st[w]; n = int((v != 0).sum()); v = v[:n]
        d = np.diff(v)
        # layout: [0]=step start, then 9 x 11 stamps (+ nothing extra for skip), last = sched update
        nm = names_bf16 if prec == "bf16" else names
        K = len(nm)
        assert n == 1 + 9 * K + 1, n
        blocks = v[1:1 + 9 * K].reshape(9, K)
        prev = np.concatenate([[v[0]], blocks[:-1, -1]])
        seg = np.diff(np.concatenate([prev[:, None], blocks], axis=1), axis=1)  # [9][11]: first col = pre-block (skip linear)
        per[w] = seg
        tot.append(int(v[-1] - v[0]))
    seg = np.mean([per[w] for w in range(4)], axis=0)
    print(f"== {prec} B={B}: step cycles per wave {tot} (s_memtime ticks; 100 MHz const clock? see below)")
    print("   per-phase mean over waves, summed over 9 blocks, and share:")
    col = seg.sum(axis=0)
    for nme, cyc in zip(["skip_linear(pre-block)"] + nm[1:], col):
        print(f"   {nme:24s} {cyc:10.0f}  {100*cyc/col.sum():5.1f}%")
    res[prec] = {"total_ticks": tot, "phase_ticks": {n: float(c_) for n, c_ in zip(["skip_linear"] + nm[1:], col)}}
json.dump(res, open(REPO / "gpurun_out/phase_profile.json", "w"), indent=1)
