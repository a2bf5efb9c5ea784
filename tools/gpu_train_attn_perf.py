"""Per-call time of the training step's attention at its shapes: the library's fp32 MFMA kernels (csrc/k_train_attn.hip) against aten's efficient-attention
op, forward and backward, dropout 0.1.  Usage: python tools/gpu_train_attn_perf.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import train_ops as T  # noqa: E402

dev = "cuda:0"


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, S in ((32, 300), (32, 302), (32, 5)):
    qkv = torch.randn(B * S, 384, device=dev)
    dout = torch.randn(B * S, 128, device=dev)
    o, lse = T.attn_fwd(qkv, B, S, 0.1, 1, 2)
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, S, 3, 4, 32).unbind(2))
    ao, vl, ps, po = T._sdpa(q, k, v, None, True, 0.1, False)
    do = dout.view(B, S, 4, 32).transpose(1, 2)
    print(f"B={B} S={S}: forward  library {timed(lambda: T.attn_fwd(qkv, B, S, 0.1, 1, 2)):6.1f} us   aten {timed(lambda: T._sdpa(q, k, v, None, True, 0.1, False)):6.1f} us;  "
          f"backward library {timed(lambda: T.attn_bwd(qkv, o, lse, dout, B, S, 0.1, 1, 2)):6.1f} us   "
          f"aten {timed(lambda: T._sdpa_bwd(do, q, k, v, None, ao, vl, ps, po, 0.1, (True, True, True, False), False)):6.1f} us", flush=True)
