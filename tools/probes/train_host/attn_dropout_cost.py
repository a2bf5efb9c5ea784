import sys, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[3]))
from amuse_amd import train_ops as T
def timed(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, S = 32, 300
qkv = torch.randn(B * S, 384, device="cuda:0"); dout = torch.randn(B * S, 128, device="cuda:0")
for p in (0.0, 0.1):
    o, lse = T.attn_fwd(qkv, B, S, p, 1, 2)
    print(f"p={p}: fwd {timed(lambda: T.attn_fwd(qkv, B, S, p, 1, 2)):.1f} us  bwd {timed(lambda: T.attn_bwd(qkv, o, lse, dout, B, S, p, 1, 2)):.1f} us")
