import sys, torch
sys.path.insert(0, ".")
from amuse_amd import train_ops as T
from amuse_amd import _lib
B, S = 32, 300
dev = "cuda:0"
qkv = torch.randn(B * S, 384, device=dev)
o, lse = T.attn_fwd(qkv, B, S, 0.1, 1, 2)
# graph of 20 calls: no host in the way
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(20):
        T.attn_fwd(qkv, B, S, 0.1, 1, 2)
for _ in range(3): g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"attn_fwd: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call (graph of 20)")
