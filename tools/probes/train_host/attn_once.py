"""The training step's attention at its shape, a few calls each way (for rocprofv3 --pmc passes).  Usage: python tools/probes/train_host/attn_once.py [B S]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from amuse_amd import train_ops as T  # noqa: E402
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 300)
dev = "cuda:0"
torch.manual_seed(0)
qkv = torch.randn(B * S, 384, device=dev)
dout = torch.randn(B * S, 128, device=dev)
for _ in range(4):
    o, lse = T.attn_fwd(qkv, B, S, 0.1, 1, 2)
    T.attn_bwd(qkv, o, lse, dout, B, S, 0.1, 1, 2)
torch.cuda.synchronize()
