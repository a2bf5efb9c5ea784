"""A few training iterations for rocprofv3 --kernel-trace --stats (per-kernel durations of the step).  Usage: rocprofv3 ... -- python3 tools/probes/train_host/step_once.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from amuse_amd.train_gesture import build_trainer, synthetic_batch
dev = torch.device("cuda", 0)
torch.manual_seed(1)
tr = build_trainer(dev, 0, 1)
b = [synthetic_batch(32, i, dev) for i in range(2)]
for i in range(8):
    tr.train_step(b[i % 2])
torch.cuda.synchronize()
