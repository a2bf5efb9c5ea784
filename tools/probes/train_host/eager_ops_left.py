"""Which eager torch operators are left in one train_gesture iteration (count and self CPU / device time per iteration), outside the layer Functions."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from amuse_amd.train_gesture import build_trainer, synthetic_batch
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda:0")
tr = build_trainer(dev)
batch = synthetic_batch(32, 1, dev)
for _ in range(8):
    tr.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.train_step(batch)
    torch.cuda.synchronize()
rows = [(e.key, e.count / 3, e.self_cpu_time_total / 3e3, e.self_device_time_total / 3e3) for e in prof.key_averages() if e.key.startswith("aten::") and e.count >= 3]
rows.sort(key=lambda r: -r[2])
print(f"{'operator':45s} {'calls/it':>9s} {'self CPU ms/it':>15s} {'device ms/it':>13s}")
for k, c, cpu, d in rows[:40]:
    print(f"{k:45s} {c:9.1f} {cpu:15.3f} {d:13.3f}")
print("sum over aten:: ops: self CPU", round(sum(r[2] for r in rows), 2), "ms/it, device", round(sum(r[3] for r in rows), 2), "ms/it, calls", round(sum(r[1] for r in rows)))
