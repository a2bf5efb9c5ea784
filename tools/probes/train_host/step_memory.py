import sys, torch
sys.path.insert(0, ".")
from amuse_amd.train_gesture import build_trainer, synthetic_batch
dev = torch.device("cuda", 0)
tr = build_trainer(dev, 0, 1, grads_mode="sink")
b = synthetic_batch(32, 1, dev)
for _ in range(3): tr.train_step(b)
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
base = torch.cuda.memory_allocated()
tr.train_step(b); torch.cuda.synchronize()
print(f"allocated between steps {base/2**20:.0f} MiB, peak inside a step {torch.cuda.max_memory_allocated()/2**20:.0f} MiB")
