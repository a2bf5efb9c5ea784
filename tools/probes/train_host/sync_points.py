"""Where does the training step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") makes every synchronising torch
call warn with its Python stack; the HIP runtime's own blocking copies (hipMemcpyWithStream from the library's C++ side, pageable host -> device
uploads) are counted from the torch profiler beside it.  Usage: python tools/probes/train_host/sync_points.py"""
import collections, sys, traceback, warnings
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from amuse_amd.train_gesture import build_trainer, synthetic_batch

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
tr = build_trainer(dev, 0, 1)
batches = [synthetic_batch(32, i, dev) for i in range(4)]
for i in range(6):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
seen = collections.Counter()
def hook(message, category, filename, lineno, file=None, line=None):
    st = [f"{Path(f.filename).name}:{f.lineno} {f.name}" for f in traceback.extract_stack()[:-1] if "amuse_amd" in f.filename]
    seen[(str(message).split("\n")[0][:80], " < ".join(reversed(st[-4:])))] += 1
warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for i in range(3):
    tr.train_step(batches[i % 4])
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("synchronising torch calls in 3 iterations:")
for (msg, st), n in seen.most_common():
    print(f"  {n:3d} x {msg}\n        {st}")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3):
        tr.train_step(batches[i % 4])
    torch.cuda.synchronize()
cnt = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if "Memcpy" in ev.name or "memcpy" in ev.name or "Synchronize" in ev.name:
        cnt[ev.name] += 1
        dur[ev.name] += ev.cpu_time_total
print("runtime copies / synchronisations in 3 iterations (calls, host us in total):")
for k, n in cnt.most_common():
    print(f"  {n:4d} {dur[k]:9.0f}  {k}")
# who issues the blocking ones: the torch op that encloses each hipMemcpyWithStream
ops = [e for e in prof.events() if e.name.startswith("aten::") or "Fn" in e.name or e.name.startswith("Optimizer")]
encl = collections.Counter()
for ev in prof.events():
    if ev.name == "hipMemcpyWithStream":
        t0 = ev.time_range.start
        best = None
        for o in ops:
            if o.time_range.start <= t0 <= o.time_range.end and (best is None or o.time_range.start >= best.time_range.start):
                best = o
        encl[best.name if best else "(no torch op: the library / python side)"] += 1
for k, n in encl.most_common():
    print(f"  hipMemcpyWithStream inside {k}: {n}")
