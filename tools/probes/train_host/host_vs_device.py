"""Is the training step host- or device-bound?  Host: wall time of the dispatch loop alone (no synchronisation inside, the device's queue drained before it starts
and the loop short enough not to fill it); device: the same iterations' kernel time with the queue kept full (HIP events around the loop after a run-ahead of 10).
Usage: python tools/probes/train_host/host_vs_device.py [graph]      (graph: the iteration as two HIP graphs, GestureTrainer.enable_graph)"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from amuse_amd.train_gesture import build_trainer, synthetic_batch

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
tr = build_trainer(dev, 0, 1)
batches = [synthetic_batch(32, i, dev) for i in range(4)]
for i in range(15):
    tr.train_step(batches[i % 4])
if len(sys.argv) > 1 and sys.argv[1] == "graph":
    print("graph mode:", tr.enable_graph(batches[0]), flush=True)
    for i in range(4):
        tr.train_step(batches[i % 4])
torch.cuda.synchronize()
# the host's own cost of one iteration: an idle device, one train_step, the time until the call returns (nothing to wait behind), then drain
ret = []
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(batches[i % 4])
    ret.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
ret.sort()
print(f"one iteration on an idle device: train_step returns after {ret[len(ret) // 2]:.2f} ms (median of 12; min {ret[0]:.2f}, max {ret[-1]:.2f})", flush=True)
for rnd in range(3):
    n = 40
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        tr.train_step(batches[i % 4])
    t_host = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"round {rnd}: dispatch loop {t_host / n * 1e3:.2f} ms per iteration, until the device is done {t_all / n * 1e3:.2f} ms, "
          f"device span (events) {e0.elapsed_time(e1) / n:.2f} ms; queued behind the host at loop end: {(t_all - t_host) * 1e3:.1f} ms", flush=True)
# device-only figure: a big sleep kernel in front lets the host run ahead, the events then time kernels that never wait for dispatch
spin = torch.empty(1 << 28, device=dev)
for rnd in range(2):
    n = 12
    torch.cuda.synchronize()
    for _ in range(40):
        spin.mul_(1.0001)            # ~40 x 0.3 ms of queued work for the host to hide behind
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        tr.train_step(batches[i % 4])
    e1.record()
    torch.cuda.synchronize()
    print(f"device-only round {rnd}: {e0.elapsed_time(e1) / n:.2f} ms per iteration (host ran ahead of the queue for the first iterations)", flush=True)
