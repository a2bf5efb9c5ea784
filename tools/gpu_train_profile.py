"""GPU: where one train_gesture iteration (BASELINE config 4, batch 32) spends its time - torch profiler over 3 steady-state
iterations: wall clock per iteration, summed device-kernel time, kernel launches, and the operators with the largest
self CPU / device time.  Usage: python tools/gpu_train_profile.py [batch] [out.txt]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from amuse_amd.train_gesture import build_trainer, synthetic_batch  # noqa: E402


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    dev = torch.device("cuda:0")
    tr = build_trainer(dev)
    batch = synthetic_batch(bs, 1, dev)
    for _ in range(8):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(batch)
    torch.cuda.synchronize()
    print(f"eager: {(time.perf_counter() - t0) * 100:.2f} ms per iteration at batch {bs}", file=out)

    # phases by wall clock with a sync after each (serialises, so the sum is an upper bound)
    def timed(fn):
        torch.cuda.synchronize()
        a = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - a) * 1e3
    acc = {"forward+losses": 0.0, "backward": 0.0, "optimizer": 0.0}
    for _ in range(5):
        loss, ms = timed(lambda: tr.forward_losses(batch))
        acc["forward+losses"] += ms
        tr.flat_grad.zero_()
        _, ms = timed(loss.backward)
        acc["backward"] += ms
        _, ms = timed(tr.lpdm_opt.step)
        acc["optimizer"] += ms
    print("phases (ms, synced): " + ", ".join(f"{k} {v / 5:.2f}" for k, v in acc.items()), file=out)

    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
    ev = prof.key_averages()
    dev_us = sum(e.self_device_time_total for e in ev if e.device_type == torch.autograd.DeviceType.CUDA)   # kernel rows only: operator rows repeat them
    launches = sum(e.count for e in ev if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipModuleLaunchKernel",
                                                    "hipExtLaunchKernel", "cudaLaunchKernel"))
    print(f"profiled 3 iterations: device kernel time {dev_us / 3e3:.2f} ms per iteration, {launches // 3} launches per iteration", file=out)
    print(ev.table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=60), file=out)
    print(ev.table(sort_by="self_device_time_total", row_limit=25, max_name_column_width=60), file=out)


if __name__ == "__main__":
    main()
