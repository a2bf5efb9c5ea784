"""Is the train_gesture step capturable as a HIP graph in this torch / ROCm build?  (round 2 found it was not: single gradient elements came back garbage as soon as an
eager allocation ran between two replays - docs/history.md 4.6.)  The probe captures forward + losses + backward (+ the side-stream HIP sampler) of GestureTrainer with
explicit draws (so that a replay must reproduce the eager step's gradient bucket), replays it with eager allocations in between, and times replay against eager.
    python tools/probes/train_host/graph_capture_probe.py [steal|sink|views]"""
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
mode = sys.argv[1] if len(sys.argv) > 1 else "sink"
from amuse_amd import train_gesture as tg  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
for with_sampler in (False, True):
    tr = tg.build_trainer(dev, 0, 1, use_hip_sampler=with_sampler, grads_mode=mode)
    for m in tr.model.values():
        m.eval()                                   # dropout off: the bucket of a replay must equal the eager one
    batch = tg.synthetic_batch(32, 0, dev)
    g = torch.Generator().manual_seed(5)
    ex = dict(noise=torch.randn(32, 1, 128, generator=g).to(dev), timesteps=torch.randint(0, 1000, (32,), generator=g).to(dev),
              eps_enc=torch.randn(1, 32, 128, generator=g).to(dev), eps_inf=torch.randn(1, 32, 128, generator=g).to(dev))

    def fwd_bwd():
        torch.set_grad_enabled(True)
        loss = tr.forward_losses(batch, **ex)
        tr.backward_into_bucket(loss)
        return loss

    for _ in range(3):
        fwd_bwd()
    torch.cuda.synchronize()
    if with_sampler:                               # the sampler's clip counter advances per call: pin it so that eager == capture
        tr.inner_sampler.clip_counter = 0
    loss_ref = float(fwd_bwd())
    ref = tr.flat_grad.clone()
    if with_sampler:
        tr.inner_sampler.clip_counter = 0
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            loss_g = fwd_bwd()
    except Exception as e:  # noqa: BLE001
        print(f"[{mode}, sampler {with_sampler}] capture FAILED: {type(e).__name__}: {str(e)[:300]}")
        continue
    res = []
    for i in range(6):
        tr.flat_grad.fill_(float("nan")) if mode == "steal" else None
        graph.replay()
        torch.cuda.synchronize()
        res.append((bool(torch.equal(tr.flat_grad, ref)), float((tr.flat_grad - ref).abs().max()), float(loss_g)))
        junk = torch.randn(6_000_000, device=dev).mul_(2)        # an eager allocation between replays (what broke round 2's attempt)
        junk2 = [torch.empty(1 << k, device=dev) for k in range(8, 22)]
        del junk, junk2
    print(f"[{mode}, sampler {with_sampler}] eager loss {loss_ref:.6f}; replays (bitwise == eager, max |diff|, loss): {res}")
    t = []
    for fn in (fwd_bwd, graph.replay):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"[{mode}, sampler {with_sampler}] forward + backward: eager {t[0]:.2f} ms, graph replay {t[1]:.2f} ms")
    del graph
