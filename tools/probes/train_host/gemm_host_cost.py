"""Host cost of enqueueing one tall projection through the library (amuse_train_linear_fwd): own kernel against rocBLAS (AMUSE_TRAIN_GEMM=vendor), us per call of a
300-call enqueue loop with an idle queue in front.  Usage: python tools/probes/train_host/gemm_host_cost.py"""
import os, subprocess, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[3]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, str(REPO))
    from amuse_amd import train_ops, _lib
    dev = torch.device("cuda", 0)
    st = train_ops._st(dev)
    lib = st["lib"]
    s = torch.cuda.current_stream().cuda_stream
    for rows, N, K, bias in [(1024, 128, 128, False), (1024, 128, 128, True), (9600, 128, 128, False), (9600, 384, 128, True)]:
        x = torch.randn(rows, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); out = torch.empty(rows, N, device=dev)
        call = lambda: lib.amuse_train_linear_fwd(x.data_ptr(), W.data_ptr(), b.data_ptr() if bias else None, rows, K, N, out.data_ptr(), s)
        for _ in range(50):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            call()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"rows={rows} Linear({K} -> {N}){' + bias' if bias else ''}: host enqueue {1e6 * (t1 - t0) / 300:6.1f} us per call; with the device {1e6 * (t2 - t0) / 300:6.1f} us", flush=True)
else:
    for mode in ("own", "vendor", "own", "vendor"):
        env = dict(os.environ)
        if mode == "vendor":
            env["AMUSE_TRAIN_GEMM"] = "vendor"
        else:
            env.pop("AMUSE_TRAIN_GEMM", None)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"--- {mode}"); print(r.stdout.strip() or r.stderr[-1500:], flush=True)
