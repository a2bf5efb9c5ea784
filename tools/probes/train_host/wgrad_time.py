"""The training step's weight-gradient GEMM dW = dy^T x (amuse_train_linear_bwd with only dW asked for): the library's chunked fp32-MFMA kernel (csrc/k_train.hip
k_train_wgrad + k_train_wgrad_sum) against rocBLAS (AMUSE_TRAIN_WGRAD=vendor, a second process), us per call by HIP events over 200 back-to-back calls, and the
error against float64.  Usage: python tools/probes/train_host/wgrad_time.py"""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[3]
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, str(REPO))
    from amuse_amd import train_ops, _lib
    dev = torch.device("cuda", 0)
    st = train_ops._st(dev)
    lib = st["lib"]
    for rows, N, K in [(9664, 128, 128), (9664, 512, 128), (9664, 128, 512), (9664, 384, 128), (9600, 128, 256), (9600, 128, 128), (160, 512, 128)]:
        g = torch.Generator().manual_seed(rows + N)
        dy = torch.randn(rows, N, generator=g).to(dev)
        x = torch.randn(rows, K, generator=g).to(dev)
        W = torch.zeros(N, K, device=dev)
        dW = torch.empty(N, K, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        call = lambda: _lib.check(lib.amuse_train_linear_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), rows, K, N, dW.data_ptr(), None, None, 0, st["ws"].data_ptr(), s))
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            call()
        e1.record(); torch.cuda.synchronize()
        ref = dy.double().T @ x.double()
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        print(f"rows={rows} dW[{N}x{K}]: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us  rel err {err:.1e}", flush=True)
else:
    for mode in ("own", "vendor", "own", "vendor"):
        env = dict(os.environ)
        if mode == "vendor":
            env["AMUSE_TRAIN_WGRAD"] = "vendor"
        else:
            env.pop("AMUSE_TRAIN_WGRAD", None)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"--- {mode}"); print(r.stdout.strip() or r.stderr[-600:], flush=True)
