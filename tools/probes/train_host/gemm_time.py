"""The training step's tall projections out = x W^T (+ b) and dx = dy W: the library's LDS-staged fp32-MFMA kernel (csrc/k_train_gemm.hip) against rocBLAS
(AMUSE_TRAIN_GEMM=vendor, a second process), us per amuse_train_linear_fwd / amuse_train_linear_bwd(dx only) call by HIP events over 200 back-to-back calls, and the
error against float64.  Usage: python tools/probes/train_host/gemm_time.py"""
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
    s = torch.cuda.current_stream().cuda_stream

    def timed(call):
        for _ in range(5):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            call()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 200 * 1e3

    # (rows, N = out features, K = in features) of the step's nn.Linear layers
    for rows, N, K in [(9600, 128, 128), (9664, 128, 128), (9600, 384, 128), (9664, 384, 128), (9600, 512, 128), (9600, 128, 512), (9600, 128, 256), (1216, 128, 128)]:
        g = torch.Generator().manual_seed(rows + N + K)
        x = torch.randn(rows, K, generator=g).to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        dy = torch.randn(rows, N, generator=g).to(dev)
        out = torch.empty(rows, N, device=dev)
        dx = torch.empty(rows, K, device=dev)
        t_f = timed(lambda: _lib.check(lib.amuse_train_linear_fwd(x.data_ptr(), W.data_ptr(), b.data_ptr(), rows, K, N, out.data_ptr(), s)))
        ref = x.double() @ W.double().T + b.double()
        e_f = float((out.double() - ref).abs().max() / ref.abs().max())
        t_n = timed(lambda: _lib.check(lib.amuse_train_linear_fwd(x.data_ptr(), W.data_ptr(), None, rows, K, N, out.data_ptr(), s)))
        t_b = timed(lambda: _lib.check(lib.amuse_train_linear_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), rows, K, N, None, None, dx.data_ptr(), 0, st["ws"].data_ptr(), s)))
        ref = dy.double() @ W.double()
        e_b = float((dx.double() - ref).abs().max() / ref.abs().max())
        print(f"rows={rows} Linear({K} -> {N}): fwd+bias {t_f:6.1f} us  fwd {t_n:6.1f} us (rel err {e_f:.1e})   dx {t_b:6.1f} us (rel err {e_b:.1e})", flush=True)
else:
    for mode in ("own", "vendor", "own", "vendor"):
        env = dict(os.environ, AMUSE_TRAIN_GEMM=mode)
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"--- {mode}"); print(r.stdout.strip() or r.stderr[-1500:], flush=True)
