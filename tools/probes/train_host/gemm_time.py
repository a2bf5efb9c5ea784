"""The training step's GEMMs on the library's own fp32-MFMA kernels (csrc/k_train_gemm.hip: k_train_gemm_tall for the tall projections, k_train_gemm_any for every other
shape; csrc/k_train.hip k_train_wgrad for the 128-wide weight gradients) against torch's (vendor) GEMM on the same operands: us per call by HIP events over 200 back-to-back
calls, and the error of both against float64.  Usage: python tools/probes/train_host/gemm_time.py"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(REPO))
from amuse_amd import train_ops, _lib  # noqa: E402

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
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3


rel = lambda a, ref: float((a.double() - ref).abs().max() / ref.abs().max())
print("# (rows, N = out features, K = in features) of the step's nn.Linear layers; own = the library's kernels, torch = torch.addmm / mm (the vendor library)")
# tall projections | the 333-wide embedding / output layers | the Denoiser's 160-row layers | the 32-row condition / memory projections
for rows, N, K in [(9600, 128, 128), (9664, 384, 128), (9600, 512, 128), (9600, 128, 512), (9600, 128, 256), (9600, 128, 333), (9600, 333, 128), (160, 384, 128),
                   (160, 128, 128), (160, 512, 128), (160, 128, 512), (32, 128, 256), (32, 128, 128)]:
    g = torch.Generator().manual_seed(rows + N + K)
    x = torch.randn(rows, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(rows, N, generator=g).to(dev)
    out, dx, dW = torch.empty(rows, N, device=dev), torch.empty(rows, K, device=dev), torch.empty(N, K, device=dev)
    t_f = timed(lambda: _lib.check(lib.amuse_train_linear_fwd(x.data_ptr(), W.data_ptr(), b.data_ptr(), rows, K, N, out.data_ptr(), s)))
    ref = x.double() @ W.double().T + b.double()
    e_f = rel(out, ref)
    v_f = timed(lambda: torch.addmm(b, x, W.T, out=out))
    ev_f = rel(out, ref)
    t_b = timed(lambda: _lib.check(lib.amuse_train_linear_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), rows, K, N, None, None, dx.data_ptr(), 0, st["ws"].data_ptr(), s)))
    ref = dy.double() @ W.double()
    e_b = rel(dx, ref)
    v_b = timed(lambda: torch.mm(dy, W, out=dx))
    t_w = timed(lambda: _lib.check(lib.amuse_train_linear_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), rows, K, N, dW.data_ptr(), None, None, 0, st["ws"].data_ptr(), s)))
    ref = dy.double().T @ x.double()
    e_w = rel(dW, ref)
    v_w = timed(lambda: torch.mm(dy.T, x, out=dW))
    print(f"rows={rows:5d} Linear({K:3d} -> {N:3d}): fwd+bias own {t_f:6.1f} us (err {e_f:.1e}) torch {v_f:6.1f} (err {ev_f:.1e}) | dx own {t_b:6.1f} (err {e_b:.1e}) torch {v_b:6.1f} | "
          f"dW own {t_w:6.1f} (err {e_w:.1e}) torch {v_w:6.1f}", flush=True)
