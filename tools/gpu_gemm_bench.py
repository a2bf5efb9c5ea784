"""k_gemm_tm (k_audio_gemm.hip) in isolation against torch.matmul (hipBLASLt): correctness and TFLOP/s on the audio front-end's
shapes; operands tile-major (amuse_debug_tile), bf16 tile-major output."""
import ctypes as C
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import _lib
lib = _lib.load()
M = 32 * 1214
Mp = (M + 127) // 128 * 128
p = lambda t: C.c_void_p(t.data_ptr())


def pack_w(w):
    """[N, K] -> the kernel's fragment order: [span][x = (p, q)][k-step][lane = (g, i = (a, b))][e], feature = 64 span +
    32 p + 8 a + 4 q + b, k = 32 ks + 8 g + e  (amuse_audio_api.hip pack_w)."""
    N, K = w.shape
    v = w.view(N // 64, 2, 4, 2, 4, K // 32, 4, 8)          # span, p, a, q, b, ks, g, e
    return v.permute(0, 1, 3, 5, 6, 2, 4, 7).contiguous().view(-1)

for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    at = torch.empty(Mp, K, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.amuse_debug_tile(p(a), p(at), M, K, 0, None))
    w = (0.05 * torch.randn(N, K, device="cuda")).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out_t = torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wpk = pack_w(w)
    _lib.check(lib.amuse_debug_gemm(p(at), p(wpk), p(bias), M, N, K, 0, p(out_t), None))
    _lib.check(lib.amuse_debug_tile(p(out_t), p(out), M, N, 1, None))
    ref = (a.float() @ w.float().T + bias)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        lib.amuse_debug_gemm(p(at), p(wpk), p(bias), M, N, K, 0, p(out_t), None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.amuse_debug_gemm(p(at), p(wpk), p(bias), M, N, K, 0, p(out_t), None)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    e0.record()
    for _ in range(20):
        (a @ w.T)
    e1.record()
    torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / 20
    print(f"N={N:5d} K={K:5d}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TFLOP/s   (torch {ms_t*1e3:7.1f} us {2*M*N*K/ms_t/1e9:7.1f})   rel err {err:.2e}", flush=True)
