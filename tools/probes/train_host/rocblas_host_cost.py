"""Host cost of enqueueing one fp32 GEMM of the training step's shapes: torch.mm / torch.addmm against rocblas_sgemm called directly (ctypes, the copy of
librocblas torch has loaded).  Usage: python tools/probes/train_host/rocblas_host_cost.py"""
import ctypes as C
import time

import torch

dev = "cuda:0"
M, K, N = 9664, 128, 384
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
torch.mm(a, w.t())
lib = C.CDLL("librocblas.so")
h = C.c_void_p()
assert lib.rocblas_create_handle(C.byref(h)) == 0
assert lib.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
one, zero = C.c_float(1.0), C.c_float(0.0)
lib.rocblas_sgemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
OP_N, OP_T = 111, 112


def sgemm_nt():   # out[M][N] = a[M][K] . w[N][K]^T
    return lib.rocblas_sgemm(h, OP_T, OP_N, N, M, K, C.byref(one), w.data_ptr(), K, a.data_ptr(), K, C.byref(zero), out.data_ptr(), N)


assert sgemm_nt() == 0
torch.cuda.synchronize()
print("max diff vs torch.mm:", float((out - torch.mm(a, w.t())).abs().max()))
for name, fn in (("torch.mm", lambda: torch.mm(a, w.t())), ("torch.addmm", lambda: torch.addmm(bias, a, w.t())), ("rocblas_sgemm (ctypes)", sgemm_nt)):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:26s} host enqueue {1e6 * (t1 - t0) / 300:6.1f} us per call; with the device {1e6 * (t2 - t0) / 300:6.1f} us per call")
