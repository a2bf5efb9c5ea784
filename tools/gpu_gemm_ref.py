"""What the vendor GEMM (torch.matmul -> hipBLASLt) reaches on the audio front-end's shapes: the practical ceiling
k_gemm_bf16 is measured against (tools, not product)."""
import torch
M = 32 * 1214
for (N, K) in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        (a @ w.T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        (a @ w.T)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={M} N={N} K={K}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s")
