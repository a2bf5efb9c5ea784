"""GPU: ms per train_gesture iteration (batch 32, one GPU) under the trainer's switches, each variant in its own process:
  AMUSE_TRAIN_OPT   foreach | fused     AdamW implementation
  AMUSE_TRAIN_GRADS views | steal       gradients accumulated into the bucket's views / handed over and packed by one copy
  AMUSE_TRAIN_BLAS  default | rocblas   torch.backends.cuda.preferred_blas_library (set by the trainer)
  tunable           0 | 1               torch.cuda.tunable (TunableOp) picks the GEMM solutions by measurement
  addmm_lt          1 | 0               0 = DISABLE_ADDMM_CUDA_LT=1 (torch's biased GEMMs without the hipBLASLt epilogue path)
Usage: python tools/gpu_train_variants.py [out.txt]        (worker: ... --worker opt grads blas tunable addmm_lt)"""
import os
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def worker(opt, grads, blas, tunable, lt="1"):
    if lt == "0":
        os.environ["DISABLE_ADDMM_CUDA_LT"] = "1"      # torch's addmm: gemm with beta = 1 on a bias-filled output instead of the Lt epilogue
    os.environ["AMUSE_TRAIN_OPT"], os.environ["AMUSE_TRAIN_GRADS"], os.environ["AMUSE_TRAIN_BLAS"] = opt, grads, blas
    import torch
    sys.path.insert(0, str(REPO))
    from amuse_amd.train_gesture import build_trainer, synthetic_batch
    if tunable == "1":
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.set_filename(str(REPO / "gpurun_out" / "tunableop_train.csv"))
    dev = torch.device("cuda:0")
    tr = build_trainer(dev)
    batch = synthetic_batch(32, 1, dev)
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(batch)
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            loss = tr.train_step(batch)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print(f"opt={opt:8s} grads={grads:6s} blas={blas:8s} tunable={tunable} addmm_lt={lt}: {best:6.2f} ms per iteration  (loss {float(loss):.5f}, "
          f"10 warm-up iterations {warm:.1f} s)", flush=True)


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    variants = [("foreach", "views", "default", "0", "1"), ("fused", "views", "default", "0", "1"), ("foreach", "steal", "default", "0", "1"),
                ("fused", "steal", "default", "0", "1"), ("fused", "steal", "rocblas", "0", "1"), ("fused", "steal", "rocblas", "0", "0"),
                ("fused", "steal", "default", "1", "1"), ("fused", "steal", "rocblas", "0", "1"), ("fused", "steal", "rocblas", "0", "0")]
    for v in variants:
        r = subprocess.run([sys.executable, __file__, "--worker", *v], capture_output=True, text=True, timeout=1500)
        line = (r.stdout.strip().splitlines() or [f"{v}: failed rc={r.returncode} {r.stderr[-400:]}"])[-1]
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(*sys.argv[2:7])
    else:
        main()
