"""Forward attention time (B = 32, S = 300, p = 0.1) across libamuse_hip*.so variants (-DAMUSE_ATTN_ABL timing ablations), each in its own process."""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import train_ops as T
    B, S = 32, 300
    qkv = torch.randn(B * S, 384, device="cuda:0")
    o = torch.empty(B * S, 128, device="cuda:0"); lse = torch.empty(B, 4, S, device="cuda:0")
    lib = T._st(qkv.device)["lib"]
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.amuse_train_attn_fwd(qkv.data_ptr(), B, S, 0.1, 1, 2, o.data_ptr(), lse.data_ptr(), None, st)
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 200 * 1e3:.1f} us per forward call (200 back-to-back launches, raw ctypes)")
else:
    for lib in sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so"))):
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, AMUSE_HIP_LIB=lib), capture_output=True, text=True)
        print(os.path.basename(lib).ljust(30), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
