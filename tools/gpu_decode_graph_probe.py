"""Are the small-batch paths launch-bound?  The staged decode (19 launches; one clip) and the single-clip DDIM-50 job (sampler + decode) eagerly against the same calls
captured into a HIP graph (torch.cuda.graph around the library call) and replayed.  python tools/gpu_decode_graph_probe.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts, scheduler as sch
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_schedule(sch.ddim_table())
g = torch.Generator().manual_seed(0)


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for prec in ("bf16", "fp32x"):
    for B in (1, 8, 32):
        z = torch.randn(B, 128, generator=g).cuda()
        c, e, s_ = (torch.randn(B, 256, generator=g).cuda() for _ in range(3))
        for name, fn in (("decode", lambda: eng.vae_decode(z, None, prec)), ("ddim50 job", lambda: eng.diffusion_backward(c, e, s_, prec, seed=1))):
            ref = fn()
            t_eager = timeit(fn)
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(st):
                    fn()
                torch.cuda.current_stream().wait_stream(st)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    out = fn()
                t_graph = timeit(gr.replay)
                same = torch.equal(out["poses"], ref["poses"])
                print(f"{prec:5s} B={B:2d} {name:10s}: eager {t_eager:7.3f} ms   graph replay {t_graph:7.3f} ms   ({t_eager / t_graph:.2f} x)   bitwise equal: {same}", flush=True)
            except Exception as ex:
                print(f"{prec} B={B} {name}: eager {t_eager:.3f} ms; capture failed: {type(ex).__name__}: {str(ex)[:160]}", flush=True)
