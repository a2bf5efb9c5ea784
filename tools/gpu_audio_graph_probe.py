"""Is the audio front-end launch-bound at small batches?  amuse_audio_features (fbank + 3 x AST on three streams, ~250 launches) eagerly against the same call captured
into a HIP graph (torch.cuda.graph around the library call: the library's fork / join is event-ordered, hence capturable) and replayed.  python tools/gpu_audio_graph_probe.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import audio_weights as aw
from amuse_amd.audio import AudioEngine
eng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS), device="cuda:0")
g = torch.Generator().manual_seed(0)
for B in (1, 2, 4, 8, 16):
    w = (0.1 * torch.randn(B, 160000, generator=g)).cuda()
    ref = eng.features(w)
    torch.cuda.synchronize()
    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3
    t_eager = timeit(lambda: eng.features(w))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            eng.features(w)
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            out = eng.features(w)
        t_graph = timeit(gr.replay)
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print(f"B={B:2d}: eager {t_eager:7.3f} ms   graph replay {t_graph:7.3f} ms   ({t_eager / t_graph:.2f} x)   bitwise equal: {same}", flush=True)
    except Exception as e:
        print(f"B={B}: eager {t_eager:.3f} ms; capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
