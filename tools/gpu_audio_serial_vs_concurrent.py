"""Audio front-end at B clips: the three encoders concurrently (amuse_audio_features) vs one after the other (amuse_audio_encode x 3)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import audio_weights as aw
from amuse_amd.audio import AudioEngine

eng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in [int(x) for x in sys.argv[1:]] or [8, 32]:
    w = 0.1 * torch.randn(B, 160000, device="cuda:0")
    fb = eng.fbank(w)
    def run(f, n=3):
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_c = run(lambda: eng.features(w))
    t_s = run(lambda: [eng.encode(n, fb) for n in aw.ENCODERS])
    t_1 = run(lambda: eng.encode(aw.ENCODERS[0], fb))
    print(f"B={B:3d}: concurrent {t_c:8.2f} ms ({t_c / B:.3f} ms/clip)   serial {t_s:8.2f} ms ({t_s / B:.3f} ms/clip)   one encoder {t_1:8.2f} ms", flush=True)
