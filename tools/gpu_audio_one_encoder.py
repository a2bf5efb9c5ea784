"""One AST encoder over B clips, nothing concurrent: the workload for per-kernel rocprofv3 statistics of the audio front-end
(rocprofv3 --kernel-trace --stats -- python3 tools/gpu_audio_one_encoder.py 32)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import audio_weights as aw
from amuse_amd.audio import AudioEngine

eng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fb = eng.fbank(0.1 * torch.randn(B, 160000, device="cuda:0"))
for _ in range(5):
    eng.encode(aw.ENCODERS[0], fb)
torch.cuda.synchronize()
