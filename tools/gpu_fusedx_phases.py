"""Phase timeline of the fused fp32x decoder (libamuse_hip built with -DAMUSE_FPROF=1 for k_vae_fusedx.hip: tools/build_variant.sh fxprof k_vae_fusedx.hip "-DAMUSE_FPROF=1 -fno-honor-nans"):
AMUSE_HIP_LIB=amuse_amd/libamuse_hip_fxprof.so python tools/gpu_fusedx_phases.py 2>&1 | python tools/gpu_fusedx_phases.py --sum"""
import sys
if '--sum' in sys.argv:
    import collections
    tot, cnt, order = collections.Counter(), collections.Counter(), []
    for l in sys.stdin:
        if l.startswith('FPROF'):
            f = l.split()
            tag, d = int(f[3]), int(f[4][1:])
            tot[tag] += d; cnt[tag] += 1
    names = {1: 'block start (from the previous stamp)', 2: 'skip linear', 3: 'k, v MFMAs', 4: 'K / V images written', 5: 'barrier A', 6: 'q', 7: 'attention + o stores', 8: 'barrier B', 9: 'out_proj stage (incl. barrier)', 10: 'norm1 + ca + norm2', 11: 'linear1 stage', 12: 'barrier', 13: 'GELU + split', 14: 'linear2 stage', 15: 'barrier', 16: 'norm3'}
    s = sum(tot.values())
    for t in sorted(tot):
        print(f'tag {t:2d} {names.get(t, ""):40s} {cnt[t]:4d} x {tot[t] / cnt[t]:9.0f} = {tot[t]:9d} cycles  {100 * tot[t] / s:5.1f} %')
    print('two blocks (1 and 6):', s, 'cycles')
    sys.exit(0)
import torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
eng.set_decode_path("clip")
z = torch.randn(256, 128, generator=torch.Generator().manual_seed(1)).cuda()
for i in range(4):
    eng.vae_decode(z, None, "fp32x")
torch.cuda.synchronize()
