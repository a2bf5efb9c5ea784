"""CPU: the gfx950 ISA of the hot kernels, as hipcc emits it from the committed sources (cross-compiles without a GPU): register
budget and spills.  A kernel that starts spilling after an edit still passes every parity test - this is where it shows."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

CSRC = Path(__file__).resolve().parents[1] / "amuse_amd" / "csrc"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _kernels(src, extra=()):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", *extra, "-S", "--cuda-device-only", "-o", "-", src],
                         cwd=CSRC, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    ks = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", out.stdout):
        ks[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    assert ks, "no kernel metadata found"
    return ks


@pytest.mark.skipif(not Path(HIPCC).exists(), reason="hipcc not installed")
@pytest.mark.parametrize("src,extra,max_spills", [
    ("k_sampler8.hip", (), 0),            # bf16 throughput sampler (two waves per SIMD: 256 registers each)
    ("k_sampler8h.hip", (), 0),           # its fp16 build
    ("k_sampler8x.hip", (), 8),           # fp32x parity sampler (two 64-bit values parked once per launch; twice that in the profiling instantiation)
    ("k_audio_gemm.hip", (), 0),          # every instantiation of the audio GEMM
    ("k_vae_rows8.hip", (), 0),           # fp32x row stages without split-K (158 registers: three waves per SIMD)
    ("k_vae_fused.hip", ("-fno-honor-nans",), 56),   # fused decoder: cold per-block / per-tile values only (150 before round 3; 21 in the product instantiation, 53 in the tapped one)
    ("k_den_fused.hip", ("-fno-honor-nans",), 32),   # fused pose-space denoiser step (26)
    ("k_train.hip", (), 0),               # training-step glue kernels
    ("k_train_attn.hip", (), 0),          # training-step attention forward / backward (fp32 MFMA)
])
def test_register_budget_and_spills(src, extra, max_spills):
    for name, (vgprs, spills) in _kernels(src, extra).items():
        # (k_train_wgrad<false> keeps all 96 operand loads of its chunk in flight - 192 registers of operands - and runs one wave per SIMD by design: launches of up to
        # ~1,000 waves; its streaming instantiation for larger launches fits three waves per SIMD)
        one_wave = src == "k_train.hip" and "k_train_wgradILb0E" in name
        assert vgprs <= (288 if one_wave else 256), (name, vgprs)
        # (the phase-stamp instantiation of the 8-wave samplers - template argument PROF = true - parks its 64-bit stamp pointer: 2 registers)
        prof = src.startswith("k_sampler8") and "ILb1E" in name
        # (the encoder instantiations of the fp32x row kernel park 14 registers once per launch, outside its stage loops)
        enc8 = src == "k_vae_rows8.hip" and "Lb1E" in name
        assert spills <= max(max_spills, 2 if prof else 0, 16 if enc8 else 0), (name, spills)
