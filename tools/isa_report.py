"""Register / spill / LDS / scratch figures of every kernel in amuse_amd/csrc, from the gfx950 ISA hipcc emits for the committed sources
(cross-compiles without a GPU).  Usage: python tools/isa_report.py > profiles/rNN_isa_resources.txt"""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parents[1] / "amuse_amd" / "csrc"
EXTRA = {"k_vae_fused.hip": ["-fno-honor-nans"], "k_vae_fusedh.hip": ["-fno-honor-nans"],
         "k_audio.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"]}   # as in the Makefile
print(f"{'kernel':70s} {'VGPR':>5s} {'spill':>6s} {'SGPR':>5s} {'LDS static':>11s} {'scratch B/lane':>15s} {'v_mfma':>7s}")
for src in sorted(CSRC.glob("k_*.hip")):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", *EXTRA.get(src.name, []), "-S",
                          "--cuda-device-only", "-o", "-", src.name], cwd=CSRC, capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stderr[-2000:])
    mfma = {}
    cur = None
    for line in out.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        elif cur and "v_mfma" in line:
            mfma[cur] = mfma.get(cur, 0) + 1
    print(f"# {src.name}")
    for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?"
                         r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", out.stdout):
        lds, name, scratch, sgpr, vgpr, spill = m.groups()
        nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        nice = re.sub(r"amuse::\(anonymous namespace\)::", "", nice)
        nice = re.sub(r"\(.*\)$", "", nice).replace("void ", "")
        print(f"{nice[:70]:70s} {vgpr:>5s} {spill:>6s} {sgpr:>5s} {lds:>11s} {scratch:>15s} {mfma.get(name, 0):>7d}")
