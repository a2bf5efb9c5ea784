"""Identity of a kernel's build inputs: sha256 over the HIP source(s) its object is compiled from, every local header they include (transitively) and the Makefile (flags).
Written next to every committed PMC pass (profiles/rNN_*pmc/kernel_id.json, tools/collect_profiles.sh) and recomputed by bench.py at run time:
`roofline.traffic` is reported from a committed pass only if the sampler that RAN is built from the same inputs as the one that was COUNTED.
  python tools/kernel_id.py k_sample8 [k_sample8x ...]      -> JSON on stdout
"""
import hashlib
import json
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parents[1] / "amuse_amd" / "csrc"
SOURCES = {"k_sample8": ["k_sampler8.hip"], "k_sample8h": ["k_sampler8.hip", "k_sampler8h.hip"], "k_sample8x": ["k_sampler8x.hip"],
           "k_sample": ["k_sampler.hip"], "k_vae_fused": ["k_vae_fused.hip"], "k_den_fused": ["k_den_fused.hip"], "k_vae_rows8x": ["k_vae_rows8.hip"], "k_vae_fusedx": ["k_vae_fusedx.hip"]}


def _closure(files):
    """The named sources plus every local header they #include, transitively, plus the Makefile (compiler flags)."""
    import re
    seen, todo = [], list(files)
    while todo:
        f = todo.pop(0)
        if f in seen or not (CSRC / f).exists():
            continue
        seen.append(f)
        todo += re.findall(r'^\s*#include\s+"([^"]+)"', (CSRC / f).read_text(), flags=re.M)
    return seen + ["Makefile"]


def kernel_id(name: str) -> dict:
    h = hashlib.sha256()
    files = _closure(SOURCES[name])
    for f in files:
        h.update(f.encode() + b"\0" + (CSRC / f).read_bytes() + b"\0")
    out = {"kernel": name, "inputs": files, "source_sha256": h.hexdigest()}
    obj = CSRC / (SOURCES[name][-1].replace(".hip", ".o"))
    if obj.exists():
        out["object_sha256"] = hashlib.sha256(obj.read_bytes()).hexdigest()     # informative: same toolchain + same inputs -> same object
    return out


if __name__ == "__main__":
    print(json.dumps({n: kernel_id(n) for n in (sys.argv[1:] or list(SOURCES))}, indent=1))
