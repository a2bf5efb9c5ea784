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


# kernel name -> what its (mangled) device symbol contains.  The product instantiation only: k_sample8<false> (no phase stamps), etc.
SYMBOLS = {"k_sample8": "9k_sample8ILb0E", "k_sample8h": "10k_sample8hILb0E", "k_sample8x": "10k_sample8xILb0E", "k_sample": "8k_sampleILi0ELb0E", "k_vae_fused": "11k_vae_fusedILb0ELb0E",
           "k_den_fused": "11k_den_fusedILb0E", "k_vae_rows8x": "12k_vae_rows8x", "k_vae_fusedx": "12k_vae_fusedx"}
LIB = CSRC.parent / "libamuse_hip.so"


def device_text_sha256(lib_path, symbol_part: str):
    """sha256 over the machine code of every gfx950 device function of `lib_path` whose symbol contains `symbol_part` - read out of the library that is LOADED, not out
    of an object file that may or may not be what was linked: the offload bundles (__CLANG_OFFLOAD_BUNDLE__, uncompressed) of the .so hold one ELF code object per
    translation unit; its symbol table gives address and size of each kernel's text.  None if the library or the symbol is not there."""
    import struct
    try:
        d = Path(lib_path).read_bytes()
    except OSError:
        return None
    found = []
    i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while i >= 0:
        n = struct.unpack_from("<Q", d, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, ts = struct.unpack_from("<QQQ", d, p)
            p += 24
            triple = d[p:p + ts]
            p += ts
            if b"gfx950" not in triple or size == 0:
                continue
            e = d[i + off:i + off + size]
            if e[:4] != b"\x7fELF":
                continue
            shoff, shentsize, shnum = struct.unpack_from("<Q", e, 0x28)[0], struct.unpack_from("<H", e, 0x3A)[0], struct.unpack_from("<H", e, 0x3C)[0]
            secs = [struct.unpack_from("<IIQQQQIIQQ", e, shoff + k * shentsize) for k in range(shnum)]
            for sec in secs:
                if sec[1] != 2:          # SHT_SYMTAB
                    continue
                strtab = secs[sec[6]]
                for k in range(sec[5] // 24):
                    st_name, st_info, _o, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", e, sec[4] + 24 * k)
                    if (st_info & 15) != 2 or st_size == 0 or st_shndx >= shnum:      # STT_FUNC
                        continue
                    nm = e[strtab[4] + st_name:e.index(b"\0", strtab[4] + st_name)]
                    if symbol_part.encode() in nm and not nm.endswith(b".kd"):
                        text = secs[st_shndx]
                        a = text[4] + (st_value - text[3])
                        found.append((nm, e[a:a + st_size]))
        i = d.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 1)
    if not found:
        return None
    h = hashlib.sha256()
    for nm, code in sorted(found):
        h.update(nm + b"\0" + code)
    return h.hexdigest()


def kernel_id(name: str) -> dict:
    h = hashlib.sha256()
    files = _closure(SOURCES[name])
    for f in files:
        h.update(f.encode() + b"\0" + (CSRC / f).read_bytes() + b"\0")
    out = {"kernel": name, "inputs": files, "source_sha256": h.hexdigest()}
    # the identity that counts: the machine code of the kernel inside the library that gets LOADED (same toolchain + same inputs -> same text; an edit to a shared
    # header that leaves the kernel's code as it was does not change it; a stale .o on disk cannot fake it)
    code = device_text_sha256(LIB, SYMBOLS[name])
    if code:
        out["object_sha256"] = code
        out["object_is"] = f"device text of *{SYMBOLS[name]}* in amuse_amd/libamuse_hip.so"
    return out


if __name__ == "__main__":
    print(json.dumps({n: kernel_id(n) for n in (sys.argv[1:] or list(SOURCES))}, indent=1))
