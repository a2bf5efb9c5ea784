"""Per-score instruction budget of the fused kernels' S ~ 300 self-attention (k_vae_fused / k_den_fused: amuse_fused.hpp `attend`), read off the ISA hipcc
emits: the smallest loop that holds the softmax exponentials is `attend` for ONE 16-query tile of ONE head against all 320 key slots (five 64-key chunks,
unrolled) - 80 scores per lane (76 exponentials: the key tile beyond every sequence is skipped).  Counts every instruction of that loop body by issue class and
prices it with the measured issue costs of MI355X_MICROARCH.md / tools/probes/valu_rate_probe.hip (profiles/r03_valu_rate_probe.txt).
   python tools/attention_isa_budget.py [k_vae_fused.hip | k_den_fused.hip]   (cross-compiles for gfx950; no GPU needed)
"""
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

CSRC = Path(__file__).resolve().parents[1] / "amuse_amd" / "csrc"


def isa(src):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-honor-nans",
                        "-save-temps=obj", "-c", str(CSRC / src), "-o", f"{d}/o.o"], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
        return next(Path(d).glob("*gfx950.s")).read_text()


def classify(m):
    if m.startswith("v_mfma"):
        return "MFMA"
    if m.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")):
        return "VALU transcendental (v_exp_f32 ...)"
    if m.startswith(("v_max3", "v_max_", "v_min")):
        return "VALU max (v_max3_f32 / v_max_f32)"
    if m.startswith("v_cvt"):
        return "VALU convert / pack (v_cvt_pk_*)"
    if m.startswith("v_pk_"):
        return "VALU packed f32 (v_pk_mul / v_pk_add)"
    if "dpp" in m or m.startswith(("v_permlane", "v_readlane", "v_readfirstlane")):
        return "VALU cross-lane (DPP / permlane)"
    if m.startswith(("v_cndmask", "v_cmp")):
        return "VALU compare / select (mask)"
    if m.startswith("v_accvgpr"):
        return "VALU accvgpr moves"
    if m.startswith("v_"):
        return "VALU other (sub / mul / fma / mov / address)"
    if m.startswith("ds_"):
        return "LDS (ds_read_b128 ...)"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM / scratch"
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_nop"):
        return "s_nop (hazard wait states)"
    return "SALU / branch / barrier"


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "k_vae_fused.hip"
    text = isa(src)
    best = None
    for km in re.finditer(r"^(_ZN\S+):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
        name, body = km.group(1), km.group(2).split("\n")
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        for i, l in enumerate(body):
            mb = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if not mb or labels.get(mb.group(1), 1 << 30) >= i:
                continue                                      # (a back edge: its target lies above it)
            seg = body[labels[mb.group(1)]:i + 1]
            ops = [s.split()[0] for s in seg if s.startswith("\t") and not s.strip().startswith((";", "."))]
            nexp = sum(o.startswith("v_exp") for o in ops)
            if nexp >= 60 and (best is None or len(ops) < len(best[2])):
                best = (name, nexp, ops)
    if best is None:
        raise SystemExit("no loop with >= 60 exponentials found")
    name, nexp, ops = best
    c = Counter(classify(o) for o in ops)
    scores = 80.0
    nm = c["MFMA"]
    print(f"# {src}: the `attend` loop body (one 16-query tile x one head x 320 key slots) inside {name[:70]}")
    print(f"# {len(ops)} instructions per trip, {nexp} exponentials, {nm} MFMAs (80 useful: 40 score + 40 PV; the rest: row sums on the ones fragment, and "
          f"whatever else of the tile loop hipcc placed inside)")
    print(f"{'class':52s} {'per trip':>9s} {'per score':>10s}")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
        print(f"{k:52s} {v:9d} {v / scores:10.2f}")
    valu_plain = sum(v for k, v in c.items() if k.startswith("VALU") and "transcendental" not in k)
    trans = c["VALU transcendental (v_exp_f32 ...)"]
    useful = 80 * 16.0
    t_mfma = 16.0 * nm
    t_valu_alone = 2.1 * valu_plain + 5.4 * trans          # a wave alone on its SIMD: 2.1 cycles per plain VALU, 5.4 per v_exp_f32
    t_valu_beside = 8.2 * valu_plain + 16.2 * trans        # beside a wave that keeps the matrix pipe busy: 8.2 / 16.2
    print(f"\n# cycles of one SIMD per trip: matrix pipe {t_mfma:.0f} (useful MFMAs {useful:.0f}); VALU issue {t_valu_alone:.0f} with the SIMD to itself, "
          f"{t_valu_beside:.0f} at the rates measured beside a busy matrix pipe")
    print(f"# useful-MFMA fraction of the attention if ... the two pipes overlapped perfectly: {useful / max(t_mfma, t_valu_alone):.2f}; "
          f"ran in series (two lock-step waves per SIMD - the regime the counters show): {useful / (t_mfma + t_valu_alone):.2f}; "
          f"VALU at its beside-MFMA rate were the only limit: {useful / max(t_mfma, t_valu_beside):.2f}")


if __name__ == "__main__":
    main()
