#!/usr/bin/env python3
"""One-off source tool of round 6: resolve the retired A/B build switches of amuse_amd/csrc to their production values.
   retire.py FILE...   (in place).  UNDEF = `#ifdef X` ablation branches that production never compiled; VALUED = `#ifndef X / #define X v`
tuning knobs whose every measured alternative lost (profiles/r0[1-5]_*_ab.txt): the guard goes, the value stays as a constexpr."""
import re
import sys

UNDEF = {"AMUSE_ABL_C1LITE", "AMUSE_ABL_C2LITE", "AMUSE_ABL_C2LITE_W", "AMUSE_ABL_C2LITE_R", "AMUSE_ABL_NOSTATS", "AMUSE_ABL_NOLATE",
         "AMUSE_F32X_FAST_ERF", "AMUSE_X_B_PRIO_KEEP", "AMUSE_ATTN_NOC"}
VALUED = {"AMUSE_GELU_SCALAR": 0, "AMUSE_LO_NT": 0, "AMUSE_STREAM_NT": 0, "AMUSE_F_SKIP_NT": 0, "AMUSE_FABL": 0, "AMUSE_F_PF": 2,
          "AMUSE_F_FFN_MIX": 0, "AMUSE_F_FFN_VALU": 9, "AMUSE_F_LN_UNROLL": 0, "AMUSE_F_ATTN_PIPE": 0, "AMUSE_F_ATTN_NQ2": 0,
          "AMUSE_C1_N1": 12, "AMUSE_C1_N2": 12, "AMUSE_B_EARLY": 20, "AMUSE_B_FFN_PRIO": 0, "AMUSE_FFN_VALU_PER_MFMA": 7,
          "AMUSE_C2_N0": 16, "AMUSE_C2_N1": 12, "AMUSE_F32X_DELAY": 0,
          "AMUSE_X_C1_N1": 12, "AMUSE_X_C1_N2": 12, "AMUSE_X_C2_N1": 8, "AMUSE_X_B_EARLY": 32, "AMUSE_X_A_DEFER": 1, "AMUSE_X_B_PRIO": 3,
          "AMUSE_X_ERF": 1, "AMUSE_X_A_DEFER2": 2, "AMUSE_ATTN_ABL": 0, "AMUSE_GEMM_ABL": 0, "AMUSE_GEMM_PROF": 0,
          "AMUSE_GEMM_COPY_WAVES": 4, "AMUSE_ATTN_SPLIT_MAX": 48, "AMUSE_FX_ERF": 2, "AMUSE_FX_ABL": 0, "AMUSE_FX_FFN_PIPE": 1,
          "AMUSE_FX_DMA_SPLIT": 0, "AMUSE_FX_ATTN": 1, "AMUSE_R8_FAST_ERF": 0, "AMUSE_R8_ABL": 0, "AMUSE_R8_PROD": 1, "AMUSE_R8_NT": 1,
          "AMUSE_R8_BUFS": 3, "AMUSE_GEMM_CG": 3}
ALL = set(UNDEF) | set(VALUED)
names = lambda s: set(re.findall(r"\bAMUSE_[A-Z0-9_]+\b", s))


def ev(expr):
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "0" if m.group(1) in UNDEF else "1", expr.split("//")[0])
    e = re.sub(r"\bAMUSE_[A-Z0-9_]+\b", lambda m: str(VALUED.get(m.group(0), 0)), e)
    e = e.replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace(" not =", "!=")
    return bool(eval(e))


def run(path):
    src = open(path).read().split("\n")
    out, i = [], 0
    stack = []     # entries: None (a conditional we do not touch) or [emitting_before, taken_already, live_now]
    live = lambda: all(s is None or s[2] for s in stack)
    while i < len(src):
        ln = src[i]
        s = ln.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", s)
        if not m:
            if live():
                out.append(ln)
            i += 1
            continue
        kind, rest = m.group(1), m.group(2).strip()
        if kind in ("ifdef", "ifndef", "if"):
            mac = names(rest)
            ours = bool(mac) and mac <= ALL
            if kind == "ifndef" and ours and rest.split()[0] in VALUED:     # guard block: #ifndef X / #define ... / #endif
                j = i + 1
                body = []
                while not src[j].strip().startswith("#endif"):
                    body.append(src[j]); j += 1
                if all(b.strip().startswith("#define") or not b.strip() or b.strip().startswith("//") for b in body):
                    if live():
                        for b in body:
                            d = re.match(r"(\s*)#define\s+(\w+)\s+(\S+)(.*)", b)
                            out.append(f"{d.group(1)}constexpr int {d.group(2)} = {d.group(3)};{d.group(4)}" if d else b)
                    i = j + 1
                    continue
            if not ours:
                stack.append(None)
                if live():
                    out.append(ln)
            else:
                v = (rest.split()[0] not in UNDEF) if kind == "ifdef" else (rest.split()[0] in UNDEF) if kind == "ifndef" else ev(rest)
                if kind == "ifdef" and rest.split()[0] in VALUED:
                    v = True
                stack.append([live(), v, v])
        elif kind == "elif":
            if stack[-1] is None:
                if live():
                    out.append(ln)
            else:
                assert names(rest) <= ALL, (path, i, ln)
                v = (not stack[-1][1]) and ev(rest)
                stack[-1][2] = v
                stack[-1][1] = stack[-1][1] or v
        elif kind == "else":
            if stack[-1] is None:
                if live():
                    out.append(ln)
            else:
                stack[-1][2] = not stack[-1][1]
                stack[-1][1] = True
        else:
            top = stack.pop()
            if top is None and live():
                out.append(ln)
        i += 1
    assert not stack, path
    open(path, "w").write("\n".join(out))


for p in sys.argv[1:]:
    run(p)
