#!/usr/bin/env python3
"""Build-container script: extract the per-actor SMPL-X shape table of the reference into a DATA file.

The reference keeps the 300-d `betas` vector of every BEAT actor as a numpy literal in dm/utils/ldm_evals.py
(`wayne = np.array([...])` ... `katya = ...`, :456-2314; dispatch `fetchbetas`, :348-379) and writes it into every
output NPZ (`subject2genderbeta`, :67-71; models/diffusion/viz/visualizer.py:357-362).  This script parses those
literals with `ast` (nothing of the module is executed) and writes amuse_amd/data/smplx_betas.npz: one float64 (300,)
array per actor that `fetchbetas` can return.  /root/reference does not travel to the GPU box; the data file does.

  python tools/extract_betas.py [/root/reference]
"""
import ast
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
src = (ref / "dm/utils/ldm_evals.py").read_text()
tree = ast.parse(src)

# actors fetchbetas() dispatches on (the commented-out ones raise NotImplementedError in the reference too)
fetch = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "fetchbetas")
actors = []
for node in ast.walk(fetch):
    if isinstance(node, ast.Compare) and isinstance(node.comparators[0], ast.Constant):
        actors.append(node.comparators[0].value)

table = {}
for node in tree.body:
    if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
        name = node.targets[0].id
        if name in actors and isinstance(node.value, ast.Call) and ast.unparse(node.value.func) == "np.array":
            vals = ast.literal_eval(node.value.args[0])
            table[name] = np.asarray(vals, dtype=np.float64)
missing = [a for a in actors if a not in table]
assert not missing, f"no literal found for {missing}"
assert all(v.shape == (300,) for v in table.values()), {k: v.shape for k, v in table.items() if v.shape != (300,)}
out = REPO / "amuse_amd" / "data" / "smplx_betas.npz"
np.savez_compressed(out, **table)
print(f"{len(table)} actors -> {out} ({out.stat().st_size} bytes): {sorted(table)}")
