"""GPU: wall clock of the infer_gesture CLI on a reference-shaped tree with N synthetic 10 s WAVs (random-init weights), split
into its stages by wrapping the trainer's calls: WAV load + batch embedding, sampling + decode, NPZ writing.
Usage: python tools/gpu_cli_time.py [n_wavs]"""
import sys
import tempfile
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    from conftest import make_reference_tree
    from amuse_amd import main as cli, trainer as T
    acc = {}

    def wrap(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize()
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
            return r
        setattr(obj, name, g)
    wrap(T.trainer, "_embed_all", "load + embed")
    wrap(T, "run_jobs", "sample + decode")
    wrap(T.trainer, "_animate", "write NPZ")
    with tempfile.TemporaryDirectory() as d:
        root = make_reference_tree(Path(d) / "amuse", n_infer_wavs=n)
        for rnd in range(2):                       # the first run pays context creation and code-object loading
            acc.clear()
            t0 = time.perf_counter()
            w = cli.main(["--fn", "infer_gesture", "--root", str(root), "--random-init"])
            wall = time.perf_counter() - t0
            print(f"run {rnd}: {len(w)} NPZ in {wall * 1e3:.1f} ms  (" + ", ".join(f"{k} {v * 1e3:.1f} ms" for k, v in acc.items()) +
                  f"; per WAV {sum(acc.values()) / max(1, len(w)) * 1e3:.2f} ms in the three stages)", flush=True)


if __name__ == "__main__":
    main()
