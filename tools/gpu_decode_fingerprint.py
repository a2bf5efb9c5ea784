"""Bitwise fingerprint of the bf16 decode / encode paths (for refactors that must not change results):
writes or checks tools/decode_fingerprint.json (sha256 of every output array)."""
import sys, hashlib
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from amuse_amd import weights as wts
from amuse_amd.engine import HipEngine
eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0))
gen = torch.Generator().manual_seed(3)
z = torch.randn(37, 128, generator=gen)
lens = torch.randint(1, 301, (37,), generator=gen).to(torch.int32)
out = {}
for prec in ("bf16", "fp32"):
    for name, ln in (("full", None), ("ragged", lens)):
        d = eng.vae_decode(z, ln, prec, return_feats=True)
        out[f"dec_{prec}_{name}_feats"] = d["feats"].cpu().numpy()
        out[f"dec_{prec}_{name}_poses"] = d["poses"].cpu().numpy()
        e = eng.vae_encode(d["feats"], ln, prec)
        for k, v in e.items():
            if torch.is_tensor(v):
                out[f"enc_{prec}_{name}_{k}"] = v.cpu().numpy()
import json
digest = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in out.items()}
path = REPO / "tools" / "decode_fingerprint.json"
if len(sys.argv) > 1 and sys.argv[1] == "write":
    (REPO / "gpurun_out").mkdir(exist_ok=True)
    json.dump(digest, open(REPO / "gpurun_out" / "decode_fingerprint.json", "w"), indent=1)
    print("written", len(digest), "digests to gpurun_out/decode_fingerprint.json (copy to tools/ to pin)")
else:
    ref = json.load(open(path))
    bad = [k for k in digest if ref.get(k) != digest[k]]
    for k in digest:
        print(k, "equal" if k not in bad else "DIFFERENT")
    sys.exit(1 if bad else 0)
