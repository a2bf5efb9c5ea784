"""GPU: the fused per-clip decode kernel (csrc/k_vae_fused.hip, bf16 mode) against the reference golden, the oracle and the
staged kernels (csrc/k_vae.hip) - MotionPrior.decode, vae.py:216-278, + 6D -> axis-angle, infer_ldm.py:168-173."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def env():
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    from oracle import amuse_oracle as orc
    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    eng = HipEngine(wd, wp, "cuda:0")
    yield {"eng": eng, "Wp": orc.to_torch(wp), "orc": orc}
    eng.set_decode_path("auto")
    eng.close()


def test_fused_decode_vs_reference_golden_and_oracle(env):
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    g = np.load(GOLDEN / "vae_decode.npz")
    eng.set_decode_path("fused")
    out = eng.vae_decode(g["z"], None, "bf16", return_feats=True)
    assert out["feats"].shape == (3, 300, 333) and bool(torch.isfinite(out["feats"]).all())
    assert _err(out["feats"], g["feats"]) < 6e-2                         # whole-network bf16 bound (|feats| ~ 3), vs the reference
    ref = orc.vae_decode(Wp, torch.from_numpy(g["z"]), None, emulate_bf16=True)
    d = (out["feats"].cpu() - ref).abs()
    assert float(d.max()) < 5e-2 and float(d.mean()) < 5e-3               # vs the oracle rounding the same operands (measured 1.7e-2 / 2.6e-3)
    # rotation epilogue on the kernel's own features (the same device function as the staged path)
    poses, trans = orc.feats_to_smplx(out["feats"].cpu(), "p3d")
    assert torch.equal(out["trans"].cpu(), out["feats"].cpu()[..., -3:])
    dd = torch.linalg.vector_norm(out["poses"].cpu() - poses, dim=-1)
    assert float(dd.median()) < 1e-5


def test_fused_decode_ragged_lengths(env):
    orc, eng, Wp = env["orc"], env["eng"], env["Wp"]
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(6, 128, generator=gen)
    lengths = [300, 173, 1, 160, 161, 17]
    eng.set_decode_path("fused")
    o = eng.vae_decode(z, lengths, "bf16", return_feats=True)
    ref = orc.vae_decode(Wp, z, lengths, emulate_bf16=True)
    assert _err(o["feats"], ref) < 5e-2
    for b, n in enumerate(lengths):                                        # output[~mask.T] = 0 (vae.py:274)
        if n < 300:
            assert float(o["feats"][b, n:].abs().max()) == 0.0 and float(o["poses"][b, n:].abs().max()) == 0.0
            assert float(o["trans"][b, n:].abs().max()) == 0.0
        assert float(o["feats"][b, :n].abs().max()) > 0.0
    # masked frames are dead inputs: the valid part does not depend on what lies behind the length, and a clip's result
    # does not depend on its neighbours or position in the batch (bitwise)
    o2 = eng.vae_decode(z[[1, 5]], [173, 17], "bf16", return_feats=True)
    assert torch.equal(o2["feats"][0], o["feats"][1]) and torch.equal(o2["feats"][1], o["feats"][5])


def test_fused_and_staged_decode_agree_and_auto_switches(env):
    eng = env["eng"]
    gen = torch.Generator().manual_seed(6)
    z = torch.randn(100, 128, generator=gen)
    eng.set_decode_path("staged")
    s = eng.vae_decode(z, None, "bf16", return_feats=True)
    eng.set_decode_path("fused")
    f = eng.vae_decode(z, None, "bf16", return_feats=True)
    d = (s["feats"] - f["feats"]).abs()
    assert not torch.equal(s["feats"], f["feats"])                         # really two kernels
    assert float(d.max()) < 5e-2 and float(d.mean()) < 5e-3               # same operands rounded, another summation order
    f2 = eng.vae_decode(z, None, "bf16", return_feats=True)
    assert torch.equal(f["feats"], f2["feats"]) and torch.equal(f["poses"], f2["poses"])   # deterministic
    eng.set_decode_path("auto")
    a = eng.vae_decode(z, None, "bf16", return_feats=True)                 # 100 clips >= 96: fused
    assert torch.equal(a["feats"], f["feats"])
    a8 = eng.vae_decode(z[:8], None, "bf16", return_feats=True)            # 8 clips: staged
    assert torch.equal(a8["feats"], s["feats"][:8])
    # fp32 parity mode never takes the fused kernel
    eng.set_decode_path("fused")
    g = np.load(GOLDEN / "vae_decode.npz")
    assert _err(eng.vae_decode(g["z"], None, "fp32", return_feats=True)["feats"], g["feats"]) < 2e-5
    with pytest.raises(Exception):
        eng.set_decode_path("nope")
