"""GPU: the audio front-end (fbank + 3 x AST) through the C ABI against oracle/audio_oracle.py.
Parity is UNPINNED for this path (timm / torchaudio absent: see the oracle header); the oracle itself is cross-checked
against transformers.ASTModel on the CPU side.  Arithmetic is bf16 operands / fp32 accumulation, so the checks are
(i) teacher-forced per block against the bf16-emulating oracle on the kernel's own block input, (ii) whole network
against the fp32 oracle with a bf16-sized tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from amuse_amd import audio_weights as aw
    from amuse_amd.audio import AudioEngine
    from oracle import audio_oracle as ao
    W = {n: aw.make_ast_weights(0, n) for n in aw.ENCODERS}
    eng = AudioEngine(W["con"], W["emo"], W["sty"], "cuda:0")
    yield {"eng": eng, "W": {n: ao.to_torch(W[n]) for n in W}, "ao": ao}
    eng.close()


def _waves(n, B=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float32) / 16000.0
    base = 0.2 * torch.sin(2 * np.pi * 220.0 * t) + 0.1 * torch.sin(2 * np.pi * 1900.0 * t)
    return torch.stack([base * (0.5 + 0.5 * i) + 0.05 * torch.randn(n, generator=g) for i in range(B)])


def test_fbank_matches_oracle(env):
    ao, eng = env["ao"], env["eng"]
    for n in (159744, 16000, 200000, 399):
        w = _waves(n, 2, seed=n)
        fb = eng.fbank(w).cpu()
        ref = torch.stack([ao.prepare_fbank(x) for x in w])
        assert fb.shape == (2, 1024, 128)
        # log-mel values normalised by 2 std ~ 10: 1e-3 here is 1e-2 in log energy only on near-silent bins
        assert float((fb - ref).abs().max()) < 2e-3, n
        assert float((fb - ref).abs().mean()) < 2e-5, n


def test_encoder_blockwise_bf16_and_whole_network(env):
    ao, eng, W = env["ao"], env["eng"], env["W"]["emo"]
    fb = torch.stack([ao.prepare_fbank(x) for x in _waves(60000, 1, seed=5)])
    taps = {}
    with torch.no_grad():
        ref32 = ao.ast_forward(W, fb, True, emulate_bf16=False, taps=taps)
    # whole network vs the fp32 oracle: bf16 operand rounding through 12 blocks
    feat, hid11 = eng.encode("emo", fb, tap_block=11)
    rel = float((hid11.cpu() - taps["block11"]).norm() / taps["block11"].norm())
    assert rel < 2e-2, rel
    assert float((feat.cpu() - ref32).abs().max()) < 5e-2 * float(ref32.abs().max())
    # teacher-forced: block l+1 of the oracle (bf16 emulation) applied to the KERNEL's output of block l
    for l in (0, 5, 10):
        _, h_in = eng.encode("emo", fb, tap_block=l)
        _, h_out = eng.encode("emo", fb, tap_block=l + 1)
        x = h_in.cpu()
        p = f"v.blocks.{l + 1}"
        ln = lambda t, q: torch.nn.functional.layer_norm(t, (768,), W[q + ".weight"], W[q + ".bias"], 1e-6)
        rb = lambda t: t.to(torch.bfloat16).float()
        h = ln(x, p + ".norm1")
        qkv = (rb(h) @ rb(W[p + ".attn.qkv.weight"]).T + W[p + ".attn.qkv.bias"]).reshape(1, -1, 3, 12, 64).permute(2, 0, 3, 1, 4)
        att = ((rb(qkv[0]) @ rb(qkv[1]).transpose(-2, -1)) * 0.125).softmax(-1)
        o = (rb(att) @ rb(qkv[2])).transpose(1, 2).reshape(1, -1, 768)
        x = x + rb(o) @ rb(W[p + ".attn.proj.weight"]).T + W[p + ".attn.proj.bias"]
        h = torch.nn.functional.gelu(rb(ln(x, p + ".norm2")) @ rb(W[p + ".mlp.fc1.weight"]).T + W[p + ".mlp.fc1.bias"])
        x = x + rb(h) @ rb(W[p + ".mlp.fc2.weight"]).T + W[p + ".mlp.fc2.bias"]
        err = float((h_out.cpu() - x).abs().max())
        assert err < 2e-2 * float(x.abs().max()), (l, err)     # flash-softmax rounds un-normalised p to bf16
        assert float((h_out.cpu() - x).abs().mean()) < 5e-4 * float(x.abs().max()), l   # bf16 operand rounding flips


def test_features_batch_and_contract(env):
    ao, eng = env["ao"], env["eng"]
    w = _waves(48000, 3, seed=9)
    con, emo, sty = eng.features(w)
    assert con.shape == emo.shape == sty.shape == (3, 256) and con.dtype == torch.float32
    assert not torch.equal(con, emo) and bool(torch.isfinite(con).all())
    # batch position does not matter; process_single_seq takes (C, n) and uses channel 0
    c1, e1, s1 = eng.process_single_seq(torch.stack([w[1], w[2]]))
    assert c1.shape == (1, 256)
    assert torch.equal(c1[0], con[1]) and torch.equal(e1[0], emo[1]) and torch.equal(s1[0], sty[1])
    # <= 8 clips: the three encoders run concurrently on side streams with their own workspaces; results are those of
    # the one-encoder-at-a-time path (amuse_audio_encode) and of the sequential large-batch path, bitwise
    fb3 = eng.fbank(w)
    for name, got in (("con", con), ("emo", emo), ("sty", sty)):
        assert torch.equal(eng.encode(name, fb3), got), name
    w9 = torch.cat([w, _waves(48000, 6, seed=10)])
    c9, e9, s9 = eng.features(w9)
    assert torch.equal(c9[:3], con) and torch.equal(e9[:3], emo) and torch.equal(s9[:3], sty)
    c2, e2, s2 = eng.features(w)      # a second call reuses streams, events and workspaces
    assert torch.equal(c2, con) and torch.equal(e2, emo) and torch.equal(s2, sty)
    # more clips than one pass over the network takes (32): the second chunk reuses the workspace - whose pad rows (2 per clip
    # of 1216, the GEMM tiles' tail) now hold the first chunk's leftovers - and an odd clip count leaves half a row tile unowned
    w35 = torch.cat([w9, _waves(48000, 26, seed=11)])
    c35, e35, s35 = eng.features(w35)
    assert torch.equal(c35[:9], c9) and torch.equal(e35[:9], e9) and torch.equal(s35[:9], s9)
    tail = eng.features(w35[32:])
    assert all(torch.equal(a[32:], b) for a, b in zip((c35, e35, s35), tail)) and bool(torch.isfinite(s35).all())
    # the two pooling variants of audio_main_new.py:191-201
    from amuse_amd.audio import AudioEngine
    from amuse_amd import audio_weights as aw
    fb = eng.fbank(w[:1])
    with torch.no_grad():
        ref = ao.ast_forward(env["W"]["sty"], fb.cpu(), frame_based_feats=True)
    assert float((eng.encode("sty", fb).cpu() - ref).abs().max()) < 5e-2 * float(ref.abs().max())
    with pytest.raises(ValueError):
        eng.encode("con", torch.zeros(1, 1000, 128))


def test_host_mirror_process_single_seq_and_loader_helper(env):
    """PretrainedLPDM_v1.process_single_seq / _loader_helper_v1 drive the HIP audio path (infer_ldm.py:180-193, 416-493)."""
    from amuse_amd import audio_weights as aw, weights as wts
    from amuse_amd.infer_ldm import PretrainedLPDM_v1
    eng = env["eng"]
    m = PretrainedLPDM_v1.from_state_dicts(wts.make_denoiser_weights(0), wts.make_prior_weights(0), device="cuda:0")
    wave = _waves(160000 * 2 + 5000, 1, seed=21)                       # (C = 1, n)
    with pytest.raises(NotImplementedError):
        m.process_single_seq(wave)
    m.audio_engine = eng                                               # what set_audio_encoders builds
    con, emo, sty = m.process_single_seq(wave)
    assert con.shape == emo.shape == sty.shape == (1, 256)
    ref = eng.features(wave[0][None])
    assert torch.equal(con, ref[0]) and torch.equal(sty, ref[2])
    # the result feeds diffusion_backward unchanged
    out = m.diffusion_backward(1, con, emo, sty)
    assert out["poses"].shape == (1, 300, 55, 3) and bool(torch.isfinite(out["poses"]).all())
    # _loader_helper_v1: n // 160000 chunks, chunk k starting at SAMPLE k (the reference's slicing, infer_ldm.py:421)
    motion = torch.cat([0.3 * torch.randn(600, 165), torch.randn(600, 3)], -1)
    z = m._loader_helper_v1(motion, wave)
    assert z["z_motion"].shape == (2, 128) and z["z_con"].shape == (2, 256)
    c0 = eng.features(wave[0, 0:160000][None])[0]
    c1 = eng.features(wave[0, 1:160001][None])[0]
    assert torch.equal(z["z_con"][0], c0[0]) and torch.equal(z["z_con"][1], c1[0])
    # process_loader computes the latents itself when the takes carry raw motion + waveform (infer_ldm.py:392-399)
    m.style_transfer, m.emotion_control, m.style_Xemo_transfer = False, True, False
    data = {"wayne": {t: {"ld_motion": motion.numpy(), "ld_waveform": _waves(160000, 1, seed=s_)} for t, s_ in (("a", 31), ("b", 32))}}
    out = m.process_loader({"emotion_control": data, "emotion_control_info": "[wayne]_[happy]_first"})["emotion_control"]
    assert out["wayne"]["a"]["ld_z"].shape == (2, 128) and out["wayne"]["a"]["ld_z_con"].shape == (1, 256)
    assert torch.equal(out["wayne"]["a"]["ld_z_emo_b"], out["wayne"]["b"]["ld_z_emo"])
    # process_seq_list: waveforms of different lengths (shorter and longer than the 1024-frame window, stereo) as ONE batch -
    # row k bitwise what the call-by-call path of the reference's loop (trainer.py:516-523) gives
    ragged = [_waves(160000, 1, seed=41), _waves(50000, 2, seed=42), _waves(200000, 1, seed=43), _waves(163840 + 240, 1, seed=44)[0]]
    many = m.process_seq_list(ragged, framerate=16000)
    assert len(many) == 4
    for wv, (c_k, e_k, s_k) in zip(ragged, many):
        c1_, e1_, s1_ = m.process_single_seq(wv, framerate=16000)
        assert c_k.shape == (1, 256) and torch.equal(c_k, c1_) and torch.equal(e_k, e1_) and torch.equal(s_k, s1_)
    assert len(m.process_seq_list(ragged[:1])) == 1
    m.audio_engine = None                                              # the fixture owns the engine


def test_cli_infer_and_edit_from_a_reference_tree(tmp_path):
    """`python -m amuse_amd.main --fn {infer,edit}_gesture` from a reference-shaped tree (scripts/main.py:226-268): config +
    override YAMLs merged in memory, WAVs from the configured directories, NPZs under the reference's directory names."""
    import hashlib

    from conftest import make_reference_tree

    from amuse_amd import main as cli
    from amuse_amd.trainer import load_wav
    root = make_reference_tree(tmp_path / "amuse", n_infer_wavs=2)
    digest = lambda: {str(p): hashlib.sha256(p.read_bytes()).hexdigest() for p in sorted((root / "configs").iterdir())}
    before = digest()
    a = load_wav(root / "viz_dump/test/e_speech/9_miranda_source.wav")
    assert a.shape == (1, 159744) and a.dtype == torch.float32 and float(a.abs().max()) <= 1.0
    w = cli.main(["--fn", "infer_gesture", "--root", str(root), "--random-init"])
    # one diffusion_backward(1, ...) per audio, each visualised as rst_0 / seq_0 (trainer.py:516-539), actor "scott"
    assert len(w) == 2 and all(p.name.startswith("scott_seq_0_") and p.name.endswith("_motion_smplx.npz") for p in w)
    rel = w[0].relative_to(root / "viz_dump/test/gesture").parts
    assert rel[0].startswith("Custom_audios_") and rel[0].endswith("_E0") and rel[1:4] == ("rep0", "rst_0", "seq_0")
    z = np.load(w[0], allow_pickle=True)
    assert z["poses"].shape == (300, 55, 3) and z["poses"].dtype == np.float32 and str(z["gender"]) == "male"
    assert not np.array_equal(z["poses"], np.load(w[1])["poses"])
    # --precision fp32x: the same job in the fast parity mode - the same clips and noise, SMPL-X poses within the two modes' conversion
    # noise of each other (the audio embeddings are bf16-computed and identical in both runs)
    wx = cli.main(["--fn", "infer_gesture", "--root", str(root), "--random-init", "--precision", "fp32x"])
    zx = np.load(wx[0], allow_pickle=True)
    dj = np.linalg.norm(zx["poses"] - z["poses"], axis=-1)
    assert float(np.median(dj)) < 3e-5 and float((dj < 1e-4).mean()) > 0.95, (float(np.median(dj)), float(dj.max()))
    # --precision fp16 (throughput mode on fp16 operands): the CLI's sampler is DDIM-50 - poses within the mode's measured drift of
    # the fp32 run (geodesic median 0.45 deg at 64 clips; bf16: 3.6 deg)
    wh = cli.main(["--fn", "infer_gesture", "--root", str(root), "--random-init", "--precision", "fp16"])
    zh = np.load(wh[0], allow_pickle=True)
    assert np.isfinite(zh["poses"]).all() and float(np.median(np.linalg.norm(zh["poses"] - z["poses"], axis=-1))) < 3e-2
    w2 = cli.main(["--fn", "edit_gesture", "--root", str(root), "--random-init"])
    assert [p.parts[-3] for p in w2] == ["rst_0", "rst_1"] and all(p.name.startswith("miranda_seq_0_") for p in w2)
    assert str(np.load(w2[0], allow_pickle=True)["gender"]) == "female"
    p0, p1 = (np.load(x, allow_pickle=True)["poses"] for x in w2)
    assert np.isfinite(p0).all() and not np.array_equal(p0, p1)   # fresh noise per call AND the target's emotion
    assert digest() == before                                       # nothing written into the configuration tree
    with pytest.raises(SystemExit):
        cli.main(["--fn", "train_audio", "--root", str(root)])
    with pytest.raises(FileNotFoundError):                          # no checkpoints in the tree and no --random-init
        cli.main(["--fn", "infer_gesture", "--root", str(root)])
