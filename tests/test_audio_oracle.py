"""CPU: the audio front-end oracle (oracle/audio_oracle.py) against what CAN be checked here - timm and torchaudio are
absent (parity unpinned, see the oracle header): an independent implementation of the same published AST model
(transformers.ASTModel), a float64 DFT restatement of the fbank, and analytically known inputs."""
import math

import numpy as np
import pytest
import torch

from amuse_amd import audio_weights as aw
from oracle import audio_oracle as ao


def test_ast_spec_counts():
    spec = aw.ast_param_spec()
    assert spec["v.pos_embed"] == (1, 1214, 768) and aw.AST_TOKENS == 1214
    # DeiT-B backbone without its two classifier heads: 85.8 M, + patch/pos adaptations + feature head
    n_block = 2 * 768 * 2 + 3 * 768 * 768 + 3 * 768 + 768 * 768 + 768 + 2 * 768 * 3072 + 3072 + 768
    assert aw.ast_param_count() == 12 * n_block + 2 * 768 + 1214 * 768 + 768 * 256 + 768 + 2 * 768 + 2 * 768 + 256 * 768 + 256


def test_fbank_shapes_and_known_inputs():
    sr = ao.SAMPLE_RATE
    n = 159744                                     # the reference's e_speech samples: 996 frames
    t = torch.arange(n, dtype=torch.float64) / sr
    fb = ao.kaldi_fbank(torch.sin(2 * math.pi * 1000.0 * t).float())
    assert fb.shape == (996, 128)
    # a 1 kHz tone peaks in the filter whose centre is nearest 1 kHz on the mel scale
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(8000.0)
    centre = (mel(1000.0) - lo) / ((hi - lo) / 129) - 1.0
    assert abs(int(fb.mean(0).argmax()) - centre) <= 1.0
    # a constant signal is removed by the DC-offset step: every bin sits on the log floor
    flat = ao.kaldi_fbank(torch.full((4000,), 0.3))
    assert torch.allclose(flat, torch.full_like(flat, math.log(torch.finfo(torch.float32).eps)))
    assert ao.kaldi_fbank(torch.zeros(399)).shape == (0, 128)
    assert ao.kaldi_fbank(torch.randn(2, 400)).shape == (1, 128)          # (channels, n): channel 0
    # padding happens BEFORE normalisation: padded rows are (0 - mean) / (2 std)   (infer_ldm.py:185-190)
    p = ao.prepare_fbank(torch.randn(16000, generator=torch.Generator().manual_seed(0)))
    assert p.shape == (1024, 128)
    assert torch.allclose(p[98:], torch.full_like(p[98:], -ao.NORM_MEAN / (2 * ao.NORM_STD)))
    assert ao.prepare_fbank(torch.randn(200000, generator=torch.Generator().manual_seed(1))).shape == (1024, 128)


def test_fbank_against_float64_dft():
    g = torch.Generator().manual_seed(3)
    w = (0.1 * torch.randn(2000, generator=g)).float()
    fb = ao.kaldi_fbank(w)
    x = w.double().unfold(0, 400, 160)
    x = x - x.mean(1, keepdim=True)
    x = x - 0.97 * torch.cat([x[:, :1], x[:, :-1]], 1)
    k = torch.arange(400, dtype=torch.float64)
    x = x * (0.5 - 0.5 * torch.cos(2 * math.pi * k / 399))
    f = torch.arange(257, dtype=torch.float64)[:, None] * torch.arange(400, dtype=torch.float64)[None, :]
    re, im = x @ torch.cos(2 * math.pi * f / 512).T, x @ torch.sin(2 * math.pi * f / 512).T
    ref = torch.log(torch.clamp((re * re + im * im) @ ao.mel_banks().double().T, min=1.1920929e-07))
    assert float((fb.double() - ref).abs().max()) < 2e-4


def test_fbank_and_normalisation_against_transformers_feature_extractor():
    """transformers' ASTFeatureExtractor without torchaudio falls back to its own numpy restatement of torchaudio's kaldi fbank
    (audio_utils.spectrogram: DC removal, pre-emphasis 0.97, hann window, 512-point power spectrum, kaldi mel filters, log with
    the FLT_EPSILON floor), pads to 1024 frames BEFORE normalising and normalises as (x - mean) / (2 std) - the pipeline of
    infer_ldm.py:182-190, written by a third party.  The oracle's kaldi_fbank / prepare_fbank must agree with it."""
    tr = pytest.importorskip("transformers")
    import warnings
    import numpy as np
    from transformers.utils import is_speech_available
    if is_speech_available():
        pytest.skip("torchaudio present: the extractor would call it instead of its own restatement")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # 'at least one mel filter has all zero values' (the same filters as Kaldi's)
        raw = tr.ASTFeatureExtractor(do_normalize=False)
        norm = tr.ASTFeatureExtractor(do_normalize=True, mean=ao.NORM_MEAN, std=ao.NORM_STD)
    rs = np.random.RandomState(0)
    for n, amp in ((159744, 0.1), (48000, 0.02), (192000, 0.3)):       # the reference's e_speech length; shorter; longer than 1024 frames
        t = np.arange(n) / 16000.0
        x = (amp * (np.sin(2 * np.pi * 220 * t) + 0.5 * np.sin(2 * np.pi * 1760 * t + 1)) + 0.3 * amp * rs.randn(n)).astype(np.float32)
        theirs = raw(x, sampling_rate=16000, return_tensors="np")["input_values"][0]
        mine = ao.kaldi_fbank(torch.from_numpy(x)).numpy()
        m = min(mine.shape[0], 1024)
        assert theirs.shape == (1024, 128) and mine.shape[0] == 1 + (n - 400) // 160
        assert float(np.abs(theirs[:m] - mine[:m]).max()) < 1e-3 and float(np.abs(theirs[:m] - mine[:m]).mean()) < 2e-5
        assert m == 1024 or float(np.abs(theirs[m:]).max()) == 0.0
        theirs_n = norm(x, sampling_rate=16000, return_tensors="np")["input_values"][0]
        assert float(np.abs(theirs_n - ao.prepare_fbank(torch.from_numpy(x)).numpy()).max()) < 1e-4


def test_ast_forward_against_transformers_implementation():
    tr = pytest.importorskip("transformers")
    W = ao.to_torch(aw.make_ast_weights(0, "emo"))
    cfg = tr.ASTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                       hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                       layer_norm_eps=1e-6, patch_size=16, qkv_bias=True, frequency_stride=10, time_stride=10,
                       max_length=1024, num_mel_bins=128)
    m = tr.ASTModel(cfg).eval()   # key names below are those of transformers 5.x
    sd = {"embeddings.cls_token": W["v.cls_token"], "embeddings.distillation_token": W["v.dist_token"],
          "embeddings.position_embeddings": W["v.pos_embed"],
          "embeddings.patch_embeddings.projection.weight": W["v.patch_embed.proj.weight"],
          "embeddings.patch_embeddings.projection.bias": W["v.patch_embed.proj.bias"],
          "layernorm.weight": W["v.norm.weight"], "layernorm.bias": W["v.norm.bias"]}
    for i in range(12):
        p, q = f"v.blocks.{i}", f"layers.{i}"
        qw, qb = W[p + ".attn.qkv.weight"], W[p + ".attn.qkv.bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[f"{q}.attention.{n}.weight"] = qw[768 * j:768 * (j + 1)]
            sd[f"{q}.attention.{n}.bias"] = qb[768 * j:768 * (j + 1)]
        sd[f"{q}.attention.o_proj.weight"], sd[f"{q}.attention.o_proj.bias"] = W[p + ".attn.proj.weight"], W[p + ".attn.proj.bias"]
        sd[f"{q}.layernorm_before.weight"], sd[f"{q}.layernorm_before.bias"] = W[p + ".norm1.weight"], W[p + ".norm1.bias"]
        sd[f"{q}.layernorm_after.weight"], sd[f"{q}.layernorm_after.bias"] = W[p + ".norm2.weight"], W[p + ".norm2.bias"]
        sd[f"{q}.mlp.fc1.weight"], sd[f"{q}.mlp.fc1.bias"] = W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"]
        sd[f"{q}.mlp.fc2.weight"], sd[f"{q}.mlp.fc2.bias"] = W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    fb = ao.prepare_fbank(0.1 * torch.randn(60000, generator=torch.Generator().manual_seed(5)))[None]
    taps = {}
    with torch.no_grad():
        feat = ao.ast_forward(W, fb, frame_based_feats=True, taps=taps)
        hs = m(input_values=fb).last_hidden_state                      # after the final LayerNorm
    assert feat.shape == (1, 256) and hs.shape == (1, 1214, 768)
    assert float((taps["final"] - hs).abs().max()) < 2e-4              # |x| ~ 3 after LayerNorm
    # both pooling variants of audio_main_new.py:191-201 on top of the cross-checked hidden states
    h = torch.nn.functional.layer_norm(hs[:, 2:].mean(1), (768,), W["feature_head.0.weight"], W["feature_head.0.bias"], 1e-5)
    assert float((h @ W["feature_head.1.weight"].T + W["feature_head.1.bias"] - feat).abs().max()) < 2e-4
