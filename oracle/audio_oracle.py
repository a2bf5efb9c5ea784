"""ORACLE (test infrastructure, CPU) for the audio front-end: waveform -> kaldi fbank -> 3 x AST -> con / emo / sty.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file; the product path is the HIP
library.  It restates, in explicit torch-CPU math:

  * PretrainedLPDM_v1.process_single_seq (models/latent_diffusion/infer_ldm.py:180-193): kaldi fbank (128 mel bins,
    25 ms / 10 ms, hanning, no dither, htk_compat) -> zero-pad / crop to 1024 frames -> (x - mean) / (2 std) with
    configs/base_new.json wav_dtw_mfcc.dataset_mean / dataset_std -> AST_EVP.eval_func (models/audio/AST_EVP.py:84-90)
  * ASTModel.forward (models/audio/audio_main_new.py:174-204) with frame_based_feats = True (base_new.json):
    conv patch embedding 16 x 16 stride 10 -> [cls | dist | 12 x 101 patches] + pos_embed -> 12 pre-norm ViT blocks ->
    LayerNorm -> mean over the patch tokens -> feature_head = LayerNorm + Linear(768 -> 256).

PARITY UNPINNED.  Both third-party pieces are absent from /root/reference and from this image, so the reference
modules cannot be imported to generate golden vectors:
  * timm==0.4.5 (audio_main_new.py:52 asserts the version): ``vit_deit_base_distilled_patch16_384`` - the block
    (x + attn(norm1 x); x + mlp(norm2 x), qkv_bias, scale = head_dim ** -0.5, exact-erf GELU, LayerNorm eps 1e-6) is
    restated from the published DeiT / ViT definition;
  * torchaudio.compliance.kaldi.fbank (infer_ldm.py:182): restated from Kaldi's documented feature pipeline
    (remove DC offset, pre-emphasis 0.97 with replicate padding, window, zero-pad to 512, power spectrum, 128
    triangular filters equally spaced on the mel scale 1127 ln(1 + f / 700) between 20 Hz and Nyquist, log with
    floor FLT_EPSILON).
Cross-checks the tests do make: ast_forward against the independent implementation of the same published model in
the ``transformers`` package (ASTModel, weights mapped key by key); kaldi_fbank / prepare_fbank against the same package's
own numpy restatement of torchaudio's kaldi fbank (ASTFeatureExtractor without torchaudio: 3.7e-4 max / 5e-6 mean in the
log-mel domain on 3-12 s signals, pad-then-normalise included), against a float64 DFT restatement and against analytically
known inputs (pure tones land in the right mel bin; a constant signal gives the log floor).  Two third-party restatements
agreeing is evidence, not a pin: neither is the reference's own timm 0.4.5 / torchaudio call.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

SAMPLE_RATE = 16000
FRAME_LEN, FRAME_SHIFT, FFT_LEN = 400, 160, 512
N_MEL = 128
TARGET_LEN = 1024
NORM_MEAN, NORM_STD = -9.173025, 5.062332      # configs/base_new.json TRAIN_PARAM.wav_dtw_mfcc
DIM, HEADS, LAYERS, MLP = 768, 12, 12, 3072
PATCH, STRIDE = 16, 10


def mel_banks() -> torch.Tensor:
    """(128, 257) triangular filters, Kaldi's get_mel_banks for 16 kHz / 512-point FFT / 20 Hz .. Nyquist."""
    nyq = 0.5 * SAMPLE_RATE
    mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)
    mel_low, mel_high = 1127.0 * math.log(1.0 + 20.0 / 700.0), 1127.0 * math.log(1.0 + nyq / 700.0)
    delta = (mel_high - mel_low) / (N_MEL + 1)
    b = torch.arange(N_MEL, dtype=torch.float32)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    m = mel((SAMPLE_RATE / FFT_LEN) * torch.arange(FFT_LEN // 2, dtype=torch.float32))[None, :]
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    bins = torch.clamp(torch.minimum(up, down), min=0.0)
    return torch.nn.functional.pad(bins, (0, 1))       # the Nyquist bin carries no weight


def kaldi_fbank(wave: torch.Tensor) -> torch.Tensor:
    """wave (n,) or (C, n) float32 (channel 0 is used) -> (frames, 128) log mel energies, frames = 1 + (n - 400) // 160."""
    w = torch.as_tensor(wave, dtype=torch.float32)
    if w.dim() == 2:
        w = w[0]
    n = w.shape[0]
    if n < FRAME_LEN:
        return torch.zeros(0, N_MEL)
    m = 1 + (n - FRAME_LEN) // FRAME_SHIFT
    frames = w.unfold(0, FRAME_LEN, FRAME_SHIFT)[:m].clone()                 # snip_edges
    frames = frames - frames.mean(dim=1, keepdim=True)                       # remove_dc_offset
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)                 # replicate padding on the left
    frames = frames - 0.97 * prev                                            # pre-emphasis
    frames = frames * torch.hann_window(FRAME_LEN, periodic=False)           # window_type = 'hanning'
    frames = torch.nn.functional.pad(frames, (0, FFT_LEN - FRAME_LEN))
    power = torch.fft.rfft(frames).abs().pow(2.0)                            # (m, 257)
    e = power @ mel_banks().T
    return torch.log(torch.clamp(e, min=torch.finfo(torch.float32).eps))


def prepare_fbank(wave: torch.Tensor) -> torch.Tensor:
    """infer_ldm.py:182-190: fbank -> zero-pad (BEFORE normalisation) / crop to 1024 frames -> (x - mean) / (2 std)."""
    fb = kaldi_fbank(wave)
    p = TARGET_LEN - fb.shape[0]
    if p > 0:
        fb = torch.nn.functional.pad(fb, (0, 0, 0, p))
    elif p < 0:
        fb = fb[:TARGET_LEN]
    return (fb - NORM_MEAN) / (NORM_STD * 2)


def _rb(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if on else x


def _lin(x, w, b, bf16):
    return _rb(x, bf16) @ _rb(w, bf16).T + b


def ast_forward(W: Dict[str, torch.Tensor], fbank: torch.Tensor, frame_based_feats: bool = True,
                emulate_bf16: bool = False, taps: Optional[dict] = None) -> torch.Tensor:
    """ASTModel.forward (audio_main_new.py:174-204) -> 'feature' (B, 256).  fbank (B, 1024, 128).
    emulate_bf16 rounds every GEMM operand (weights and activations) to bf16, accumulating in fp32, as the HIP
    throughput path does; softmax, LayerNorm, GELU and the residual stream stay fp32."""
    bf = emulate_bf16
    x = fbank[:, None].transpose(2, 3)                                       # (B, 1, 128, 1024)
    B = x.shape[0]
    cols = torch.nn.functional.unfold(x, kernel_size=PATCH, stride=STRIDE)   # (B, 256, 12 * 101), frequency-major
    pw = W["v.patch_embed.proj.weight"].reshape(DIM, PATCH * PATCH)
    x = _lin(cols.transpose(1, 2), pw, W["v.patch_embed.proj.bias"], bf)     # flatten(2).transpose(1, 2)
    x = torch.cat([W["v.cls_token"].expand(B, -1, -1), W["v.dist_token"].expand(B, -1, -1), x], dim=1)
    x = x + W["v.pos_embed"]
    ln = lambda t, p, eps: torch.nn.functional.layer_norm(t, (t.shape[-1],), W[p + ".weight"], W[p + ".bias"], eps)
    hd = DIM // HEADS
    for i in range(LAYERS):
        p = f"v.blocks.{i}"
        h = ln(x, p + ".norm1", 1e-6)
        qkv = _lin(h, W[p + ".attn.qkv.weight"], W[p + ".attn.qkv.bias"], bf).reshape(B, -1, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]                                     # (B, H, N, hd)
        att = (_rb(q, bf) @ _rb(k, bf).transpose(-2, -1)) * (hd ** -0.5)
        att = att.softmax(dim=-1)
        o = (_rb(att, bf) @ _rb(v, bf)).transpose(1, 2).reshape(B, -1, DIM)
        x = x + _lin(o, W[p + ".attn.proj.weight"], W[p + ".attn.proj.bias"], bf)
        h = ln(x, p + ".norm2", 1e-6)
        h = torch.nn.functional.gelu(_lin(h, W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"], bf))
        x = x + _lin(h, W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"], bf)
        if taps is not None:
            taps[f"block{i}"] = x.clone()
    x = ln(x, "v.norm", 1e-6)
    if taps is not None:
        taps["final"] = x.clone()
    pooled = x[:, 2:].mean(dim=1) if frame_based_feats else (x[:, 0] + x[:, 1]) / 2
    h = torch.nn.functional.layer_norm(pooled, (DIM,), W["feature_head.0.weight"], W["feature_head.0.bias"], 1e-5)
    return _lin(h, W["feature_head.1.weight"], W["feature_head.1.bias"], bf)


def audio_features(W3: Dict[str, Dict[str, torch.Tensor]], waves: Sequence[torch.Tensor], emulate_bf16: bool = False):
    """process_single_seq for a list of waveforms -> (con, emo, sty), each (B, 256)."""
    fb = torch.stack([prepare_fbank(w) for w in waves])
    return tuple(ast_forward(W3[n], fb, True, emulate_bf16) for n in ("con", "emo", "sty"))


def to_torch(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}
