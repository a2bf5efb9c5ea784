"""Host side of the audio front-end (reference models/latent_diffusion/infer_ldm.py:180-193 process_single_seq):
waveform -> kaldi fbank -> 3 x AST -> (con, emo, sty).  All compute is in libamuse_hip.so (csrc/k_audio.hip); this
module only builds the two constant tables of the fbank, flattens the three state dicts and moves pointers."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from . import audio_weights as aw
from .engine import _ptr, flatten_state_dict

NORM_MEAN, NORM_STD = -9.173025, 5.062332   # configs/base_new.json TRAIN_PARAM.wav_dtw_mfcc.dataset_{mean,std}
WHICH = {"con": 0, "emo": 1, "sty": 2}


def kaldi_tables() -> Tuple[np.ndarray, np.ndarray]:
    """(mel_banks [128, 257], hann window [400]) of kaldi fbank at 16 kHz / 25 ms / 512-point FFT / 20 Hz .. Nyquist
    (torchaudio.compliance.kaldi.get_mel_banks, window_type='hanning'; call site infer_ldm.py:182)."""
    f32 = np.float32
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + f / f32(700.0), dtype=f32)
    mel_low, mel_high = f32(1127.0 * math.log(1.0 + 20.0 / 700.0)), f32(1127.0 * math.log(1.0 + 8000.0 / 700.0))
    delta = f32((mel_high - mel_low) / f32(129))
    b = np.arange(128, dtype=f32)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + f32(1)) * delta, mel_low + (b + f32(2)) * delta
    m = mel(f32(16000.0 / 512.0) * np.arange(256, dtype=f32))[None, :]
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    banks = np.zeros((128, 257), dtype=f32)
    banks[:, :256] = np.maximum(np.minimum(up, down), f32(0))
    k = np.arange(400, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * math.pi * k / 399.0)).astype(f32)
    return np.ascontiguousarray(banks), np.ascontiguousarray(window)


class AudioEngine:
    """One amuse_audio_ctx on one GPU: the three AST encoders of AST_EVP (models/audio/AST_EVP.py:53-61)."""

    def __init__(self, con_sd: Dict[str, np.ndarray], emo_sd: Dict[str, np.ndarray], sty_sd: Dict[str, np.ndarray],
                 device="cuda:0", norm_mean: float = NORM_MEAN, norm_std: float = NORM_STD, frame_based_feats: bool = True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AmuseHipError("amuse_amd runs on an MI355X (torch device 'cuda:N'); there is no CPU path")
        spec = aw.ast_param_spec()
        flat = [flatten_state_dict(sd, spec) for sd in (con_sd, emo_sd, sty_sd)]
        banks, window = kaldi_tables()
        fp = C.POINTER(C.c_float)
        torch.cuda.init()
        self.ctx = self.lib.amuse_audio_create(self.device.index or 0, flat[0].ctypes.data_as(fp), flat[1].ctypes.data_as(fp),
                                               flat[2].ctypes.data_as(fp), flat[0].size, banks.ctypes.data_as(fp),
                                               window.ctypes.data_as(fp), norm_mean, norm_std, int(bool(frame_based_feats)))
        if not self.ctx:
            raise _lib.AmuseHipError(f"amuse_audio_create failed: {self.lib.amuse_last_error().decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.amuse_audio_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _waves(self, waves) -> torch.Tensor:
        w = torch.as_tensor(waves) if not isinstance(waves, torch.Tensor) else waves
        if w.dim() == 1:
            w = w[None]
        if w.dim() != 2:
            raise ValueError(f"waves must be (B, n_samples), got {tuple(w.shape)}")
        return w.to(device=self.device, dtype=torch.float32).contiguous()

    def fbank(self, waves) -> torch.Tensor:
        """(B, n) 16 kHz waveforms -> (B, 1024, 128) normalised, padded fbanks (infer_ldm.py:182-190)."""
        w = self._waves(waves)
        out = torch.empty(w.shape[0], 1024, 128, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_audio_fbank(self.ctx, _ptr(w), w.shape[1], w.shape[0], _ptr(out), self._stream()))
        return out

    def encode(self, which: str, fbank, tap_block: Optional[int] = None):
        """ASTModel.forward(...)['feature'] of encoder 'con' | 'emo' | 'sty': (B, 1024, 128) -> (B, 256)
        (+ the fp32 residual stream (B, 1214, 768) after block `tap_block`, for tests)."""
        fb = torch.as_tensor(fbank).to(device=self.device, dtype=torch.float32).contiguous()
        if fb.dim() != 3 or tuple(fb.shape[1:]) != (1024, 128):
            raise ValueError(f"fbank must be (B, 1024, 128), got {tuple(fb.shape)}")
        B = fb.shape[0]
        feat = torch.empty(B, 256, device=self.device, dtype=torch.float32)
        hid = torch.empty(B, 1214, 768, device=self.device, dtype=torch.float32) if tap_block is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_audio_encode(self.ctx, WHICH[which], _ptr(fb), B, _ptr(feat), _ptr(hid),
                                                   0 if tap_block is None else int(tap_block), self._stream()))
        return feat if tap_block is None else (feat, hid)

    def features(self, waves):
        """-> (con, emo, sty), each (B, 256): process_single_seq for a batch of waveforms."""
        w = self._waves(waves)
        B = w.shape[0]
        outs = [torch.empty(B, 256, device=self.device, dtype=torch.float32) for _ in range(3)]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.amuse_audio_features(self.ctx, _ptr(w), w.shape[1], B, _ptr(outs[0]), _ptr(outs[1]),
                                                     _ptr(outs[2]), self._stream()))
        return tuple(outs)

    def features_ragged(self, waves: Sequence):
        """process_single_seq for waveforms of DIFFERENT lengths as one batch: a fbank launch per waveform (each is padded /
        cropped to 1024 frames on its own, as the reference does call by call), then every encoder once over all of them.
        Row k is bitwise what process_single_seq(waves[k]) returns: a clip's features do not depend on its batch."""
        if len(waves) == 0:
            raise ValueError("no waveforms")
        fb = torch.cat([self.fbank(torch.as_tensor(w)[0] if torch.as_tensor(w).dim() == 2 else w) for w in waves])
        return tuple(self.encode(which, fb) for which in ("con", "emo", "sty"))

    def process_single_seq(self, sliced_chunk, framerate=16000 // 2, baseline=False):
        """(C, n) or (n,) waveform -> (con, emo, sty), each (1, 256) (infer_ldm.py:180-193; channel 0 as kaldi does)."""
        w = torch.as_tensor(sliced_chunk)
        if w.dim() == 2:
            w = w[0]
        return self.features(w[None])
