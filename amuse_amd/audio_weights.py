"""Parameter inventory + deterministic weights for the audio front-end (models/audio: 3 x AST -> con / emo / sty).

The reference builds each encoder as ``ASTModel(label_dim, fstride=10, tstride=10, input_fdim=128, input_tdim=1024,
model_size='base384')`` (AST_EVP.py:53-61) around timm 0.4.5's ``vit_deit_base_distilled_patch16_384``
(audio_main_new.py:66-70): 12 pre-norm blocks of width 768 / 12 heads / MLP 3072 over 2 + 12 x 101 tokens.
timm is not installed, so the key list below is restated from timm 0.4.5's ``DistilledVisionTransformer`` attribute
names (parity unpinned - see oracle/audio_oracle.py); only the tensors the forward pass reads are listed.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .weights import _rng_for

AST_DIM = 768
AST_HEADS = 12
AST_LAYERS = 12
AST_MLP = 3072
AST_FDIM, AST_TDIM = 128, 1024            # mel bins, frames (configs/base_new.json wav_dtw_mfcc)
AST_PATCH, AST_STRIDE = 16, 10
AST_F = (AST_FDIM - AST_PATCH) // AST_STRIDE + 1   # 12
AST_T = (AST_TDIM - AST_PATCH) // AST_STRIDE + 1   # 101
AST_TOKENS = 2 + AST_F * AST_T                     # 1214
AST_FEAT = 256
ENCODERS = ("con", "emo", "sty")


def ast_param_spec() -> "OrderedDict[str, Tuple[int, ...]]":
    """Forward-pass tensors of one ASTModel, timm 0.4.5 names under ``v.`` plus ``feature_head`` (audio_main_new.py:77)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["v.cls_token"] = (1, 1, AST_DIM)
    s["v.dist_token"] = (1, 1, AST_DIM)
    s["v.pos_embed"] = (1, AST_TOKENS, AST_DIM)
    s["v.patch_embed.proj.weight"] = (AST_DIM, 1, AST_PATCH, AST_PATCH)
    s["v.patch_embed.proj.bias"] = (AST_DIM,)
    for i in range(AST_LAYERS):
        p = f"v.blocks.{i}"
        s[f"{p}.norm1.weight"] = (AST_DIM,)
        s[f"{p}.norm1.bias"] = (AST_DIM,)
        s[f"{p}.attn.qkv.weight"] = (3 * AST_DIM, AST_DIM)
        s[f"{p}.attn.qkv.bias"] = (3 * AST_DIM,)
        s[f"{p}.attn.proj.weight"] = (AST_DIM, AST_DIM)
        s[f"{p}.attn.proj.bias"] = (AST_DIM,)
        s[f"{p}.norm2.weight"] = (AST_DIM,)
        s[f"{p}.norm2.bias"] = (AST_DIM,)
        s[f"{p}.mlp.fc1.weight"] = (AST_MLP, AST_DIM)
        s[f"{p}.mlp.fc1.bias"] = (AST_MLP,)
        s[f"{p}.mlp.fc2.weight"] = (AST_DIM, AST_MLP)
        s[f"{p}.mlp.fc2.bias"] = (AST_DIM,)
    s["v.norm.weight"] = (AST_DIM,)
    s["v.norm.bias"] = (AST_DIM,)
    s["feature_head.0.weight"] = (AST_DIM,)
    s["feature_head.0.bias"] = (AST_DIM,)
    s["feature_head.1.weight"] = (AST_FEAT, AST_DIM)
    s["feature_head.1.bias"] = (AST_FEAT,)
    return s


def ast_param_count() -> int:
    return int(sum(int(np.prod(v)) for v in ast_param_spec().values()))


_MADE: Dict[tuple, Dict[str, np.ndarray]] = {}


def make_ast_weights(seed: int, encoder: str) -> Dict[str, np.ndarray]:
    """Deterministic float32 weights for encoder ``con`` | ``emo`` | ``sty``: N(0, 0.02) matrices (timm's trunc_normal
    scale), LayerNorm parameters perturbed away from (1, 0), small non-zero biases.  The arrays of a (seed, encoder) pair are generated once per process
    (345 MB, 3.3 s) and handed out read-only in a fresh dict."""
    assert encoder in ENCODERS
    if (seed, encoder) in _MADE:
        return OrderedDict(_MADE[(seed, encoder)])
    out: Dict[str, np.ndarray] = OrderedDict()
    for name, shape in ast_param_spec().items():
        g = _rng_for(seed, f"ast.{encoder}.{name}")
        parts = name.split(".")
        leaf = parts[-1]
        is_norm = parts[-2].startswith("norm") or name.startswith("feature_head.0")
        if is_norm:
            a = (1.0 if leaf == "weight" else 0.0) + g.uniform(-0.1, 0.1, shape)
        elif leaf == "bias":
            a = g.uniform(-0.05, 0.05, shape)
        else:
            a = 0.02 * g.standard_normal(shape, dtype=np.float32)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
        out[name].flags.writeable = False
    if len(_MADE) < 6:
        _MADE[(seed, encoder)] = out
    return OrderedDict(out)
