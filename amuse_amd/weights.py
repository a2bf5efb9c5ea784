"""Parameter inventory + deterministic weight generator for the AMUSE LDM hot path.

The reference ships no checkpoints (they are gated behind the project website), so every
parity test and bench in this repo runs on *regenerable* random weights: a counter-based
generator keyed by (seed, parameter name) that produces the same float32 tensors on any box.

The key lists and shapes reproduce the reference state dicts exactly:
  * ``Denoiser``   - /root/reference/models/latent_diffusion/denoiser.py:16-133 (130 entries,
    checkpoint prefix ``denoiser.`` - infer_ldm.py:91-104)
  * ``MotionPrior`` - /root/reference/models/latent_diffusion/vae.py:24-146 (un-prefixed keys,
    infer_pretrained_vae.py:46-47)
``tests/golden/state_dict_spec.json`` (written by oracle/gen_golden.py from the reference
modules themselves) pins both lists.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

# Architecture constants of configs/diff_latent_v2.json:23-47 and configs/prior_emotional_fing.json:6-20
D_MODEL = 128
N_HEADS = 4
FF_SIZE = 512
N_LAYERS = 9
N_SKIP = (N_LAYERS - 1) // 2  # 4 input blocks, 1 middle, 4 output blocks
COND_DIM = 256
PE_MAX_LEN = 500
N_FRAMES = 300
N_JOINTS = 55
N_FEATS = N_JOINTS * 6 + 3  # 333 = 55 x 6D rotation + 3 translation (vae.py:66-68: 201 + 132)


def _enc_layer(prefix: str, spec: "OrderedDict[str, Tuple[int, ...]]", d=D_MODEL, ff=FF_SIZE):
    spec[f"{prefix}.self_attn.in_proj_weight"] = (3 * d, d)
    spec[f"{prefix}.self_attn.in_proj_bias"] = (3 * d,)
    spec[f"{prefix}.self_attn.out_proj.weight"] = (d, d)
    spec[f"{prefix}.self_attn.out_proj.bias"] = (d,)
    spec[f"{prefix}.linear1.weight"] = (ff, d)
    spec[f"{prefix}.linear1.bias"] = (ff,)
    spec[f"{prefix}.linear2.weight"] = (d, ff)
    spec[f"{prefix}.linear2.bias"] = (d,)
    spec[f"{prefix}.norm1.weight"] = (d,)
    spec[f"{prefix}.norm1.bias"] = (d,)
    spec[f"{prefix}.norm2.weight"] = (d,)
    spec[f"{prefix}.norm2.bias"] = (d,)


def _dec_layer(prefix: str, spec: "OrderedDict[str, Tuple[int, ...]]", d=D_MODEL, ff=FF_SIZE):
    spec[f"{prefix}.self_attn.in_proj_weight"] = (3 * d, d)
    spec[f"{prefix}.self_attn.in_proj_bias"] = (3 * d,)
    spec[f"{prefix}.self_attn.out_proj.weight"] = (d, d)
    spec[f"{prefix}.self_attn.out_proj.bias"] = (d,)
    spec[f"{prefix}.multihead_attn.in_proj_weight"] = (3 * d, d)
    spec[f"{prefix}.multihead_attn.in_proj_bias"] = (3 * d,)
    spec[f"{prefix}.multihead_attn.out_proj.weight"] = (d, d)
    spec[f"{prefix}.multihead_attn.out_proj.bias"] = (d,)
    spec[f"{prefix}.linear1.weight"] = (ff, d)
    spec[f"{prefix}.linear1.bias"] = (ff,)
    spec[f"{prefix}.linear2.weight"] = (d, ff)
    spec[f"{prefix}.linear2.bias"] = (d,)
    for n in ("norm1", "norm2", "norm3"):
        spec[f"{prefix}.{n}.weight"] = (d,)
        spec[f"{prefix}.{n}.bias"] = (d,)


def _skip_stack(prefix: str, spec, layer_fn):
    """Key order of Skip Transformer {Encoder,Decoder} (cross_attention.py:18-34, 66-82)."""
    spec[f"{prefix}.norm.weight"] = (D_MODEL,)  # self.norm is registered first (cross_attention.py:23)
    spec[f"{prefix}.norm.bias"] = (D_MODEL,)
    for i in range(N_SKIP):
        layer_fn(f"{prefix}.input_blocks.{i}", spec)
    layer_fn(f"{prefix}.middle_block", spec)
    for i in range(N_SKIP):
        layer_fn(f"{prefix}.output_blocks.{i}", spec)
    for i in range(N_SKIP):
        spec[f"{prefix}.linear_blocks.{i}.weight"] = (D_MODEL, 2 * D_MODEL)
        spec[f"{prefix}.linear_blocks.{i}.bias"] = (D_MODEL,)


# Denoiser variants (denoiser.py:64-66,92-131): `arch` "trans_enc" (skip-transformer encoder over [sample | time, con, emo, sty]) or
# "trans_dec" (9 plain TransformerDecoderLayers: tgt = the sample, memory = [time, con, emo, sty] + mem_pos);
# `diffusion_only` (denoiser.py:64-66,177-187,191-199): the sample is the 300 x 333 pose sequence itself, embedded by pose_embd and
# projected back by pose_proj - the per-step S = 304 / S = 300 transformers of SURVEY.md section 0.2 / Appendix B.
ARCHS = {"trans_enc": 0, "trans_dec": 1, "trans_enc_pose": 2, "trans_dec_pose": 3}   # include/amuse_hip.h AMUSE_ARCH_*


def arch_id(arch: str = "trans_enc", diffusion_only: bool = False) -> int:
    return ARCHS[arch] + (2 if diffusion_only else 0)


def arch_of_id(aid: int):
    return ("trans_enc", "trans_dec")[aid & 1], bool(aid & 2)


def denoiser_param_spec(arch: str = "trans_enc", diffusion_only: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict keys/shapes of the reference ``Denoiser``: the shipped configuration (trans_enc, skip connections, learned PE)
    by default, or one of the variants above (key order = registration order in Denoiser.__init__)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if diffusion_only:
        spec["pose_embd.weight"] = (D_MODEL, N_FEATS)
        spec["pose_embd.bias"] = (D_MODEL,)
        spec["pose_proj.weight"] = (N_FEATS, D_MODEL)
        spec["pose_proj.bias"] = (N_FEATS,)
    spec["time_embedding.linear_1.weight"] = (D_MODEL, COND_DIM)
    spec["time_embedding.linear_1.bias"] = (D_MODEL,)
    spec["time_embedding.linear_2.weight"] = (D_MODEL, D_MODEL)
    spec["time_embedding.linear_2.bias"] = (D_MODEL,)
    for n in ("con", "emo", "sty"):
        spec[f"emb_proj_{n}.1.weight"] = (D_MODEL, COND_DIM)
        spec[f"emb_proj_{n}.1.bias"] = (D_MODEL,)
    spec["query_pos.pe"] = (PE_MAX_LEN, 1, D_MODEL)
    spec["mem_pos.pe"] = (PE_MAX_LEN, 1, D_MODEL)
    if arch == "trans_enc":
        _skip_stack("encoder", spec, _enc_layer)
    elif arch == "trans_dec":   # TransformerDecoder registers its layers first, then the norm (cross_attention.py:195-203)
        for i in range(N_LAYERS):
            _dec_layer(f"decoder.layers.{i}", spec)
        spec["decoder.norm.weight"] = (D_MODEL,)
        spec["decoder.norm.bias"] = (D_MODEL,)
    else:
        raise ValueError(f"Not supported architechure{arch}!")
    return spec


def prior_param_spec() -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict keys/shapes of the reference ``MotionPrior`` (encoder_decoder, MLP_DIST false)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    spec["global_motion_token"] = (2, D_MODEL)
    spec["query_pos_encoder.pe"] = (PE_MAX_LEN, 1, D_MODEL)
    spec["query_pos_decoder.pe"] = (PE_MAX_LEN, 1, D_MODEL)
    _skip_stack("encoder", spec, _enc_layer)
    _skip_stack("decoder", spec, _dec_layer)
    spec["skel_embedding.weight"] = (D_MODEL, N_FEATS)
    spec["skel_embedding.bias"] = (D_MODEL,)
    spec["final_layer.weight"] = (N_FEATS, D_MODEL)
    spec["final_layer.bias"] = (N_FEATS,)
    return spec


def _rng_for(seed: int, name: str) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = int.from_bytes(h[:16], "little")
    return np.random.Generator(np.random.Philox(key=key))


def _init_one(seed: int, name: str, shape: Tuple[int, ...]) -> np.ndarray:
    g = _rng_for(seed, name)
    parts = name.split(".")
    leaf = parts[-1]
    is_norm = len(parts) >= 2 and parts[-2].startswith("norm")
    if leaf == "pe":  # PositionEmbeddingLearned1D.reset_parameters: U(0,1) (position_encoding.py:150-151)
        a = g.random(shape, dtype=np.float64)
    elif leaf == "global_motion_token":
        a = g.standard_normal(shape)
    elif is_norm:
        # LayerNorm: perturbed away from (1, 0) so a swapped/ignored gamma/beta is visible in tests
        a = (1.0 if leaf == "weight" else 0.0) + g.uniform(-0.1, 0.1, shape)
    elif len(shape) >= 2:  # xavier-uniform, as _reset_parameters does (cross_attention.py:36-39)
        fan_out, fan_in = shape[0], shape[1]
        if leaf == "in_proj_weight":
            fan_out = shape[0] // 3
        lim = float(np.sqrt(6.0 / (fan_in + fan_out)))
        a = g.uniform(-lim, lim, shape)
    else:  # biases: small but non-zero (torch zero-inits MHA biases; zeros would hide bias bugs)
        a = g.uniform(-0.05, 0.05, shape)
    return np.ascontiguousarray(a, dtype=np.float32)


def make_weights(spec: "OrderedDict[str, Tuple[int, ...]]", seed: int, tag: str) -> Dict[str, np.ndarray]:
    """Deterministic float32 tensors for every entry of ``spec``; ``tag`` separates denoiser / prior."""
    return OrderedDict((k, _init_one(seed, f"{tag}/{k}", s)) for k, s in spec.items())


def make_denoiser_weights(seed: int = 0, arch: str = "trans_enc", diffusion_only: bool = False) -> Dict[str, np.ndarray]:
    tag = "denoiser" if (arch == "trans_enc" and not diffusion_only) else f"denoiser/{arch}{'/pose' if diffusion_only else ''}"
    return make_weights(denoiser_param_spec(arch, diffusion_only), seed, tag)


def make_prior_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    return make_weights(prior_param_spec(), seed, "prior")


def _joint_rest_6d(seed: int) -> np.ndarray:
    """55 rest rotations as the 6D representation the decoder emits: the first two ROWS of the matrix (pytorch3d's
    matrix_to_rotation_6d convention, the inverse of models/diffusion/utils/rotation_conversions.py:480-500).  Joint 0 = identity;
    most joints turn about a random axis by up to 1 rad; every 7th joint turns by 2.6-2.9 rad about a nearly axis-aligned axis of
    either sign, the regime in which the deployed matrix_to_axis_angle writes |aa| > pi.  Both groups sit away
    from the ties of the quaternion candidate selection (w = cos(a/2) against sin(a/2) * max|axis|), where that function jumps."""
    g = _rng_for(seed, "prior/rest_pose")
    ax = g.standard_normal((N_JOINTS, 3))
    ang = g.uniform(0.0, 1.0, N_JOINTS)
    for n, j in enumerate(range(3, N_JOINTS, 7)):
        ax[j] = 0.2 * ax[j]
        ax[j, n % 3] = 1.0 if (n // 3) % 2 == 0 else -1.0
        ang[j] = g.uniform(2.6, 2.9)
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang[0] = 0.0
    K = np.zeros((N_JOINTS, 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    s, c = np.sin(ang)[:, None, None], np.cos(ang)[:, None, None]
    R = np.eye(3)[None] + s * K + (1 - c) * (K @ K)       # Rodrigues
    return R[:, :2, :].reshape(N_JOINTS, 6)


def make_wellcond_prior_weights(seed: int = 1) -> Dict[str, np.ndarray]:
    """MotionPrior weights whose decoder emits WELL-CONDITIONED 6D rotations, as a trained decoder does: the random draw of
    `make_prior_weights(seed)` with final_layer.weight x 0.1 and final_layer.bias = the 6D image of a rest pose per joint (| 0 for the
    translation), so the Gram-Schmidt pivots of rotation_6d_to_matrix stay >= 0.5 (a plain random draw gives pivots down to 1e-3,
    which amplify any upstream rounding difference ~1000 x in the pose - a property of random weights, not of the path).
    Used by the `wellcond` reference-module fixtures (oracle/gen_golden.py --wellcond) and the end-to-end every-joint parity tests."""
    w = make_prior_weights(seed)
    w["final_layer.weight"] = np.ascontiguousarray(w["final_layer.weight"] * np.float32(0.1))
    b = np.zeros(N_FEATS, dtype=np.float32)
    b[: N_JOINTS * 6] = _joint_rest_6d(seed).reshape(-1).astype(np.float32)
    w["final_layer.bias"] = b
    return w


def n_params(w: Dict[str, np.ndarray]) -> int:
    return int(sum(v.size for v in w.values()))
