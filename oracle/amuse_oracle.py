"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement of the AMUSE latent-diffusion sampling path.

This file is *not* part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the shipped path (``amuse_amd``) never does and
fails loudly when its HIP library is missing.

What it restates (all citations relative to /root/reference):
  * ``Denoiser.forward`` (trans_enc + skip connections)   models/latent_diffusion/denoiser.py:135-204
  * ``Timesteps`` / ``TimestepEmbedding``                  models/latent_diffusion/utils/embeddings.py:245-322
  * ``SkipTransformerEncoder`` / ``...Decoder`` and the post-norm layers
                                                           models/latent_diffusion/utils/cross_attention.py:18-125,236-345
  * ``MotionPrior.decode``                                  models/latent_diffusion/vae.py:216-278
  * the ``Denoiser`` variants - ``arch: "trans_dec"`` (TransformerDecoder over memory = [time, con, emo, sty]) and
    ``diffusion_only`` (pose_embd / pose_proj around 300 raw-pose frames)   denoiser.py:64-66,116-131,174-204;
                                                           cross_attention.py:195-234,297-345
  * the sampling loop + output conversion                  models/latent_diffusion/infer_ldm.py:130-178
  * output packing / NPZ post-processing                   scripts/trainer.py:524-526, models/diffusion/viz/visualizer.py:344-364

Pinning status
  * Networks: PINNED.  tests/golden/*.npz were produced by the reference's own ``Denoiser`` and
    ``MotionPrior`` classes (imported in the build container by oracle/gen_golden.py) and
    tests/test_oracle_golden.py checks this restatement against them.
  * ``rotation_6d_to_matrix``, ``quaternion_to_axis_angle`` and the *legacy* ``matrix_to_quaternion``:
    PINNED against the pytorch3d snapshot vendored at models/diffusion/utils/rotation_conversions.py.
  * Third-party arithmetic that is NOT under /root/reference (diffusers==0.17.1, amuse.yml:143; pytorch3d, unpinned,
    README.md:105) - restated from the published algorithms and PINNED against what the reference tree itself holds:
      - ``matrix_to_axis_angle`` with the candidate-selection ``matrix_to_quaternion``: PINNED by the ``poses`` of the three
        committed sample outputs viz_dump/test/**/*_motion_smplx.npz (tests/golden/ref_poses.npz; outputs of the deployed
        pytorch3d at infer_ldm.py:171-172): 49,500 joints reproduce to <= 1e-5 through axis_angle_to_matrix ->
        matrix_to_axis_angle(.., "p3d"), including all 10 with |aa| > pi, which the vendored ("legacy") variant cannot produce.
      - ``DDPMScheduler`` (``add_noise``; ancestral ``step`` with fixed_small, no clip, 1000 steps): PINNED against the
        reference tree's own GaussianDiffusion (models/diffusion/utils/mdm_gaussian_diffusion.py:198-278,323-366,528-533,690;
        tests/golden/sched_ref.npz): posterior tables for all 1000 t, single-step known answers, a 1000-step trajectory.
      - ``DDIMScheduler`` (eta 0, steps_offset 1, 50 steps): the update is PINNED against SpacedDiffusion + ddim_sample
        (mdm_respace.py:64-87, mdm_gaussian_diffusion.py:895-940) on {1,21,..,981}: all 50 steps for
        set_alpha_to_one=True, 49 of 50 for the reference's set_alpha_to_one=False, without clipping, eta = 0 and eta = 0.5; and with clipping for
        use_clipped_model_output=True.  TWO conventions remain recollections of the diffusers 0.17.1 source, unpinned:
        (i) set_alpha_to_one=False -> the last step's alpha_bar_prev is alphas_cumprod[0]; (ii) with clip_sample=True
        (the default the reference inherits) and use_clipped_model_output=False the direction term uses the UN-clipped
        eps_hat.  Also not reference-held: torch-fp32 ``cumprod`` for alphas_cumprod (the reference tree's code uses fp64
        numpy; the difference is <= 7e-6 relative on every coefficient, tests/test_pins_cpu.py).

Everything is written with explicit tensor math (no nn.Module) so that ``emulate_bf16=True`` can round
exactly the operands the bf16 HIP kernels round (MFMA A/B inputs), giving a tight checker for the
bf16 path as well as for the fp32 path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

D, H, DH, FF, L, C = 128, 4, 32, 512, 9, 256
NSKIP = 4
N_FRAMES, N_JOINTS, N_FEATS = 300, 55, 333


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def to_torch(w: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in w.items()}


class Ops:
    """Arithmetic policy: fp32/fp64 exact, or bf16-operand emulation of the MFMA GEMMs."""

    def __init__(self, emulate_bf16: bool = False, poly_gelu: bool = False, fp16: bool = False):
        self.emu = emulate_bf16
        self.poly_gelu = poly_gelu   # model of the HIP 16-bit modes' FFN activation (gelu_poly below)
        self.fp16 = fp16             # emulate the fp16 throughput mode (AMUSE_PREC_F16) instead of the bf16 one: operands rounded to
        #                              fp16, GELU polynomial one degree higher (amuse_dev.hpp gelu_poly4h)

    def act(self, x: torch.Tensor) -> torch.Tensor:
        return gelu_poly(x, _GELU_POLY_H if self.fp16 else _GELU_POLY) if self.poly_gelu else gelu(x)

    def r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.float16 if self.fp16 else torch.bfloat16).to(x.dtype) if self.emu else x

    def lin(self, x, w, b=None):
        y = self.r(x) @ self.r(w).transpose(-1, -2)
        return y if b is None else y + b

    def mm(self, a, b):
        return self.r(a) @ self.r(b)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu(x):  # exact erf form, F.gelu default (cross_attention.py:408-409)
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


_GELU_POLY = (7.975201607e-01, -1.319021881e-01, 1.900408231e-02, -1.993848477e-03, 1.449597330e-04,
              -6.815091183e-06, 1.840825661e-07, -2.152084733e-09)


# the fp16 mode's polynomial (tools/fit_gelu_poly.py 8 3.0), lowest power first
_GELU_POLY_H = (7.978953719e-01, -1.328561008e-01, 1.973768137e-02, -2.249475103e-03, 1.923068630e-04, -1.179083301e-05, 4.817333092e-07,
                -1.159289287e-08, 1.232038360e-10)


def gelu_poly(x, coeffs=None):
    """The bf16 sampling kernel's GELU (amuse_amd/csrc/amuse_dev.hpp gelu_poly4), restated so that the block-wise
    bf16 emulation sees the operands the kernel rounds: erf(a / sqrt2) ~ a P(a^2), a = clamp(x, +-3 sqrt2), degree-7
    minimax P (|erf error| <= 8.7e-5, value at the clamp point pinned to 1).  NOT the reference's activation - that is
    gelu() above, which every fp32 parity check uses."""
    x = x.float()
    a = torch.clamp(x, min=-4.24264068711928514641, max=4.24264068711928514641)
    s2 = a * a
    coeffs = _GELU_POLY if coeffs is None else coeffs
    p = torch.full_like(x, coeffs[-1])
    for c in coeffs[-2::-1]:
        p = p * s2 + c
    hx = 0.5 * x
    return hx * (a * p) + hx


def mha_self(ops: Ops, x, W, p, key_mask: Optional[torch.Tensor] = None):
    """nn.MultiheadAttention(128, 4) with q = k = v = x, batch-first (B,S,D).
    key_mask: (B,S) bool, True = key is valid (the reference passes key_padding_mask = ~mask)."""
    B, S, _ = x.shape
    qkv = ops.lin(x, W[p + ".in_proj_weight"], W[p + ".in_proj_bias"])
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    q = q * math.sqrt(1.0 / DH)
    sh = lambda t: t.reshape(B, S, H, DH).permute(0, 2, 1, 3)  # (B,H,S,dh)
    q, k, v = sh(q), sh(k), sh(v)
    s = ops.mm(q, k.transpose(-1, -2))
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = ops.mm(a, v).permute(0, 2, 1, 3).reshape(B, S, D)
    return ops.lin(o, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def enc_block(ops, x, W, p, key_mask=None):
    """TransformerEncoderLayer.forward_post (cross_attention.py:259-272); dropout = identity in eval."""
    x = layer_norm(x + mha_self(ops, x, W, p + ".self_attn", key_mask), W[p + ".norm1.weight"], W[p + ".norm1.bias"])
    h = ops.act(ops.lin(x, W[p + ".linear1.weight"], W[p + ".linear1.bias"]))
    x = layer_norm(x + ops.lin(h, W[p + ".linear2.weight"], W[p + ".linear2.bias"]),
                   W[p + ".norm2.weight"], W[p + ".norm2.bias"])
    return x


def cross_attn_const(ops, z, W, p):
    """Cross-attention onto a ONE-token memory (cross_attention.py:331-336): softmax over a single key
    is 1, so the result is out_proj(v_proj(z)) for every query row.  z: (B,D) -> (B,D)."""
    wv, bv = W[p + ".in_proj_weight"][2 * D:], W[p + ".in_proj_bias"][2 * D:]
    return ops.lin(ops.lin(z, wv, bv), W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def dec_block(ops, x, z, W, p, key_mask=None):
    """TransformerDecoderLayer.forward_post (cross_attention.py:323-345) with a 1-token memory z."""
    x = layer_norm(x + mha_self(ops, x, W, p + ".self_attn", key_mask), W[p + ".norm1.weight"], W[p + ".norm1.bias"])
    ca = cross_attn_const(ops, z, W, p + ".multihead_attn")
    x = layer_norm(x + ca[:, None, :], W[p + ".norm2.weight"], W[p + ".norm2.bias"])
    h = ops.act(ops.lin(x, W[p + ".linear1.weight"], W[p + ".linear1.bias"]))
    x = layer_norm(x + ops.lin(h, W[p + ".linear2.weight"], W[p + ".linear2.bias"]),
                   W[p + ".norm3.weight"], W[p + ".norm3.bias"])
    return x


def mha(ops: Ops, xq, xkv, W, p):
    """nn.MultiheadAttention(128, 4)(query = xq, key = value = xkv), batch-first, no masks: xq (B,Sq,D), xkv (B,Sk,D)."""
    B, Sq, _ = xq.shape
    Sk = xkv.shape[1]
    w, b = W[p + ".in_proj_weight"], W[p + ".in_proj_bias"]
    q = ops.lin(xq, w[:D], b[:D]) * math.sqrt(1.0 / DH)
    k = ops.lin(xkv, w[D:2 * D], b[D:2 * D])
    v = ops.lin(xkv, w[2 * D:], b[2 * D:])
    sh = lambda t, S: t.reshape(B, S, H, DH).permute(0, 2, 1, 3)
    a = torch.softmax(ops.mm(sh(q, Sq), sh(k, Sk).transpose(-1, -2)), dim=-1)
    o = ops.mm(a, sh(v, Sk)).permute(0, 2, 1, 3).reshape(B, Sq, D)
    return ops.lin(o, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def dec_block_mem(ops, x, mem, W, p):
    """TransformerDecoderLayer.forward_post (cross_attention.py:323-345) with a multi-token memory and no masks / positional
    arguments - how TransformerDecoder.forward (cross_attention.py:205-234) calls it from Denoiser(arch="trans_dec")."""
    x = layer_norm(x + mha(ops, x, x, W, p + ".self_attn"), W[p + ".norm1.weight"], W[p + ".norm1.bias"])
    x = layer_norm(x + mha(ops, x, mem, W, p + ".multihead_attn"), W[p + ".norm2.weight"], W[p + ".norm2.bias"])
    h = ops.act(ops.lin(x, W[p + ".linear1.weight"], W[p + ".linear1.bias"]))
    return layer_norm(x + ops.lin(h, W[p + ".linear2.weight"], W[p + ".linear2.bias"]), W[p + ".norm3.weight"], W[p + ".norm3.bias"])


def skip_stack(ops, x, W, p, block_fn, taps: Optional[dict] = None):
    """SkipTransformerEncoder/Decoder.forward wiring (cross_attention.py:41-64, 89-125)."""
    xs: List[torch.Tensor] = []
    for i in range(NSKIP):
        x = block_fn(x, f"{p}.input_blocks.{i}")
        if taps is not None:
            taps[f"{p}.input_blocks.{i}"] = x
        xs.append(x)
    x = block_fn(x, f"{p}.middle_block")
    if taps is not None:
        taps[f"{p}.middle_block"] = x
    for i in range(NSKIP):
        x = torch.cat([x, xs.pop()], dim=-1)
        x = ops.lin(x, W[f"{p}.linear_blocks.{i}.weight"], W[f"{p}.linear_blocks.{i}.bias"])
        x = block_fn(x, f"{p}.output_blocks.{i}")
        if taps is not None:
            taps[f"{p}.output_blocks.{i}"] = x
    return layer_norm(x, W[f"{p}.norm.weight"], W[f"{p}.norm.bias"])


# --------------------------------------------------------------------------------------------
# Denoiser
# --------------------------------------------------------------------------------------------
def timestep_freqs() -> torch.Tensor:
    """get_timestep_embedding frequencies (embeddings.py:262-267), always float32 as in the reference."""
    half = C // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - 0)  # downscale_freq_shift = 0
    return torch.exp(exponent)


def timestep_sinusoid(t: int, dtype=torch.float32) -> torch.Tensor:
    """(256,) [cos | sin] - flip_sin_to_cos = true (embeddings.py:269-279)."""
    emb = torch.tensor(float(t), dtype=torch.float32) * timestep_freqs()
    if dtype == torch.float64:
        emb = emb.double()
    return torch.cat([torch.cos(emb), torch.sin(emb)]).to(dtype)


def time_embed(W, t: int, dtype=torch.float32) -> torch.Tensor:
    """TimestepEmbedding: Linear -> SiLU -> Linear (embeddings.py:288-305).  Always full precision
    (the HIP path computes this table once per schedule in fp32)."""
    e = timestep_sinusoid(t, dtype)
    h = e @ W["time_embedding.linear_1.weight"].T + W["time_embedding.linear_1.bias"]
    h = h * torch.sigmoid(h)
    return h @ W["time_embedding.linear_2.weight"].T + W["time_embedding.linear_2.bias"]


def cond_project(W, name: str, z: torch.Tensor) -> torch.Tensor:
    """emb_proj_*: Linear(ReLU(z)) - ReLU first, on the raw embedding (denoiser.py:74-79)."""
    return torch.relu(z) @ W[f"emb_proj_{name}.1.weight"].T + W[f"emb_proj_{name}.1.bias"]


def denoiser_tokens(W, x, t, con, emo, sty) -> torch.Tensor:
    """Token assembly (denoiser.py:144-181): [latent, time, con, (emo), (sty)] + learned PE.  (B,S,128).
    t: one int for the whole batch (sampling) or a sequence of B ints (training-time diffusion_forward)."""
    B = x.shape[0]
    if isinstance(t, (int, np.integer)):
        te = time_embed(W, int(t), x.dtype)[None].expand(B, -1)
    else:
        te = torch.stack([time_embed(W, int(ti), x.dtype) for ti in t])
    toks = [x, te, cond_project(W, "con", con)]
    if emo is not None:
        toks.append(cond_project(W, "emo", emo))
    if sty is not None:
        toks.append(cond_project(W, "sty", sty))
    xs = torch.stack(toks, dim=1)
    return xs + W["query_pos.pe"][: xs.shape[1], 0][None]


def denoiser_forward(W, x, t, con, emo, sty, emulate_bf16=False, taps: Optional[dict] = None, fp16=False):
    """eps_hat = Denoiser(x_t, t, con, emo, sty).  x: (B,128), con/emo/sty: (B,256) or None -> (B,128).
    emulate_bf16 models the HIP bf16 sampling kernel: bf16 GEMM operands and its polynomial FFN activation; with fp16 = True
    the fp16 throughput mode instead (fp16 operands, its polynomial)."""
    ops = Ops(emulate_bf16 or fp16, poly_gelu=emulate_bf16 or fp16, fp16=fp16)
    xs = denoiser_tokens(W, x, t, con, emo, sty)
    if taps is not None:
        taps["tokens"] = xs
    out = skip_stack(ops, xs, W, "encoder", lambda h, p: enc_block(ops, h, W, p), taps)
    return out[:, 0]


def denoiser_memory(W, B, t, con, emo, sty, dtype=torch.float32) -> torch.Tensor:
    """emb_latent of Denoiser.forward (denoiser.py:146-174): [time, con, (emo), (sty)] -> (B, 2..4, 128), no positions added."""
    if isinstance(t, (int, np.integer)):
        te = time_embed(W, int(t), dtype)[None].expand(B, -1)
    else:
        te = torch.stack([time_embed(W, int(ti), dtype) for ti in t])
    toks = [te, cond_project(W, "con", con)]
    if emo is not None:
        toks.append(cond_project(W, "emo", emo))
    if sty is not None:
        toks.append(cond_project(W, "sty", sty))
    return torch.stack(toks, dim=1)


def denoiser_forward_variant(W, x, t, con, emo, sty, arch="trans_enc", diffusion_only=False, lengths=None,
                             emulate_bf16=False, fp16=False, taps: Optional[dict] = None):
    """Denoiser.forward for the variants the shipped configuration does not reach (denoiser.py:174-204).
      arch "trans_enc", diffusion_only: x (B,300,333) -> pose_embd -> [time, con, emo, sty | 300 frames] + query_pos -> skip encoder
                                        (no key mask: the padded frames are attended, denoiser.py:182) -> pose_proj of the frame rows;
      arch "trans_dec":                 tgt = x (B,128) as ONE token (or pose_embd of the 300 frames) + query_pos, memory = the 2..4
                                        condition tokens + mem_pos; 9 x TransformerDecoderLayer, decoder.norm (+ pose_proj).
    With diffusion_only the rows of frames >= lengths[b] are zeroed (`sample[~mask.T] = 0`, denoiser.py:187,199)."""
    ops = Ops(emulate_bf16 or fp16, poly_gelu=emulate_bf16 or fp16, fp16=fp16)
    B = x.shape[0]
    mem = denoiser_memory(W, B, t, con, emo, sty, x.dtype)
    nmem = mem.shape[1]
    if diffusion_only:
        h = ops.lin(x, W["pose_embd.weight"], W["pose_embd.bias"])            # (B,300,128)
    else:
        h = x[:, None, :]                                                     # (B,1,128)
    if arch == "trans_enc":
        assert diffusion_only, "the latent trans_enc configuration is denoiser_forward()"
        xs = torch.cat([mem, h], dim=1)
        xs = xs + W["query_pos.pe"][: xs.shape[1], 0][None]
        if taps is not None:
            taps["tokens"] = xs
        out = skip_stack(ops, xs, W, "encoder", lambda a, p: enc_block(ops, a, W, p), taps)[:, nmem:]
    elif arch == "trans_dec":
        h = h + W["query_pos.pe"][: h.shape[1], 0][None]
        mem = mem + W["mem_pos.pe"][:nmem, 0][None]
        if taps is not None:
            taps["tokens"], taps["memory"] = h, mem
        for i in range(L):
            h = dec_block_mem(ops, h, mem, W, f"decoder.layers.{i}")
            if taps is not None:
                taps[f"decoder.layers.{i}"] = h
        out = layer_norm(h, W["decoder.norm.weight"], W["decoder.norm.bias"])
    else:
        raise ValueError(arch)
    if not diffusion_only:
        return out[:, 0]
    out = ops.lin(out, W["pose_proj.weight"], W["pose_proj.bias"])
    if lengths is not None:
        keep = torch.arange(out.shape[1])[None, :] < torch.tensor(list(lengths))[:, None]
        out = out * keep[..., None].to(out.dtype)
    return out


def sample_variant(W, sched, con, emo, sty, x_init, arch, diffusion_only, step_noise=None, traj: Optional[list] = None):
    """The sampling loop of infer_ldm.py:137-161 around a variant denoiser; x_init (B,128) or (B,300,333)."""
    x = x_init * sched.init_noise_sigma
    for i, t in enumerate(sched.timesteps):
        eps = denoiser_forward_variant(W, x, t, con, emo, sty, arch, diffusion_only)
        nz = step_noise[i] if (step_noise is not None and sched.needs_noise(t)) else None
        x = sched.step(eps, t, x, nz)
        if traj is not None:
            traj.append(x.clone())
    return x


# --------------------------------------------------------------------------------------------
# Schedulers (diffusers 0.17.1 semantics, restated; pinned against the reference tree's GaussianDiffusion, see header)
# --------------------------------------------------------------------------------------------
class SchedulerBase:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.n_train = num_train_timesteps
        # beta_schedule == "scaled_linear"
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0


class DDIM(SchedulerBase):
    """DDIMScheduler as constructed at infer_ldm.py:116-123: clip_sample is NOT passed -> default True."""

    def __init__(self, num_inference_steps=50, steps_offset=1, set_alpha_to_one=False, eta=0.0,
                 clip_sample=True, clip_sample_range=1.0, use_clipped_model_output=False, **kw):
        super().__init__(**kw)
        self.use_clipped = use_clipped_model_output   # diffusers' step() argument; the reference leaves it False
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.eta, self.clip, self.clip_range = eta, clip_sample, clip_sample_range
        self.n_inf = num_inference_steps
        ratio = self.n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + steps_offset
        self.timesteps = [int(v) for v in ts]

    def step(self, eps, t: int, x, noise=None):
        prev_t = t - self.n_train // self.n_inf
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        cast = lambda v: v.to(x.dtype)
        x0 = (x - cast(b_t ** 0.5) * eps) / cast(a_t ** 0.5)
        if self.clip:
            x0 = x0.clamp(-self.clip_range, self.clip_range)
        if self.use_clipped:
            eps = (x - cast(a_t ** 0.5) * x0) / cast(b_t ** 0.5)
        var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std = self.eta * var ** 0.5
        direction = cast((1 - a_prev - std ** 2) ** 0.5) * eps
        prev = cast(a_prev ** 0.5) * x0 + direction
        if self.eta > 0:
            prev = prev + cast(std) * noise
        return prev

    def needs_noise(self, t: int) -> bool:
        return self.eta > 0


class DDPM(SchedulerBase):
    """Ancestral sampler following the training-side DDPMScheduler config (ldm.py:41-49;
    configs/diff_latent_v2.json:48-56: fixed_small, clip_sample false, epsilon)."""

    def __init__(self, num_inference_steps=None, **kw):
        super().__init__(**kw)
        self.n_inf = num_inference_steps or self.n_train
        ratio = self.n_train // self.n_inf
        self.timesteps = [int(v) for v in (np.arange(0, self.n_inf) * ratio).round()[::-1].astype(np.int64)]

    def step(self, eps, t: int, x, noise=None):
        prev_t = t - self.n_train // self.n_inf
        one = torch.tensor(1.0)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else one
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        cast = lambda v: v.to(x.dtype)
        x0 = (x - cast(b_t ** 0.5) * eps) / cast(a_t ** 0.5)
        c0 = (a_prev ** 0.5 * cur_b) / b_t
        cx = cur_a ** 0.5 * b_prev / b_t
        prev = cast(c0) * x0 + cast(cx) * x
        if t > 0:
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
            prev = prev + cast(var ** 0.5) * noise
        return prev

    def needs_noise(self, t: int) -> bool:
        return t > 0


# --------------------------------------------------------------------------------------------
# Counter-based noise (the BUILD's own contract; restated here so the checker can reproduce it)
#   Philox4x32-10, key = (seed_lo, seed_hi), counter = (clip, step, feature/4, stream)
#   stream 0 = initial latent (step field = 0), stream 1 = per-step ancestral noise
#   u = (x >> 8) * 2^-24 + 2^-25 ; Box-Muller: (u0,u1)->(z0,z1), (u2,u3)->(z2,z3)
# --------------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2511F53, 0xCD9E8D57
_PH_W0, _PH_W1 = 0x9E3779B9, 0xBB67AE85
_M32 = 0xFFFFFFFF


def philox4x32_10(ctr: np.ndarray, key: Sequence[int]) -> np.ndarray:
    """ctr: (...,4) uint64-held uint32 values -> (...,4)."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0] & _M32), np.uint64(key[1] & _M32)
    for _ in range(10):
        p0 = np.uint64(_PH_M0) * c[0]
        p1 = np.uint64(_PH_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(_M32)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(_M32)
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(_PH_W0)) & np.uint64(_M32)
        k1 = (k1 + np.uint64(_PH_W1)) & np.uint64(_M32)
    return np.stack(c, axis=-1)


def counter_normal(seed: int, clips: np.ndarray, step: int, stream: int, nfeat: int = D) -> np.ndarray:
    """(len(clips), nfeat) float32 standard normals."""
    clips = np.asarray(clips, dtype=np.uint64)
    q = np.arange(nfeat // 4, dtype=np.uint64)
    ctr = np.zeros((len(clips), nfeat // 4, 4), dtype=np.uint64)
    ctr[..., 0] = clips[:, None]
    ctr[..., 1] = np.uint64(step)
    ctr[..., 2] = q[None, :]
    ctr[..., 3] = np.uint64(stream)
    r = philox4x32_10(ctr, (seed & _M32, (seed >> 32) & _M32))
    u = (r >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)
    rad0 = np.sqrt(np.float32(-2.0) * np.log(u[..., 0]))
    rad1 = np.sqrt(np.float32(-2.0) * np.log(u[..., 2]))
    th0 = np.float32(2.0 * math.pi) * u[..., 1]
    th1 = np.float32(2.0 * math.pi) * u[..., 3]
    z = np.stack([rad0 * np.cos(th0), rad0 * np.sin(th0), rad1 * np.cos(th1), rad1 * np.sin(th1)], axis=-1)
    return z.reshape(len(clips), nfeat).astype(np.float32)


# --------------------------------------------------------------------------------------------
# Sampling loop (infer_ldm.py:130-161)
# --------------------------------------------------------------------------------------------
def sample_latents(W, sched, con, emo, sty, x_init, step_noise=None, emulate_bf16=False,
                   traj: Optional[list] = None):
    """x_init: (B,128) explicit initial noise.  step_noise: (T,B,128) or None (required when the
    scheduler draws noise).  Returns final latents (B,128)."""
    x = x_init * sched.init_noise_sigma
    for i, t in enumerate(sched.timesteps):
        eps = denoiser_forward(W, x, t, con, emo, sty, emulate_bf16)
        nz = step_noise[i] if (step_noise is not None and sched.needs_noise(t)) else None
        x = sched.step(eps, t, x, nz)
        if traj is not None:
            traj.append(x.clone())
    return x


# --------------------------------------------------------------------------------------------
# VAE decode (vae.py:216-278, encoder_decoder arch, learned PE) and rotation conversions
# --------------------------------------------------------------------------------------------
def vae_decode(Wp, z, lengths: Optional[Sequence[int]] = None, emulate_bf16=False, taps: Optional[dict] = None, fp16=False):
    """z: (B,128) -> feats (B,300,333).  Frames >= length are excluded as keys and zeroed on output."""
    ops = Ops(emulate_bf16 or fp16, poly_gelu=emulate_bf16 or fp16, fp16=fp16)   # the HIP 16-bit modes: rounded GEMM operands + polynomial GELU
    B = z.shape[0]
    if lengths is None:
        lengths = [N_FRAMES] * B
    n = max(lengths)
    mask = torch.arange(n)[None, :] < torch.tensor(list(lengths))[:, None]  # (B,n) True = valid
    x = torch.zeros(B, n, D, dtype=z.dtype) + Wp["query_pos_decoder.pe"][:n, 0][None]
    km = None if bool(mask.all()) else mask
    x = skip_stack(ops, x, Wp, "decoder", lambda h, p: dec_block(ops, h, z, Wp, p, km), taps)
    feats = ops.lin(x, Wp["final_layer.weight"], Wp["final_layer.bias"])
    return feats * mask[..., None].to(feats.dtype)


def vae_encode(Wp, feats, lengths: Optional[Sequence[int]] = None, emulate_bf16=False, fp16=False):
    """MotionPrior.encode (vae.py:154-214, MLP_DIST false, learned PE): feats (B,300,333) -> (mu, std), each (B,128).
    xseq = [2 distribution tokens | skel_embedding(frames)] + PE; SkipTransformerEncoder with key padding mask;
    mu = token 0, logvar = token 1, std = exp(logvar) ** 0.5.  (latent = mu + std * eps is left to the caller.)"""
    ops = Ops(emulate_bf16 or fp16, poly_gelu=emulate_bf16 or fp16, fp16=fp16)   # the HIP 16-bit modes: rounded GEMM operands + polynomial GELU
    B, n, _ = feats.shape
    if lengths is None:
        lengths = [n] * B
    mask = torch.arange(n)[None, :] < torch.tensor(list(lengths))[:, None]
    aug = torch.cat([torch.ones(B, 2, dtype=torch.bool), mask], dim=1)           # (B, 2 + n)
    x = ops.lin(feats, Wp["skel_embedding.weight"], Wp["skel_embedding.bias"])
    xseq = torch.cat([Wp["global_motion_token"][None].expand(B, -1, -1).to(x.dtype), x], dim=1)
    xseq = xseq + Wp["query_pos_encoder.pe"][: n + 2, 0][None]
    km = None if bool(aug.all()) else aug
    out = skip_stack(ops, xseq, Wp, "encoder", lambda h, p: enc_block(ops, h, Wp, p, km))
    mu, logvar = out[:, 0], out[:, 1]
    return mu, logvar.exp().pow(0.5)


def axis_angle_to_rotation_6d(aa):
    """infer_ldm.py:459-463: axis_angle_to_matrix then matrix_to_rotation_6d (first two rows)."""
    m = axis_angle_to_matrix(aa)
    return m[..., :2, :].reshape(*m.shape[:-2], 6)


def rotation_6d_to_matrix(d6):
    """pytorch3d rotation_6d_to_matrix (vendored copy: models/diffusion/utils/rotation_conversions.py:512-533)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    nrm = lambda v: v / torch.clamp(torch.linalg.vector_norm(v, dim=-1, keepdim=True), min=1e-12)
    b1 = nrm(a1)
    b2 = nrm(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def _sqrt_pos(x):
    return torch.sqrt(torch.clamp(x, min=0.0))


def matrix_to_quaternion(m, mode: str = "p3d"):
    """mode "legacy": vendored snapshot (rotation_conversions.py:97-119), q_w >= 0 always.
    mode "p3d": pytorch3d >= 0.5 candidate selection WITHOUT sign standardisation (see header)."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    if mode == "legacy":
        o0 = 0.5 * _sqrt_pos(1 + m00 + m11 + m22)
        x = 0.5 * _sqrt_pos(1 + m00 - m11 - m22)
        y = 0.5 * _sqrt_pos(1 - m00 + m11 - m22)
        z = 0.5 * _sqrt_pos(1 - m00 - m11 + m22)
        cs = lambda a, b: torch.where((a < 0) != (b < 0), -a, a)
        return torch.stack((o0, cs(x, m21 - m12), cs(y, m02 - m20), cs(z, m10 - m01)), -1)
    q_abs = _sqrt_pos(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                   1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * torch.clamp(q_abs[..., None], min=0.1))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(*idx.shape, 1, 4)).squeeze(-2)


def quaternion_to_axis_angle(q):
    """rotation_conversions.py:480-509."""
    norms = torch.linalg.vector_norm(q[..., 1:], dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    return q[..., 1:] / s


def matrix_to_axis_angle(m, mode: str = "p3d"):
    return quaternion_to_axis_angle(matrix_to_quaternion(m, mode))


def axis_angle_to_matrix(aa):
    """Used by tests to compare poses *as rotations* (rotation_conversions.py:425-478 composition)."""
    ang = torch.linalg.vector_norm(aa, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    s = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    q = torch.cat([torch.cos(half), aa * s], dim=-1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def diffusion_forward(W, z0, noise, timesteps, con, emo, sty, emulate_bf16=False):
    """LatentDiffusionModel.diffusion_forward (ldm.py:71-115) in eval semantics (no dropout), with the caller's noise
    and per-sample timesteps: noisy = sqrt(abar_t) z0 + sqrt(1 - abar_t) noise (DDPMScheduler.add_noise, diffusers
    0.17.1; pinned against the reference tree's q_sample, tests/test_pins_cpu.py), noise_pred = Denoiser(noisy, t, cond).
    -> {"noisy_latents", "noise", "noise_pred"}, each (B,128)."""
    ac = SchedulerBase().alphas_cumprod
    t = torch.as_tensor(list(timesteps), dtype=torch.long)
    sa, sb = ac[t].sqrt()[:, None], (1.0 - ac[t]).sqrt()[:, None]
    noisy = sa * z0 + sb * noise
    return {"noisy_latents": noisy, "noise": noise,
            "noise_pred": denoiser_forward(W, noisy, [int(v) for v in t], con, emo, sty, emulate_bf16)}


def feats_to_smplx(feats, mode: str = "p3d"):
    """infer_ldm.py:168-173: split 330|3, 6D -> matrix -> axis-angle.  -> poses (B,T,55,3), trans (B,T,3)."""
    rot6d, trans = feats[..., :-3], feats[..., -3:]
    rot6d = rot6d.reshape(*rot6d.shape[:-1], N_JOINTS, 6)
    return matrix_to_axis_angle(rotation_6d_to_matrix(rot6d), mode), trans


def diffusion_backward(Wd, Wp, sched, con, emo, sty, x_init, step_noise=None, emulate_bf16=False,
                       quat_mode: str = "p3d", emulate_bf16_decode: Optional[bool] = None):
    """PretrainedLPDM_v1.diffusion_backward (infer_ldm.py:130-178) with explicit noise."""
    lat = sample_latents(Wd, sched, con, emo, sty, x_init, step_noise, emulate_bf16)
    edec = emulate_bf16 if emulate_bf16_decode is None else emulate_bf16_decode
    feats = vae_decode(Wp, lat, None, edec)
    poses, trans = feats_to_smplx(feats, quat_mode)
    return {"latents": lat, "feats": feats, "poses": poses, "trans": trans}


def pack_feats(poses, trans):
    """trainer.py:524-526: "b t j d -> b t (j d)" then cat trans -> (B,300,168)."""
    return torch.cat([poses.reshape(*poses.shape[:2], -1), trans], dim=-1)


LOWER_BODY = [1, 2, 4, 5, 7, 8, 10, 11]


def npz_fields(feat_168: np.ndarray, gender: str = "neutral", betas: Optional[np.ndarray] = None, fps: float = 30.0):
    """CaMNVisualizer.animate_ldm_sample_v1 SMPL-X branch (visualizer.py:344-364): (300,168) -> NPZ fields."""
    f = np.array(feat_168, dtype=np.float32).reshape(feat_168.shape[0], -1, 3)
    if f.shape[1] == 56:
        f = f[:, :-1, :]
    f[:, LOWER_BODY, :] = f[0, LOWER_BODY, :]
    return {"poses": f, "trans": np.zeros((f.shape[0], 3)), "gender": np.array(gender),
            "betas": np.zeros(300) if betas is None else betas, "mocap_frame_rate": np.array(fps, dtype="float64")}
