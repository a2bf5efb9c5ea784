"""A transformer layer of the train_gesture step as ONE autograd.Function (BASELINE config 4; reference scripts/trainer.py:335-498 runs
utils/cross_attention.py:259-272 (TransformerEncoderLayer.forward_post) and :323-345 (TransformerDecoderLayer.forward_post) op by op under
autograd): the GEMMs stay on rocBLAS (torch.mm / addmm) and the self-attention on the vendor's fused kernel (aten's efficient-attention
forward / backward ops, called directly), everything between them - biases, the three dropouts, residual adds, LayerNorms, GELU and every
bias / LayerNorm gradient reduction - runs in the hand-written HIP kernels of csrc/k_train.hip, forward and backward.

Why a Function per layer and not per op: the eager step was host-bound AND device-bound at once (~1,950 launches, ~27 ms of host
dispatch over ~24 ms of device time per iteration).  Inside `forward` / `backward` nothing is recorded by autograd, so a layer costs ~8 + ~18
launches with no graph nodes in between instead of ~14 + ~25 with one node each, and the glue's device time (LayerNorm forward / backward, dropout,
adds, GELU, the bias gradients' column sums: ~9 of the 24 ms) shrinks to one pass over each array.

Dropout: counter-based masks (k_train.hip), keyed by torch's seed of the process (`torch.initial_seed()`: the trainer seeds every rank
differently) and a host-side call counter - nothing stored, the backward kernels regenerate them.  The draws differ from nn.Dropout's (another
generator); the distribution is the same.  Attention dropout stays inside the vendor kernel (its own Philox state, returned and replayed).

The eager layers of nn_modules.py remain the definition (CPU, key-padding masks, AMUSE_TRAIN_FUSED=0); tests/test_gpu_train_ops.py pins this
path to them: outputs and every gradient, eval mode exactly the same arithmetic, train mode through the masks.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib

_STATE = {}
_OFFSET = [0]


def enabled() -> bool:
    return os.environ.get("AMUSE_TRAIN_FUSED", "1") != "0"


def usable(x: torch.Tensor, mask) -> bool:
    """The fused path takes CUDA fp32 activations of width 128 without a key-padding mask (the training batches are full length)."""
    return enabled() and x.is_cuda and x.dtype == torch.float32 and mask is None and x.shape[-1] == 128


def _st(device):
    s = _STATE.get(device)
    if s is None:
        lib = _lib.load()
        s = {"lib": lib, "ws": torch.empty(int(lib.amuse_train_ws_floats()), device=device, dtype=torch.float32)}
        _STATE[device] = s
    return s


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _seed() -> int:
    return torch.initial_seed() & 0x7FFFFFFFFFFFFFFF


def next_offset() -> int:
    _OFFSET[0] += 1
    return _OFFSET[0]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _c(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.float32
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------- raw kernels
def ln_fwd(x, y, bias, gamma, beta, p: float, seed: int, off: int, keep: bool = True):
    """LayerNorm(x + dropout(y + bias)) over rows of 128 -> (out, zhat, rstd); zhat / rstd None unless `keep`."""
    y = _c(y)
    rows = y.numel() // 128
    s = _st(y.device)
    out = torch.empty_like(y)
    zhat = torch.empty_like(y) if keep else None
    rstd = torch.empty(rows, device=y.device, dtype=torch.float32) if keep else None
    _lib.check(s["lib"].amuse_train_ln_fwd(_p(None if x is None else _c(x)), _p(y), _p(bias), _p(gamma), _p(beta), float(p), seed, off, rows,
                                           _p(out), _p(zhat), _p(rstd), _stream()))
    return out, zhat, rstd


def ln_bwd(dout, zhat, rstd, gamma, p: float, seed: int, off: int, need_dx: bool = True, need_dbias: bool = True, dout2=None):
    """LayerNorm backward of dout (+ dout2) -> (dx, dy, dgamma, dbeta, dbias)."""
    dout = _c(dout)
    rows = dout.numel() // 128
    s = _st(dout.device)
    dy = torch.empty_like(dout)
    dx = torch.empty_like(dout) if need_dx else None
    small = torch.empty(3, 128, device=dout.device, dtype=torch.float32)
    _lib.check(s["lib"].amuse_train_ln_bwd(_p(dout), _p(None if dout2 is None else _c(dout2)), _p(zhat), _p(rstd), _p(gamma), float(p), seed, off, rows, _p(dx), _p(dy), small[0].data_ptr(),
                                           small[1].data_ptr(), small[2].data_ptr() if need_dbias else None, _p(s["ws"]), _stream()))
    return dx, dy, small[0], small[1], (small[2] if need_dbias else None)


def bias_gelu_drop_fwd(h, b, p: float, seed: int, off: int):
    h = _c(h)
    F = h.shape[-1]
    out = torch.empty_like(h)
    _lib.check(_st(h.device)["lib"].amuse_train_bias_gelu_drop_fwd(_p(h), _p(b), float(p), seed, off, h.numel() // F, F, _p(out), _stream()))
    return out


def bias_gelu_drop_bwd(da, h, b, p: float, seed: int, off: int):
    da = _c(da)
    F = h.shape[-1]
    s = _st(h.device)
    dh = torch.empty_like(h)
    db = torch.empty(F, device=h.device, dtype=torch.float32)
    _lib.check(s["lib"].amuse_train_bias_gelu_drop_bwd(_p(da), _p(h), _p(b), float(p), seed, off, h.numel() // F, F, _p(dh), _p(db), _p(s["ws"]),
                                                      _stream()))
    return dh, db


def colsum(x):
    x = _c(x)
    C = x.shape[-1]
    s = _st(x.device)
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    _lib.check(s["lib"].amuse_train_colsum(_p(x), x.numel() // C, C, _p(out), _p(s["ws"]), _stream()))
    return out


# ---------------------------------------------------------------------------------------------------- sub-layers (no autograd inside)
_sdpa = torch.ops.aten._scaled_dot_product_efficient_attention
_sdpa_bwd = torch.ops.aten._scaled_dot_product_efficient_attention_backward


def _self_attn_fwd(x2, B, S, H, Win, bin_, Wo, p_attn):
    """x2 (B S, 128) -> (y = attention output through out_proj WITHOUT its bias, saved tensors)."""
    D = x2.shape[1]
    qkv = torch.addmm(bin_, x2, Win.t())                                       # (rows, 3 D): packed in-projection
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, S, 3, H, D // H).unbind(2))   # (B, H, S, d) views
    out, lse, ps, po = _sdpa(q, k, v, None, True, p_attn, False)
    o2 = out.transpose(1, 2).reshape(B * S, D)
    return torch.mm(o2, Wo.t()), (qkv, out, lse, ps, po, o2)


def _self_attn_bwd(dy, x2, B, S, H, Win, Wo, p_attn, saved):
    """dy = gradient of the out_proj output -> (dx contribution as (d_qkv, Win) for the caller's addmm, dWin, dbin, dWo)."""
    qkv, out, lse, ps, po, o2 = saved
    D = x2.shape[1]
    dWo = torch.mm(dy.t(), o2)
    do = torch.mm(dy, Wo).view(B, S, H, D // H).transpose(1, 2)
    q, k, v = (t.transpose(1, 2) for t in qkv.view(B, S, 3, H, D // H).unbind(2))
    dq, dk, dv, _ = _sdpa_bwd(do, q, k, v, None, out, lse, ps, po, p_attn, (True, True, True, False), False)
    dqkv = torch.stack([dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)], dim=2).view(B * S, 3 * D)
    return dqkv, torch.mm(dqkv.t(), x2), colsum(dqkv), dWo


class EncoderLayerFn(torch.autograd.Function):
    """norm2(x1 + dropout2(linear2(dropout(gelu(linear1(x1)))))), x1 = norm1(x + dropout1(self_attn(x)))   (cross_attention.py:259-272)."""

    @staticmethod
    def forward(ctx, x, Win, bin_, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, H, p, p_attn):
        B, S, D = x.shape
        x2 = _c(x).view(B * S, D)
        keep = any(ctx.needs_input_grad)          # (no-grad passes - the iteration's second encode - keep nothing)
        seed, o1, o2_, o3 = _seed(), next_offset(), next_offset(), next_offset()
        y, sa = _self_attn_fwd(x2, B, S, H, Win, bin_, Wo, p_attn)
        x1, zh1, r1 = ln_fwd(x2, y, bo, g1, be1, p, seed, o1, keep)
        h = torch.mm(x1, W1.t())
        a = bias_gelu_drop_fwd(h, b1, p, seed, o2_)
        out, zh2, r2 = ln_fwd(x1, torch.mm(a, W2.t()), b2, g2, be2, p, seed, o3, keep)
        if keep:
            ctx.save_for_backward(x2, *sa, zh1, r1, x1, h, a, zh2, r2, Win, Wo, g1, W1, b1, W2, g2)
            ctx.cfg = (B, S, H, p, p_attn, seed, o1, o2_, o3)
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, dout):
        x2, qkv, ao, lse, ps, po, o2, zh1, r1, x1, h, a, zh2, r2, Win, Wo, g1, W1, b1, W2, g2 = ctx.saved_tensors
        B, S, H, p, p_attn, seed, o1, o2_, o3 = ctx.cfg
        D = x2.shape[1]
        dx1, df, dg2, dbe2, db2 = ln_bwd(dout.reshape(B * S, D), zh2, r2, g2, p, seed, o3)
        dW2 = torch.mm(df.t(), a)
        dh, db1 = bias_gelu_drop_bwd(torch.mm(df, W2), h, b1, p, seed, o2_)
        dW1 = torch.mm(dh.t(), x1)
        dx, dy, dg1, dbe1, dbo = ln_bwd(dx1, zh1, r1, g1, p, seed, o1, dout2=torch.mm(dh, W1))   # (both branches of the residual stream)
        dqkv, dWin, dbin, dWo = _self_attn_bwd(dy, x2, B, S, H, Win, Wo, p_attn, (qkv, ao, lse, ps, po, o2))
        dx = torch.addmm(dx, dqkv, Win)
        return dx.view(B, S, D), dWin, dbin, dWo, dbo, dg1, dbe1, dW1, db1, dW2, db2, dg2, dbe2, None, None, None


class DecoderLayerFn(torch.autograd.Function):
    """TransformerDecoderLayer.forward_post with a memory of ONE token (cross_attention.py:323-345; MotionPrior.decode, vae.py:252-259): self-attention,
    the cross-attention's value path (a softmax over one key is 1: nn_modules.mha_one_key), FFN; three norms.  Wv / bv are the value rows of the
    cross-attention's packed in-projection (sliced by the caller, whose autograd zero-fills the q / k rows' gradients as eager does)."""

    @staticmethod
    def forward(ctx, x, mem, Win, bin_, Wo, bo, g1, be1, Wv, bv, Wc, bc, g2, be2, W1, b1, W2, b2, g3, be3, H, p, p_attn):
        B, S, D = x.shape
        x2 = _c(x).view(B * S, D)
        mem2 = _c(mem).view(B, D)
        keep = any(ctx.needs_input_grad)          # (no-grad passes - the iteration's second encode - keep nothing)
        seed, o1, o2_, o3, o4 = _seed(), next_offset(), next_offset(), next_offset(), next_offset()
        y, sa = _self_attn_fwd(x2, B, S, H, Win, bin_, Wo, p_attn)
        x1, zh1, r1 = ln_fwd(x2, y, bo, g1, be1, p, seed, o1, keep)
        # cross-attention onto the one memory token: value projection, attention dropout on the probability 1 per (clip, query, head), out_proj
        c = torch.addmm(bv, mem2, Wv.t())                                      # (B, D)
        if p_attn > 0:
            kp = torch.nn.functional.dropout(torch.ones(B, S, H, 1, device=x.device, dtype=x.dtype), p_attn, True)
            vk = (kp * c.view(B, 1, H, D // H)).reshape(B * S, D)
        else:
            kp = None
            vk = c[:, None, :].expand(B, S, D).reshape(B * S, D)
        xm, zh2, r2 = ln_fwd(x1, torch.mm(vk, Wc.t()), bc, g2, be2, p, seed, o2_, keep)
        h = torch.mm(xm, W1.t())
        a = bias_gelu_drop_fwd(h, b1, p, seed, o3)
        out, zh3, r3 = ln_fwd(xm, torch.mm(a, W2.t()), b2, g3, be3, p, seed, o4, keep)
        if keep:
            ctx.save_for_backward(x2, *sa, zh1, r1, mem2, vk, zh2, r2, xm, h, a, zh3, r3, Win, Wo, g1, Wv, Wc, g2, W1, b1, W2, g3,
                                  *(() if kp is None else (kp,)))
            ctx.cfg = (B, S, H, p, p_attn, seed, o1, o2_, o3, o4)
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, dout):
        t = ctx.saved_tensors
        x2, qkv, ao, lse, ps, po, o2, zh1, r1, mem2, vk, zh2, r2, xm, h, a, zh3, r3, Win, Wo, g1, Wv, Wc, g2, W1, b1, W2, g3 = t[:28]
        kp = t[28] if len(t) > 28 else None
        B, S, H, p, p_attn, seed, o1, o2_, o3, o4 = ctx.cfg
        D = x2.shape[1]
        dxm, df, dg3, dbe3, db2 = ln_bwd(dout.reshape(B * S, D), zh3, r3, g3, p, seed, o4)
        dW2 = torch.mm(df.t(), a)
        dh, db1 = bias_gelu_drop_bwd(torch.mm(df, W2), h, b1, p, seed, o3)
        dW1 = torch.mm(dh.t(), xm)
        dx1, dyc, dg2, dbe2, dbc = ln_bwd(dxm, zh2, r2, g2, p, seed, o2_, dout2=torch.mm(dh, W1))
        dWc = torch.mm(dyc.t(), vk)
        dvk = torch.mm(dyc, Wc)
        dc = (dvk.view(B, S, H, D // H) * kp).sum(1).view(B, D) if kp is not None else dvk.view(B, S, D).sum(1)
        dWv = torch.mm(dc.t(), mem2)
        dbv = dc.sum(0)
        dmem = torch.mm(dc, Wv)
        dx, dy, dg1, dbe1, dbo = ln_bwd(dx1, zh1, r1, g1, p, seed, o1)
        dqkv, dWin, dbin, dWo = _self_attn_bwd(dy, x2, B, S, H, Win, Wo, p_attn, (qkv, ao, lse, ps, po, o2))
        dx = torch.addmm(dx, dqkv, Win)
        return (dx.view(B, S, D), dmem.view(B, 1, D), dWin, dbin, dWo, dbo, dg1, dbe1, dWv, dbv, dWc, dbc, dg2, dbe2, dW1, db1, dW2, db2, dg3, dbe3,
                None, None, None)


# ---------------------------------------------------------------------------------------------------- module adapters (nn_modules.py)
def encoder_layer(m, x: torch.Tensor) -> torch.Tensor:
    a = m.self_attn
    p = float(m.dropout.p) if m.training else 0.0
    pa = float(a.dropout) if m.training else 0.0
    return EncoderLayerFn.apply(x, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, m.norm1.weight, m.norm1.bias,
                                m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias, m.norm2.weight, m.norm2.bias, a.num_heads, p, pa)


def decoder_layer(m, x: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
    a, c = m.self_attn, m.multihead_attn
    E = x.shape[-1]
    p = float(m.dropout.p) if m.training else 0.0
    pa = float(a.dropout) if m.training else 0.0
    return DecoderLayerFn.apply(x, memory, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, m.norm1.weight, m.norm1.bias,
                                c.in_proj_weight[2 * E:], c.in_proj_bias[2 * E:], c.out_proj.weight, c.out_proj.bias, m.norm2.weight, m.norm2.bias,
                                m.linear1.weight, m.linear1.bias, m.linear2.weight, m.linear2.bias, m.norm3.weight, m.norm3.bias, a.num_heads, p, pa)
