"""A transformer layer of the train_gesture step as ONE autograd.Function (BASELINE config 4; reference scripts/trainer.py:335-498 runs
utils/cross_attention.py:259-272 (TransformerEncoderLayer.forward_post) and :323-345 (TransformerDecoderLayer.forward_post) op by op under
autograd): per direction ONE call into the library (csrc/k_train.hip: amuse_train_layer_fwd / amuse_train_layer_bwd - the packed in-projection, the
self-attention core on the library's fp32 MFMA kernels (csrc/k_train_attn.hip), and the rest of the layer; for attention shapes the library's kernels do not take, aten's
efficient-attention forward / backward ops sit between amuse_train_linear_* and amuse_train_layer_* calls instead).  Inside
the calls the plain GEMMs run on the library's own fp32-MFMA kernels (csrc/k_train_gemm.hip; no vendor BLAS) and everything between them - biases, the three dropouts, residual adds, LayerNorms, GELU,
the decoder's one-key cross-attention and every bias / LayerNorm gradient reduction - runs in the hand-written HIP kernels, forward and backward.

Why a Function per layer and not per op: the eager step was host-bound AND device-bound at once (~1,950 launches, ~27 ms of host
dispatch over ~24 ms of device time per iteration; a torch.mm costs ~19 us of host time where rocblas_sgemm itself takes ~5).  Inside `forward` /
`backward` nothing is recorded by autograd and the ~8 + ~20 launches of a layer are issued from C++, and the glue's device time (LayerNorm forward / backward, dropout,
adds, GELU, the bias gradients' column sums: ~9 of the 24 ms) shrinks to one pass over each array.

Dropout: counter-based masks (k_train.hip), keyed by torch's seed of the process (`torch.initial_seed()`: the trainer seeds every rank
differently) and a host-side call counter - nothing stored, the backward kernels regenerate them.  The draws differ from nn.Dropout's (another
generator); the distribution is the same.  Attention dropout: a hash of (seed, offset, clip, head, query, key) inside the attention kernels (the vendor
kernel, when selected, keeps its own Philox state, returned and replayed).

The eager layers of nn_modules.py remain the definition (CPU, key-padding masks, AMUSE_TRAIN_FUSED=0); tests/test_gpu_train_ops.py pins this
path to them: outputs and every gradient, eval mode exactly the same arithmetic, train mode through the masks.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
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


# ---- lanes: two chains of layers may be in flight on two streams of a device (the trainer issues the Denoiser's forward / backward pass on a stream of its own
# beside the prior's).  Every scratch buffer - this module's column-sum workspace, the library's split-k partials - exists once per LANE; a stream registered with
# register_lane runs on lane 1, every other stream on lane 0.  Autograd runs a node's backward pass on the stream of its forward pass, so both directions of a
# chain land on the chain's lane without the layer functions knowing about it.
_LANES: dict = {}
_TLS = threading.local()


def register_lane(stream: "torch.cuda.Stream", lane: int = 1):
    """`stream` is THE stream of lane `lane` from now on (one at a time: torch hands out streams from a pool of 32 per device, so a handle registered by a
    trainer long gone may come back as somebody's main or capture stream - a stale entry would put two concurrent chains on one lane)."""
    assert lane in (0, 1)
    h = int(stream.cuda_stream)
    if _LANES.get(h) != lane or len(_LANES) != 1:
        _LANES.clear()
        _LANES[h] = lane


def _lane(device) -> int:
    return _LANES.get(torch.cuda.current_stream(device).cuda_stream, 0) if _LANES else 0


class _State(dict):
    """{"lib", "ws" (the CURRENT lane's column-sum workspace), "ws_all"}"""

    def __getitem__(self, k):
        if k == "ws":
            return dict.__getitem__(self, "ws_all")[_lane(dict.__getitem__(self, "device"))]
        return dict.__getitem__(self, k)


def _st(device):
    s = _STATE.get(device)
    if s is None:
        lib = _lib.load()
        n = int(lib.amuse_train_ws_floats())
        s = _State(lib=lib, device=device, ws_all=[torch.empty(n, device=device, dtype=torch.float32) for _ in range(2)])
        _STATE[device] = s
    return s


def _stream(device=None):
    """torch's current stream OF THE TENSORS' DEVICE (not of the process's current device)."""
    return torch.cuda.current_stream(device).cuda_stream


class _on:
    """Device guard around every library call: the C side (rocBLAS handle, workspaces, kernel launches) keys on hipGetDevice, so the tensors' device
    must be the current one - also on the autograd thread and for a single rank on cuda:N, N > 0.  One integer compare when it already is."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = torch.device(device).index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.idx is None:
            self.idx = self.prev
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)
        if _LANES or getattr(_TLS, "lane", 0):      # the library's scratch lane of this thread follows the current stream (one dict lookup; nothing while no lane is registered)
            lane = _LANES.get(torch.cuda.current_stream(self.idx).cuda_stream, 0)
            if lane != getattr(_TLS, "lane", 0):
                _lib.check(_lib.load().amuse_train_set_lane(lane))
                _TLS.lane = lane

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)


def _seed() -> int:
    return torch.initial_seed() & 0x7FFFFFFFFFFFFFFF


def next_offset() -> int:
    _OFFSET[0] += 1
    return _OFFSET[0]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# ---- gradient sink: while a trainer's backward pass runs, the layers' parameter gradients are written by the library STRAIGHT into the trainer's bucket -
# the flat buffer with the layout of its flat parameter buffer (train_gesture.GestureTrainer) - instead of into fresh tensors that autograd hands to 420
# AccumulateGrad nodes and the trainer packs with one more copy: a parameter at byte offset o of the parameter buffer has its gradient at offset o of the
# bucket (also a row slice of a parameter, e.g. the value rows of the decoder's cross-attention in-projection).  The functions return None for a gradient
# they wrote themselves.  A parameter that gets a second gradient in the same backward pass (no layer of the step does) takes the ordinary path: autograd
# adds it onto the bucket's view.
_SINK = None   # [first byte of the parameter buffer, its size in bytes, bucket address - parameter address, {addresses written in this pass}]


def sink_begin(flat_param: torch.Tensor, flat_grad: torch.Tensor):
    global _SINK
    assert flat_param.is_contiguous() and flat_grad.is_contiguous() and flat_param.shape == flat_grad.shape and flat_param.dtype == flat_grad.dtype == torch.float32
    _SINK = [flat_param.data_ptr(), flat_param.numel() * 4, flat_grad.data_ptr() - flat_param.data_ptr(), set()]


def sink_end():
    global _SINK
    _SINK = None


def _sink_ptr(t: torch.Tensor) -> int:
    """Bucket address for the gradient of parameter (slice) t, or 0: no sink active / t outside the parameter buffer / already written in this pass."""
    s = _SINK
    if s is None:
        return 0
    a = t.data_ptr()
    if a < s[0] or a >= s[0] + s[1] or a in s[3]:
        return 0
    s[3].add(a)
    return a + s[2]


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
    with _on(y.device):
        _lib.check(s["lib"].amuse_train_ln_fwd(_p(None if x is None else _c(x)), _p(y), _p(bias), _p(gamma), _p(beta), float(p), seed, off, rows,
                                               _p(out), _p(zhat), _p(rstd), _stream(y.device)))
    return out, zhat, rstd


def ln_bwd(dout, zhat, rstd, gamma, p: float, seed: int, off: int, need_dx: bool = True, need_dbias: bool = True, dout2=None):
    """LayerNorm backward of dout (+ dout2) -> (dx, dy, dgamma, dbeta, dbias)."""
    dout = _c(dout)
    rows = dout.numel() // 128
    s = _st(dout.device)
    dy = torch.empty_like(dout)
    dx = torch.empty_like(dout) if need_dx else None
    small = torch.empty(3, 128, device=dout.device, dtype=torch.float32)
    with _on(dout.device):
        _lib.check(s["lib"].amuse_train_ln_bwd(_p(dout), _p(None if dout2 is None else _c(dout2)), _p(zhat), _p(rstd), _p(gamma), float(p), seed, off, rows, _p(dx), _p(dy), small[0].data_ptr(),
                                               small[1].data_ptr(), small[2].data_ptr() if need_dbias else None, _p(s["ws"]), _stream(dout.device)))
    return dx, dy, small[0], small[1], (small[2] if need_dbias else None)


def bias_gelu_drop_fwd(h, b, p: float, seed: int, off: int):
    h = _c(h)
    F = h.shape[-1]
    out = torch.empty_like(h)
    with _on(h.device):
        _lib.check(_st(h.device)["lib"].amuse_train_bias_gelu_drop_fwd(_p(h), _p(b), float(p), seed, off, h.numel() // F, F, _p(out), _stream(h.device)))
    return out


def bias_gelu_drop_bwd(da, h, b, p: float, seed: int, off: int):
    da = _c(da)
    F = h.shape[-1]
    s = _st(h.device)
    dh = torch.empty_like(h)
    db = torch.empty(F, device=h.device, dtype=torch.float32)
    with _on(h.device):
        _lib.check(s["lib"].amuse_train_bias_gelu_drop_bwd(_p(da), _p(h), _p(b), float(p), seed, off, h.numel() // F, F, _p(dh), _p(db), _p(s["ws"]),
                                                          _stream(h.device)))
    return dh, db


def attn_fwd(qkv: torch.Tensor, B: int, S: int, p: float, seed: int, off: int, want_mask: bool = False):
    """qkv (B S, 384) -> (o (B S, 128), lse (B, 4, S)[, keep / (1 - p) (B, 4, S, S)])   csrc/k_train_attn.hip"""
    qkv = _c(qkv)
    o = torch.empty(B * S, 128, device=qkv.device, dtype=torch.float32)
    lse = torch.empty(B, 4, S, device=qkv.device, dtype=torch.float32)
    mask = torch.ones(B, 4, S, S, device=qkv.device, dtype=torch.float32) if want_mask else None
    with _on(qkv.device):
        _lib.check(_st(qkv.device)["lib"].amuse_train_attn_fwd(_p(qkv), B, S, float(p), seed, off, _p(o), _p(lse), _p(mask), _stream(qkv.device)))
    return (o, lse, mask) if want_mask else (o, lse)


def attn_bwd(qkv, o, lse, dout, B: int, S: int, p: float, seed: int, off: int) -> torch.Tensor:
    dqkv = torch.empty_like(qkv)
    with _on(qkv.device):
        _lib.check(_st(qkv.device)["lib"].amuse_train_attn_bwd(_p(_c(qkv)), _p(o), _p(lse), _p(_c(dout)), B, S, float(p), seed, off, _p(dqkv), _stream(qkv.device)))
    return dqkv


def colsum(x):
    x = _c(x)
    C = x.shape[-1]
    s = _st(x.device)
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    with _on(x.device):
        _lib.check(s["lib"].amuse_train_colsum(_p(x), x.numel() // C, C, _p(out), _p(s["ws"]), _stream(x.device)))
    return out


# ---------------------------------------------------------------------------------------------------- layers (no autograd inside)
_sdpa = torch.ops.aten._scaled_dot_product_efficient_attention
_sdpa_bwd = torch.ops.aten._scaled_dot_product_efficient_attention_backward


def own_attention(S: int, H: int, D: int) -> bool:
    """The library's attention kernels take 4 heads of 32 and up to 304 tokens (every attention of the training step); other shapes fall back to aten's op."""
    return H == 4 and D == 128 and 1 <= S <= 304


_ENC_PARAMS = ("Wo", "bo", "g1", "be1", "W1", "b1", "W2", "b2", "g3", "be3")
_DEC_PARAMS = ("Wo", "bo", "g1", "be1", "Wv", "bv", "Wc", "bc", "g2", "be2", "W1", "b1", "W2", "b2", "g3", "be3")


def _layer_forward(ctx, x, mem, Win, bin_, prm: dict, H: int, p: float, p_attn: float):
    with _on(x.device):
        return _layer_forward_on(ctx, x, mem, Win, bin_, prm, H, p, p_attn)


def _layer_backward(ctx, dout):
    with _on(dout.device):
        return _layer_backward_on(ctx, dout)


def _layer_forward_on(ctx, x, mem, Win, bin_, prm: dict, H: int, p: float, p_attn: float):
    """ONE call into the library (amuse_train_layer_fwd: in-projection, attention, the rest of the layer) - or, with aten's attention, the in-projection call, aten's op,
    and the layer call behind it."""
    B, S, D = x.shape
    rows, ff, dev = B * S, prm["W1"].shape[0], x.device
    st = _st(dev)
    lib, stream = st["lib"], _stream(dev)
    x2 = _c(x).view(rows, D)
    keep = any(ctx.needs_input_grad)                       # (no-grad passes - the iteration's second encode - keep nothing beyond the call)
    qkv = torch.empty(rows, 3 * D, device=dev, dtype=torch.float32)
    own = own_attention(S, H, D)
    if own:   # the library's fp32 attention (k_train_attn.hip) inside the layer call below: mask = hash of (seed, offset, clip, head, query, key)
        o2 = torch.empty(rows, D, device=dev, dtype=torch.float32)
        lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
        ao, ps, po = o2, o2, o2          # (placeholders in the saved list)
    else:     # the vendor's fused kernel (shapes the library's kernels do not take)
        _lib.check(lib.amuse_train_linear_fwd(x2.data_ptr(), Win.data_ptr(), bin_.data_ptr(), rows, D, 3 * D, qkv.data_ptr(), stream))
        q, k, v = (t.transpose(1, 2) for t in qkv.view(B, S, 3, H, D // H).unbind(2))   # (B, H, S, d) views
        ao, lse, ps, po = _sdpa(q, k, v, None, True, p_attn, False)
        o2 = ao.transpose(1, 2).reshape(rows, D)
    dec = mem is not None
    n128 = 9 if dec else 6
    b128 = torch.empty(n128, rows, D, device=dev, dtype=torch.float32)                   # x1 zh1 out zh3 tmp (vk xm zh2) | one spare row block for r1 r2 r3
    b512 = torch.empty(2, rows, ff, device=dev, dtype=torch.float32)
    L = _lib.TrainLayer()
    L.rows, L.B, L.S, L.H, L.ff, L.p, L.p_attn, L.seed = rows, B, S, H, ff, p, p_attn, _seed()
    for i in range(5):
        L.off[i] = next_offset()
    for n, t in prm.items():
        setattr(L, n, t.data_ptr())
    L.x, L.o2 = x2.data_ptr(), o2.data_ptr()
    if own:
        L.Win, L.bin, L.qkv, L.lse, L.off_self = Win.data_ptr(), bin_.data_ptr(), qkv.data_ptr(), lse.data_ptr(), next_offset()
    p128, blk = b128.data_ptr(), 4 * rows * D                                             # (addresses by arithmetic: a tensor view per pointer costs ~1 us of host time each)
    L.x1, L.zh1, L.out, L.zh3, L.tmp = (p128 + i * blk for i in range(5))
    rs = p128 + (n128 - 1) * blk                                                          # the three [rows] vectors of 1 / sigma
    L.r1, L.r2, L.r3 = rs, rs + 4 * rows, rs + 8 * rows
    L.h, L.a = b512.data_ptr(), b512.data_ptr() + 4 * rows * ff
    c = None
    if dec:
        mem2 = _c(mem).view(B, D)
        c = torch.empty(B, D, device=dev, dtype=torch.float32)
        L.mem, L.c = mem2.data_ptr(), c.data_ptr()
        L.vk, L.xm, L.zh2 = (p128 + i * blk for i in (5, 6, 7))
    _lib.check(lib.amuse_train_layer_fwd(C.byref(L), stream))
    if keep:
        ctx.save_for_backward(x2, qkv, ao, lse, ps, po, o2, b128, b512, Win, *((mem2, c) if dec else ()), *prm.values())
        ctx.L, ctx.cfg = L, (B, S, D, H, p_attn, dec, tuple(prm), own)
        ctx.bin_ = bin_   # (address only: the gradient sink; not needed by the arithmetic)
        ctx.sink_ok = all(ctx.needs_input_grad[2 if dec else 1:-3])   # every parameter of the layer takes a gradient
    return b128[2].view(B, S, D)


def _layer_backward_on(ctx, dout):
    t = ctx.saved_tensors
    B, S, D, H, p_attn, dec, names, own = ctx.cfg
    x2, qkv, ao, lse, ps, po, o2, b128, b512, Win = t[:10]
    prm = dict(zip(names, t[12 if dec else 10:]))
    rows, ff, dev = B * S, b512.shape[2], x2.device
    st = _st(dev)
    lib, stream = st["lib"], _stream(dev)
    L = ctx.L
    dout = _c(dout).view(rows, D)
    g128 = torch.empty(4, rows, D, device=dev, dtype=torch.float32)                      # dx do2 s128a s128b
    g512 = torch.empty(2, rows, ff, device=dev, dtype=torch.float32)
    sink = {n: _sink_ptr(prm[n]) for n in names} if ctx.sink_ok else dict.fromkeys(names, 0)   # (gradients the library writes into the trainer's bucket itself)
    sizes = {n: (0 if sink[n] else prm[n].numel()) for n in names}
    flat = torch.empty(sum(sizes.values()) + 2 * B * D, device=dev, dtype=torch.float32)   # the layer's parameter gradients (+ d(mem), the d(c) scratch)
    grads, o = {}, 0
    for n in names:
        if sink[n]:
            grads[n] = None
            setattr(L, "d" + n, sink[n])
            continue
        grads[n] = flat[o:o + sizes[n]].view_as(prm[n])
        setattr(L, "d" + n, grads[n].data_ptr())
        o += sizes[n]
    dmem = flat[o:o + B * D].view(B, 1, D)
    L.dmem, L.sdc = dmem.data_ptr(), flat[o + B * D:].data_ptr()
    L.dout = dout.data_ptr()
    L.dx, L.do2, L.s128a, L.s128b = (g128.data_ptr() + i * 4 * rows * D for i in range(4))
    L.s512a, L.s512b, L.ws = g512.data_ptr(), g512.data_ptr() + 4 * rows * ff, st["ws"].data_ptr()
    bin_ = ctx.bin_
    pW, pb = (_sink_ptr(Win), _sink_ptr(bin_)) if ctx.sink_ok and bin_ is not None else (0, 0)
    dWin = None if pW else torch.empty_like(Win)
    dbin = None if pb else torch.empty(3 * D, device=dev, dtype=torch.float32)
    pW, pb = pW or dWin.data_ptr(), pb or dbin.data_ptr()
    if own:   # one call: the layer, the attention's backward pass and the in-projection's
        dqkv = torch.empty(rows, 3 * D, device=dev, dtype=torch.float32)
        L.dqkv, L.dWin, L.dbin = dqkv.data_ptr(), pW, pb
        _lib.check(lib.amuse_train_layer_bwd(C.byref(L), stream))
    else:
        _lib.check(lib.amuse_train_layer_bwd(C.byref(L), stream))
        do = g128[1].view(B, S, H, D // H).transpose(1, 2)
        q, k, v = (u.transpose(1, 2) for u in qkv.view(B, S, 3, H, D // H).unbind(2))
        dq, dk, dv, _ = _sdpa_bwd(do, q, k, v, None, ao, lse, ps, po, p_attn, (True, True, True, False), False)
        dqkv = torch.stack([dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)], dim=2).view(rows, 3 * D)
        _lib.check(lib.amuse_train_linear_bwd(dqkv.data_ptr(), x2.data_ptr(), Win.data_ptr(), rows, D, 3 * D, pW, pb, g128[0].data_ptr(), 1,
                                              st["ws"].data_ptr(), stream))
    return g128[0].view(B, S, D), (dmem if dec else None), dWin, dbin, grads


class EncoderLayerFn(torch.autograd.Function):
    """norm2(x1 + dropout2(linear2(dropout(gelu(linear1(x1)))))), x1 = norm1(x + dropout1(self_attn(x)))   (cross_attention.py:259-272)."""

    @staticmethod
    def forward(ctx, x, Win, bin_, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, H, p, p_attn):
        return _layer_forward(ctx, x, None, Win, bin_, dict(zip(_ENC_PARAMS, (Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2))), H, p, p_attn)

    @staticmethod
    def backward(ctx, dout):
        dx, _, dWin, dbin, g = _layer_backward(ctx, dout)
        return (dx, dWin, dbin, *(g[n] for n in _ENC_PARAMS), None, None, None)


class DecoderLayerFn(torch.autograd.Function):
    """TransformerDecoderLayer.forward_post with a memory of ONE token (cross_attention.py:323-345; MotionPrior.decode, vae.py:252-259): self-attention,
    the cross-attention's value path (a softmax over one key is 1: nn_modules.mha_one_key), FFN; three norms.  Wv / bv are the value rows of the
    cross-attention's packed in-projection (sliced by the caller, whose autograd zero-fills the q / k rows' gradients as eager does)."""

    @staticmethod
    def forward(ctx, x, mem, Win, bin_, Wo, bo, g1, be1, Wv, bv, Wc, bc, g2, be2, W1, b1, W2, b2, g3, be3, H, p, p_attn):
        prm = dict(zip(_DEC_PARAMS, (Wo, bo, g1, be1, _c(Wv), _c(bv), Wc, bc, g2, be2, W1, b1, W2, b2, g3, be3)))
        return _layer_forward(ctx, x, mem, Win, bin_, prm, H, p, p_attn)

    @staticmethod
    def backward(ctx, dout):
        dx, dmem, dWin, dbin, g = _layer_backward(ctx, dout)
        return (dx, dmem, dWin, dbin, *(g[n] for n in _DEC_PARAMS), None, None, None)


class LinearFn(torch.autograd.Function):
    """nn.Linear on the library's two entry points (amuse_train_linear_fwd / _bwd: the library's own fp32-MFMA GEMMs, the bias in the GEMM's epilogue, the bias
    gradient by the deterministic column sum) - the skip linears, embeddings, condition projections and output layer around the transformer layers."""

    @staticmethod
    def forward(ctx, x, W, b):
        N, K = W.shape
        x2 = _c(x).view(-1, K)
        rows = x2.shape[0]
        out = torch.empty(rows, N, device=x.device, dtype=torch.float32)
        with _on(x.device):
            _lib.check(_st(x.device)["lib"].amuse_train_linear_fwd(x2.data_ptr(), W.data_ptr(), _p(b), rows, K, N, out.data_ptr(), _stream(x.device)))
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.b = b   # (address only: the gradient sink)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dout):
        x2, W = ctx.saved_tensors
        N, K = W.shape
        st = _st(x2.device)
        dy = _c(dout).view(-1, N)
        rows = dy.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty(rows, K, device=dy.device, dtype=torch.float32) if need_x else None
        pW = _sink_ptr(W) if need_w else 0                      # (the trainer's gradient bucket, see sink_begin)
        pb = _sink_ptr(ctx.b) if need_b else 0
        dW = torch.empty_like(W) if need_w and not pW else None
        db = torch.empty(N, device=dy.device, dtype=torch.float32) if need_b and not pb else None
        with _on(dy.device):
            _lib.check(st["lib"].amuse_train_linear_bwd(dy.data_ptr(), x2.data_ptr(), W.data_ptr(), rows, K, N, pW or _p(dW), pb or _p(db), _p(dx), 0, st["ws"].data_ptr(),
                                                        _stream(dy.device)))
        return (None if dx is None else dx.view(*dout.shape[:-1], K)), dW, db


def linear(m, x: torch.Tensor) -> torch.Tensor:
    """nn.Linear `m` on x: the library path for CUDA fp32 inputs (any width up to 1024 outputs: the tall projections on k_train_gemm_tall, every other shape -
    the 333-wide embedding / output layers, the Denoiser's 32-row condition projections - on the generic fp32-MFMA kernel), else F.linear."""
    N = m.weight.shape[0]
    if enabled() and x.is_cuda and x.dtype == torch.float32 and N <= 1024 and m.weight.is_contiguous():
        return LinearFn.apply(x, m.weight, m.bias)
    return torch.nn.functional.linear(x, m.weight, m.bias)


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


# ---------------------------------------------------------------------------------------------------- optimizer
class FlatAdamW(torch.optim.AdamW):
    """torch.optim.AdamW (trainer.py:181-184) over the trainer's flat buffers: `step()` is one amuse_train_adamw launch per contiguous run of its
    parameters (all of them but the ones the iteration never reaches: 1-3 runs) instead of torch's multi-tensor path (36 launches, 0.75 ms of device and
    ~0.9 ms of host time per iteration for 6.8 M parameters).  Both moments live in flat buffers of the parameters' layout; `state` exposes them per
    parameter as views, so `state_dict()` has torch's own structure (the checkpoint writer stores it, trainer.py:468-496).
    layout: [(parameter, offset, numel)] of EVERY parameter in the flat buffers, in order."""

    def __init__(self, params, flat_param: torch.Tensor, flat_grad: torch.Tensor, layout, **kw):
        self._ranges = None
        super().__init__(params, **kw)
        assert flat_param.is_cuda and flat_param.dtype == torch.float32 and flat_param.is_contiguous() and flat_grad.shape == flat_param.shape
        self._p, self._g = flat_param, flat_grad
        self._m, self._v = torch.zeros_like(flat_param), torch.zeros_like(flat_param)
        self._t = 0
        self._t_dev = torch.zeros(1, dtype=torch.int64, device=flat_param.device)    # step_dev(): the step count and the two bias-correction scalars on the device
        self._scal = torch.zeros(2, dtype=torch.float32, device=flat_param.device)
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        assert len(self.param_groups) == 1 and not self.param_groups[0].get("amsgrad") and not self.param_groups[0].get("maximize")
        self._ranges = []
        for p, off, n in layout:
            if id(p) not in mine:
                continue
            assert p.data_ptr() == flat_param.data_ptr() + 4 * off and p.numel() == n
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self._m[off:off + n].view_as(p), "exp_avg_sq": self._v[off:off + n].view_as(p)}
            if self._ranges and self._ranges[-1][0] + self._ranges[-1][1] == off:
                self._ranges[-1][1] += n
            else:
                self._ranges.append([off, n])

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        self._t += 1
        g = self.param_groups[0]
        dev = self._p.device
        lib, stream = _st(dev)["lib"], _stream(dev)
        b = self._p.data_ptr()
        with _on(dev):
            for off, n in self._ranges:
                _lib.check(lib.amuse_train_adamw(b + 4 * off, self._g.data_ptr() + 4 * off, self._m.data_ptr() + 4 * off, self._v.data_ptr() + 4 * off, n, float(g["lr"]),
                                                 float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._t, stream))

    @torch.no_grad()
    def step_dev(self):
        """The same update with the step count kept ON THE DEVICE (amuse_train_adamw_dev): capturable in a HIP graph - every replay advances the count by one.
        The host's count follows through `sync_step()` (state_dict calls it)."""
        g = self.param_groups[0]
        dev = self._p.device
        lib, stream = _st(dev)["lib"], _stream(dev)
        b = self._p.data_ptr()
        with _on(dev):
            for i, (off, n) in enumerate(self._ranges):
                _lib.check(lib.amuse_train_adamw_dev(b + 4 * off, self._g.data_ptr() + 4 * off, self._m.data_ptr() + 4 * off, self._v.data_ptr() + 4 * off, n, float(g["lr"]),
                                                     float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._t_dev.data_ptr(),
                                                     self._scal.data_ptr(), int(i == 0), stream))

    def push_step(self):
        """host count -> device (before a graph with step_dev() is captured / replayed for the first time)"""
        self._t_dev.fill_(self._t)

    def sync_step(self):
        """device count -> host (after replays of a captured step_dev())"""
        self._t = max(self._t, int(self._t_dev.item()))

    def state_dict(self):
        self.sync_step()
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._t))
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Resume (the `optimizer_state_dict` the checkpoint writer stores, trainer.py:468-496): torch's loader replaces the per-parameter state tensors
        with copies, which would detach them from the flat buffers the kernel reads - copy the loaded moments INTO the flat buffers, re-create the views and
        restore the step count the bias correction uses."""
        mine = list(self.state.keys())                               # parameters in the order torch numbers them
        views = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        steps = set()
        for p in mine:
            st = self.state.get(p)
            m, v = views[p]
            if st is None or "exp_avg" not in st:                    # a parameter the saved run never stepped
                m.zero_(), v.zero_()
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": m, "exp_avg_sq": v}
                continue
            m.copy_(st["exp_avg"]), v.copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
            st["exp_avg"], st["exp_avg_sq"] = m, v
        steps.discard(0)
        if len(steps) > 1:
            raise ValueError(f"FlatAdamW takes ONE step count for all parameters, the state dict holds {sorted(steps)}")
        self._t = steps.pop() if steps else 0
        self._t_dev.fill_(self._t)

    def zero_grad(self, set_to_none: bool = False):
        """The gradients are views of the flat bucket the kernel reads: zero the bucket, never detach the views."""
        self._g.zero_()

    def add_param_group(self, param_group):
        if getattr(self, "_ranges", None) is not None:
            raise NotImplementedError("FlatAdamW is laid out over ONE parameter group at construction")
        super().add_param_group(param_group)

