"""Autograd twins of the two networks on the hot path, for the train_gesture step (BASELINE config 4): `Denoiser`
(reference models/latent_diffusion/denoiser.py:16-204, trans_enc + learned PE) and `MotionPrior`
(models/latent_diffusion/vae.py:24-278, encoder_decoder).  Plain torch modules; on the GPU each transformer layer runs as one autograd.Function
whose non-GEMM arithmetic is hand-written HIP (train_ops.py, csrc/k_train.hip; AMUSE_TRAIN_FUSED=0 = the eager layers below).  What matters here is that

  * `state_dict()` has exactly the reference's keys and shapes (tests/golden/state_dict_spec.json), so checkpoints written
    by either side load in the other and in the HIP engine (amuse_amd/checkpoint.py, amuse_update_weights), and
  * the forward pass IS the reference's, dropout included (cross_attention.py:259-272,323-345: attention dropout inside
    nn.MultiheadAttention, dropout1/2/3 on the residual branches, dropout between the FFN linears) - pinned in eval mode
    against the golden vectors of the reference modules (tests/test_train_cpu.py).

Layout.  The reference runs sequence-first (S, B, D); every operator but attention is row-wise, so the stacks here run
batch-first (B, S, D) - the layout in which the packed q|k|v projection splits into (B, H, S, d) views and the attention
output is already the out_proj input: no transposed copies (the sequence-first nn.MultiheadAttention path made ~10 strided
copies per layer and iteration, 2.6 ms of device time per training step).  `nn.MultiheadAttention` remains the parameter
holder (state-dict keys in_proj_weight / in_proj_bias / out_proj.*); its arithmetic - q scaled by 1/sqrt(d), additive
-inf key-padding mask, softmax, dropout on the probabilities, out_proj - is `mha_self` below.  The decoder's cross-attention
looks at ONE memory token: the softmax over a single key is exactly 1 whatever q and k are (and their gradients exactly
zero), so `mha_one_key` computes only the value path, as the HIP decoder does.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_ops

D, H, FF, L, COND, NFEATS, PE_LEN = 128, 4, 512, 9, 256, 333, 500


class LearnedPE(nn.Module):
    """position_encoding.py:138-159 (PositionEmbeddingLearned1D, sequence-first): x + pe[:len(x)]."""

    def __init__(self, d_model=D, max_len=PE_LEN):
        super().__init__()
        self.pe = nn.Parameter(torch.zeros(max_len, 1, d_model))
        nn.init.uniform_(self.pe)

    def forward(self, x):                                  # x (B, S, D)
        return x + self.pe[: x.shape[1]].transpose(0, 1)


def mha_self(attn: nn.MultiheadAttention, x: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MultiheadAttention(x, x, x, key_padding_mask, need_weights=False) on batch-first x (B, S, E); key_padding_mask (B, S)
    True = ignore that key, None = no padding (F.multi_head_attention_forward: packed in-projection, heads = contiguous
    E / H slices, scores q k^T / sqrt(d), dropout on the softmax, out_proj)."""
    B, S, E = x.shape
    h = attn.num_heads
    q, k, v = F.linear(x, attn.in_proj_weight, attn.in_proj_bias).view(B, S, 3, h, E // h).unbind(2)
    mask = None if key_padding_mask is None else ~key_padding_mask[:, None, None, :]
    o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask,
                                       dropout_p=attn.dropout if attn.training else 0.0)
    return F.linear(o.transpose(1, 2).reshape(B, S, E), attn.out_proj.weight, attn.out_proj.bias)


def mha_one_key(attn: nn.MultiheadAttention, tgt: torch.Tensor, memory: torch.Tensor) -> torch.Tensor:
    """nn.MultiheadAttention(tgt, memory, memory) for a memory of one token, batch-first: tgt (B, S, E), memory (B, 1, E).
    Every query's probability vector is [1]; attention dropout turns it into 0 or 1 / (1 - p) per (clip, head, query)."""
    B, S, E = tgt.shape
    h = attn.num_heads
    v = F.linear(memory, attn.in_proj_weight[2 * E:], attn.in_proj_bias[2 * E:])              # (B, 1, E)
    if attn.training and attn.dropout > 0:
        keep = F.dropout(torch.ones(B, S, h, 1, device=tgt.device, dtype=tgt.dtype), attn.dropout, True)
        return F.linear((keep * v.view(B, 1, h, E // h)).reshape(B, S, E), attn.out_proj.weight, attn.out_proj.bias)
    return F.linear(v, attn.out_proj.weight, attn.out_proj.bias).expand(B, S, E)


class EncoderLayer(nn.Module):
    """cross_attention.py:236-272, forward_post (normalize_before False), activation gelu (exact erf)."""

    def __init__(self, d=D, h=H, ff=FF, p=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, h, dropout=p)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.dropout, self.dropout1, self.dropout2 = nn.Dropout(p), nn.Dropout(p), nn.Dropout(p)
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, src, key_padding_mask=None):
        if train_ops.usable(src, key_padding_mask):       # one autograd.Function on the HIP glue kernels (train_ops.py); same arithmetic
            return train_ops.encoder_layer(self, src)
        src2 = mha_self(self.self_attn, src, key_padding_mask)
        src = self.norm1(src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(F.gelu(self.linear1(src))))
        return self.norm2(src + self.dropout2(src2))


class DecoderLayer(nn.Module):
    """cross_attention.py:297-345, forward_post: self-attention, cross-attention onto `memory`, FFN; three norms."""

    def __init__(self, d=D, h=H, ff=FF, p=0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, h, dropout=p)
        self.multihead_attn = nn.MultiheadAttention(d, h, dropout=p)
        self.linear1, self.linear2 = nn.Linear(d, ff), nn.Linear(ff, d)
        self.dropout, self.dropout1, self.dropout2, self.dropout3 = (nn.Dropout(p) for _ in range(4))
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, tgt, memory, tgt_key_padding_mask=None):
        if memory.shape[1] == 1 and train_ops.usable(tgt, tgt_key_padding_mask):
            return train_ops.decoder_layer(self, tgt, memory)
        t2 = mha_self(self.self_attn, tgt, tgt_key_padding_mask)
        tgt = self.norm1(tgt + self.dropout1(t2))
        if memory.shape[1] == 1:
            t2 = mha_one_key(self.multihead_attn, tgt, memory)
        else:
            t2 = self.multihead_attn(tgt.transpose(0, 1), memory.transpose(0, 1), memory.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
        tgt = self.norm2(tgt + self.dropout2(t2))
        t2 = self.linear2(self.dropout(F.gelu(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout3(t2))


class SkipStack(nn.Module):
    """SkipTransformerEncoder / SkipTransformerDecoder (cross_attention.py:18-125): 4 input blocks whose outputs are
    stacked, a middle block, 4 output blocks each preceded by Linear(cat(x, stack.pop())), final LayerNorm."""

    def __init__(self, make_layer, num_layers=L, d=D):
        super().__init__()
        assert num_layers % 2 == 1
        n = (num_layers - 1) // 2
        self.input_blocks = nn.ModuleList(make_layer() for _ in range(n))
        self.middle_block = make_layer()
        self.output_blocks = nn.ModuleList(make_layer() for _ in range(n))
        self.linear_blocks = nn.ModuleList(nn.Linear(2 * d, d) for _ in range(n))
        self.norm = nn.LayerNorm(d)
        for p in self.parameters():                       # _reset_parameters (cross_attention.py:36-39)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, x, *args, **kw):
        xs = []
        for m in self.input_blocks:
            x = m(x, *args, **kw)
            xs.append(x)
        x = self.middle_block(x, *args, **kw)
        for m, lin in zip(self.output_blocks, self.linear_blocks):
            x = train_ops.linear(lin, torch.cat([x, xs.pop()], dim=-1))
            x = m(x, *args, **kw)
        return self.norm(x)


class TimestepEmbedding(nn.Module):
    """embeddings.py:269-322: Linear -> SiLU -> Linear."""

    def __init__(self, channel=COND, time_embed_dim=D):
        super().__init__()
        self.linear_1 = nn.Linear(channel, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return train_ops.linear(self.linear_2, F.silu(train_ops.linear(self.linear_1, sample)))


def timestep_sinusoid(timesteps: torch.Tensor, dim=COND) -> torch.Tensor:
    """Timesteps(256, flip_sin_to_cos=True, freq_shift=0) (embeddings.py:245-267 -> get_timestep_embedding): [cos | sin]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / (half - 0)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class Denoiser(nn.Module):
    """denoiser.py:16-204, arch trans_enc with skip connections, pe_type mld / learned.  130 state-dict entries."""

    def __init__(self, dropout=0.1):
        super().__init__()
        self.time_embedding = TimestepEmbedding()
        self.emb_proj_con = nn.Sequential(nn.ReLU(), nn.Linear(COND, D))
        self.emb_proj_emo = nn.Sequential(nn.ReLU(), nn.Linear(COND, D))
        self.emb_proj_sty = nn.Sequential(nn.ReLU(), nn.Linear(COND, D))
        self.query_pos, self.mem_pos = LearnedPE(), LearnedPE()
        self.encoder = SkipStack(lambda: EncoderLayer(p=dropout))

    def forward(self, sample, timestep, con_hidden, emo_hidden=None, sty_hidden=None, lengths=None, **kw):
        """sample (B, 1, 128); timestep (B,) or scalar; *_hidden (B, 256) -> (eps_hat (B, 1, 128),)   (denoiser.py:135-204)."""
        bsz = sample.shape[0]
        timesteps = torch.as_tensor(timestep, device=sample.device).expand(bsz)
        time_emb = self.time_embedding(timestep_sinusoid(timesteps).to(sample.dtype)).unsqueeze(1)
        proj = lambda seq, z: train_ops.linear(seq[1], F.relu(z))           # nn.Sequential(ReLU, Linear) (denoiser.py:85-90): the Linear on the library's GEMM
        toks = [time_emb, proj(self.emb_proj_con, con_hidden).unsqueeze(1)]
        if emo_hidden is not None:
            toks.append(proj(self.emb_proj_emo, emo_hidden).unsqueeze(1))
        if sty_hidden is not None:
            toks.append(proj(self.emb_proj_sty, sty_hidden).unsqueeze(1))
        xseq = self.query_pos(torch.cat([sample] + toks, dim=1))          # (B, S, 128), latent token first
        tokens = self.encoder(xseq)
        return (tokens[:, : sample.shape[1]],)


_MASKS = {}


def lengths_to_mask(lengths: Sequence[int], device, max_len: Optional[int] = None) -> torch.Tensor:
    """temos_utils.py: True where the frame is valid.  Cached per (lengths, device): the list -> tensor upload is a host-to-device copy,
    which a HIP-graph capture of the training step (train_gesture.py) must not contain; the callers never modify the mask."""
    key = (tuple(int(v) for v in lengths), str(device), max_len)
    m = _MASKS.get(key)
    if m is None:
        lt = torch.as_tensor(list(key[0]), device=device)
        n = max_len or int(max(key[0]))
        m = torch.arange(n, device=device)[None, :] < lt[:, None]
        if len(_MASKS) > 64:
            _MASKS.clear()
        _MASKS[key] = m
    return m


def _all_valid(lengths: Sequence[int], n: int) -> bool:
    return all(int(v) >= n for v in lengths)


class MotionPrior(nn.Module):
    """vae.py:24-278 (emotional prior with finger joints: 333 features, 1 latent token, mld learned PE)."""

    def __init__(self, dropout=0.1):
        super().__init__()
        self.global_motion_token = nn.Parameter(torch.randn(2, D))
        self.query_pos_encoder, self.query_pos_decoder = LearnedPE(), LearnedPE()
        self.encoder = SkipStack(lambda: EncoderLayer(p=dropout))
        self.decoder = SkipStack(lambda: DecoderLayer(p=dropout))
        self.skel_embedding = nn.Linear(NFEATS, D)
        self.final_layer = nn.Linear(D, NFEATS)
        self.validate_args = None   # passed to torch.distributions.Normal in encode

    def encode(self, features, lengths: Optional[List[int]] = None):
        """features (B, T, 333) -> (latent (1, B, 128) = dist.rsample(), dist = Normal(mu, exp(logvar) ** 0.5))   (vae.py:154-214)."""
        if lengths is None:
            lengths = [features.shape[1]] * features.shape[0]
        bs, nframes, _ = features.shape
        x = train_ops.linear(self.skel_embedding, features)                # (B, T, 128)
        dist = self.global_motion_token[None].expand(bs, -1, -1)           # (B, 2, 128)
        xseq = self.query_pos_encoder(torch.cat([dist, x], 1))
        kpm = None
        if not _all_valid(lengths, nframes):
            mask = lengths_to_mask(lengths, features.device, nframes)
            kpm = ~torch.cat([torch.ones(bs, 2, dtype=torch.bool, device=x.device), mask], 1)
        out = self.encoder(xseq, key_padding_mask=kpm)
        mu, logvar = out[:, 0][None], out[:, 1][None]                      # (1, B, 128) each
        std = logvar.exp().pow(0.5)
        # validate_args: torch's default (None = on) checks `std > 0` ON THE HOST - two blocking device -> host reads per construction.  The trainer
        # switches it off (train_gesture.Trainer: the step has no other host synchronisation, and a bad std shows up in the losses it logs)
        d = torch.distributions.Normal(mu, std, validate_args=self.validate_args)
        return d.rsample(), d

    def decode(self, z, lengths: List[int]):
        """z (1, B, 128) -> feats (B, T, 333), frames beyond a clip's length zeroed   (vae.py:216-278)."""
        bs, nframes = len(lengths), int(max(lengths))
        queries = self.query_pos_decoder(torch.zeros(bs, nframes, D, device=z.device, dtype=z.dtype))
        full = _all_valid(lengths, nframes)
        mask = None if full else lengths_to_mask(lengths, z.device)
        out = self.decoder(queries, z.transpose(0, 1), tgt_key_padding_mask=None if full else ~mask)
        out = train_ops.linear(self.final_layer, out)
        return out if full else out.masked_fill(~mask[:, :, None], 0.0)


def load_numpy_state(module: nn.Module, sd) -> nn.Module:
    """Copy a {name: ndarray} state dict (amuse_amd/weights.py, checkpoint.py) into the module; keys / shapes must match."""
    own = module.state_dict()
    assert set(own) == set(sd), (sorted(set(own) ^ set(sd))[:6])
    with torch.no_grad():
        for k, v in own.items():
            t = torch.as_tensor(sd[k])
            assert tuple(t.shape) == tuple(v.shape), (k, tuple(t.shape), tuple(v.shape))
            v.copy_(t)
    return module


def numpy_state(module: nn.Module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}
