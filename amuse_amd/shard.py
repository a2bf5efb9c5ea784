"""Clip-batch sharding for multi-GPU sampling: one process per GPU, contiguous clip ranges, NO collective
on the data path (sampling is embarrassingly parallel over clips; the reference itself refuses multi-GPU
inference, models/audio/infer_pretrained_ast_evp.py:45).  Noise is keyed by the GLOBAL clip index, so the
result is independent of the number of shards.  The only communication is the optional gather of outputs."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced: the first (total % world) ranks take one extra clip."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} out of range for world {world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sample_sharded(sample_fn: Callable[..., Dict[str, torch.Tensor]], z_con, z_emo, z_sty, rank: int, world: int,
                   gather: bool = False, group=None) -> Optional[Dict[str, torch.Tensor]]:
    """Run `sample_fn(bsz, con, emo, sty, clip_index0=...)` on this rank's shard of the global batch.
    gather=False: returns the local shard's outputs (stay on this rank's device).
    gather=True : all ranks exchange shards (torch.distributed all_gather_object) and return the full batch
                  in global clip order - for tests and small jobs; large jobs should keep outputs sharded."""
    total = z_con.shape[0]
    lo, hi = shard_range(total, rank, world)
    sl = lambda t: None if t is None else t[lo:hi]
    out = sample_fn(hi - lo, sl(z_con), sl(z_emo), sl(z_sty), clip_index0=lo) if hi > lo else {}
    if not gather or world == 1:
        return out
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, {k: v.cpu() for k, v in out.items()}, group=group)
    keys = [k for p in parts for k in p.keys()]
    return {k: torch.cat([p[k] for p in parts if k in p]) for k in dict.fromkeys(keys)}
