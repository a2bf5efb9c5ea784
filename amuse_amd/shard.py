"""Clip-batch sharding for multi-GPU sampling: one process per GPU, contiguous clip ranges, NO collective
on the data path (sampling is embarrassingly parallel over clips; the reference itself refuses multi-GPU
inference, models/audio/infer_pretrained_ast_evp.py:45).  Noise is keyed by the GLOBAL clip index, so the
result is independent of the number of shards.  The only communication is the optional gather of outputs."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch


def job_clips_per_group(total: int, tokens: int = 5) -> int:
    """Clips per workgroup tile for a job of `total` clips (the library's auto rule applied to the WHOLE job, not to a
    shard): ceil(total / 128), at most 16 // tokens.  Every rank sets it (HipEngine.set_clips_per_group) and shards start
    at multiples of it, so that a clip sits in the same slot of its tile however the job is sharded - which is what makes
    sharded results BITWISE equal to the single-GPU ones (include/amuse_hip.h amuse_set_clips_per_group)."""
    return max(1, min(16 // tokens, -(-total // 128)))


FUSED_DECODE_MIN_CLIPS = 64   # amuse_api.hip kFusedMinClips: the library's per-launch rule for the bf16 / fp16 / fp32x decode kernels


def fusedx_rule(total: int) -> bool:
    """amuse_api.hip fusedx_rule: does a call of `total` clips decode on the fp32x mode's per-clip kernel (k_vae_fusedx.hip)?  A clip takes ~1.5 ms on its CU whatever
    the batch, so the kernel wins when the clips fill rounds of the chip's 256 CUs: from 160 clips in the first round, in round r >= 2 with at least 164 - 50 (r - 2)."""
    if total < 160:
        return False
    r = -(-total // 256)
    return r == 1 or total - 256 * (r - 1) >= 164 - 50 * (r - 2)


def job_decode_path(total: int) -> str:
    """The decode kernels (amuse_hip.h amuse_set_decode_path) a job of `total` clips would get on one GPU.  Like the clips
    per tile it must be chosen from the WHOLE job, not per shard: each 16-bit mode has a staged and a fused decoder and the
    fp32x mode three (k_vae_rows<f16x2> below 64 clips, k_vae_rows8x from 64, the per-clip k_vae_fusedx where the clips fill
    rounds of the chip) that sum in different orders, so a 256-clip job cut into 32-clip shards would otherwise decode on
    other kernels than the same job on one GPU (only the fp32 mode has a single decode path).  "clip" means "fused" in the
    modes that have no third kernel."""
    if fusedx_rule(total):
        return "clip"
    return "fused" if total >= FUSED_DECODE_MIN_CLIPS else "staged"


def shard_range(total: int, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous, balanced in units of `align` clips: the first ranks take one extra unit; the last unit may be short."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} out of range for world {world}")
    units = -(-total // align)
    base, extra = divmod(units, world)
    start = rank * base + min(rank, extra)
    stop = start + base + (1 if rank < extra else 0)
    return min(total, start * align), min(total, stop * align)


def job_range(bszs, rank: int, world: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous range [ja, jb) of JOBS for this rank, for jobs of bszs[j] clips each laid out back to back (job j's clips have the
    global indices offs[j] .. offs[j+1]): the clip axis is cut by shard_range(total, rank, world, align) and every cut is moved
    forward to the next job boundary whose clip offset is a multiple of `align` (the job's clips per tile: a clip then sits in the same
    slot of its tile as in the single-process launch, which is what keeps sharded results bitwise).  The ranges of ranks 0..world-1
    are disjoint, ordered and cover every job; a rank may get none."""
    offs = [0]
    for b in bszs:
        offs.append(offs[-1] + int(b))
    total, n = offs[-1], len(bszs)
    ok = [j for j in range(n + 1) if offs[j] % align == 0 or j == n]

    def snap(clip):
        for j in ok:
            if offs[j] >= clip:
                return j
        return n
    lo, hi = shard_range(total, rank, world, align)
    ja = 0 if rank == 0 else snap(lo)
    jb = n if rank == world - 1 else snap(hi)
    return ja, max(ja, jb)


def sample_sharded(sample_fn: Callable[..., Dict[str, torch.Tensor]], z_con, z_emo, z_sty, rank: int, world: int,
                   gather: bool = False, group=None, set_clips_per_group: Optional[Callable[[int], None]] = None,
                   set_decode_path: Optional[Callable[[str], None]] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Run `sample_fn(bsz, con, emo, sty, clip_index0=...)` on this rank's shard of the global batch.
    set_clips_per_group (e.g. HipEngine.set_clips_per_group): called with job_clips_per_group(total), and the shards are
                  aligned to it - sharded results are then bitwise those of one GPU running the whole job.
    set_decode_path (e.g. HipEngine.set_decode_path): called with job_decode_path(total) for the same reason; both are
                  put back to the library's automatic rule afterwards.
    gather=False: returns the local shard's outputs (stay on this rank's device).
    gather=True : all ranks exchange shards (torch.distributed all_gather_object) and return the full batch
                  in global clip order - for tests and small jobs; large jobs should keep outputs sharded."""
    total = z_con.shape[0]
    tokens = 3 + (z_emo is not None) + (z_sty is not None)
    g = job_clips_per_group(total, tokens) if set_clips_per_group is not None else 1
    if set_clips_per_group is not None:
        set_clips_per_group(g)
    if set_decode_path is not None:
        set_decode_path(job_decode_path(total))
    lo, hi = shard_range(total, rank, world, align=g)
    sl = lambda t: None if t is None else t[lo:hi]
    try:
        out = sample_fn(hi - lo, sl(z_con), sl(z_emo), sl(z_sty), clip_index0=lo) if hi > lo else {}
    finally:
        if set_clips_per_group is not None:
            set_clips_per_group(0)   # back to the library's per-launch rule: the job's tiling must not leak into later calls
        if set_decode_path is not None:
            set_decode_path("auto")
    if not gather or world == 1:
        return out
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, {k: v.cpu() for k, v in out.items()}, group=group)
    keys = [k for p in parts for k in p.keys()]
    return {k: torch.cat([p[k] for p in parts if k in p]) for k in dict.fromkeys(keys)}
