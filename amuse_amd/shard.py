"""Clip-batch sharding for multi-GPU sampling: one process per GPU, contiguous clip ranges, NO collective
on the data path (sampling is embarrassingly parallel over clips; the reference itself refuses multi-GPU
inference, models/audio/infer_pretrained_ast_evp.py:45).  Noise is keyed by the GLOBAL clip index, so the
result is independent of the number of shards.  The only communication is the optional gather of outputs."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from . import _lib


def job_plan(total: int, tokens: int = 5, precision: int = _lib.PREC_F32X, arch: int = _lib.ARCH_ENC) -> Dict[str, object]:
    """The launch plan of a JOB of `total` clips, from the library itself (include/amuse_hip.h amuse_plan; no GPU needed) - nothing here restates
    its rules.  Every rank applies the WHOLE job's plan to its shard:
      "clips_per_group"  clips per workgroup tile (HipEngine.set_clips_per_group); shards start at multiples of it, so that a clip sits in the same
                         slot of its tile however the job is sharded - which is what makes sharded results BITWISE equal to the single-GPU ones;
      "decode_path"      the decode / encode / pose-step kernel family (HipEngine.set_decode_path): the modes' kernel families sum in different orders,
                         so a 256-clip job cut into 32-clip shards would otherwise decode on other kernels than the same job on one GPU.
    The default precision is fp32x, whose rule is the superset ("clip" where the clips fill rounds of the chip, "fused" from 64 clips, else "staged"):
    pinned on a context it resolves to the right family in every mode ("clip" means "fused" in the 16-bit modes, everything means "staged" in fp32 -
    csrc/amuse_host.hpp resolve_path), so one pin serves a job whatever precision its calls use."""
    return _lib.plan(total, precision, tokens, arch)


def all_ranks_ok(ok: bool, device, group=None) -> bool:
    """Agree on a per-rank outcome BEFORE the next collective: an all-reduce (MIN) of an ok flag.  A rank whose share of the work raised (OOM, an audio-engine
    error on its WAVs) must not skip a collective its peers then block in forever - every rank calls this with its own flag and acts on the common answer."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return ok
    f = torch.tensor([1.0 if ok else 0.0], device=torch.device("cpu") if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
    return bool(f.item() == 1.0)


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
    set_clips_per_group (e.g. HipEngine.set_clips_per_group): called with job_plan(total)["clips_per_group"], and the shards are
                  aligned to it - sharded results are then bitwise those of one GPU running the whole job.
    set_decode_path (e.g. HipEngine.set_decode_path): called with job_plan(total)["decode_path"] for the same reason; both are
                  put back to the library's automatic rule afterwards.
    gather=False: returns the local shard's outputs (stay on this rank's device).
    gather=True : all ranks exchange shards (torch.distributed all_gather_object) and return the full batch
                  in global clip order - for tests and small jobs; large jobs should keep outputs sharded."""
    total = z_con.shape[0]
    tokens = 3 + (z_emo is not None) + (z_sty is not None)
    plan = job_plan(total, tokens)
    g = plan["clips_per_group"] if set_clips_per_group is not None else 1
    if set_clips_per_group is not None:
        set_clips_per_group(g)
    if set_decode_path is not None:
        set_decode_path(plan["decode_path"])
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
