"""Preflight of the N > 1 paths - the first thing to run when a multi-GPU node appears (none was available in rounds 1-6: the RCCL branch with more than one
rank has never executed; every N > 1 test so far used gloo on the CPU, two processes on ONE GPU, or an RCCL world of one).

    python tools/multigpu_preflight.py --gpus N            # one process per GPU, started as CHILDREN before anything touches the GPU (amuse_amd/launch.py)
    python tools/multigpu_preflight.py --gpus 2 --cpu      # the same protocol on gloo / CPU tensors (tests/test_launch_cpu.py runs this; no GPU needed)

Each step is one thing the product's N > 1 paths do, in the order they would first fail; rank 0 prints ONE JSON record {"ok": bool, "steps": [...]}, and the
exit code is non-zero if any step failed on any rank.  A step that raises on one rank is agreed on by an all-reduce of an ok flag BEFORE the next collective, so
no rank is left hanging inside a collective its peers never entered (the process group also carries a timeout).
  1 init        init_process_group("nccl" = RCCL, device_id = cuda:<LOCAL_RANK>)            bench.py, main.py --all-pairs, train_gesture.py
  2 barrier     one barrier + torch.cuda.synchronize()                                       bench.py's timing bracket
  3 all_reduce  the flat fp32 gradient bucket of train_gesture: 6,835,661 floats, SUM       train_gesture.py (ONE all-reduce per iteration)
  4 all_gather  3 x 256 floats per rank                                                      the all-pairs edit batch's embedding exchange (main.py)
  5 shard       a 32-clip shard of an N x 32-clip job sampled by every rank (DDIM-50 + decode through shard.sample_sharded: the JOB's amuse_plan pinned,
                global clip indices) and compared BITWISE with rank 0's own recomputation of that clip range inside the whole job (GPU only)
"""
import argparse
import datetime
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

GRAD_FLOATS = 6835661   # train_gesture's flat parameter / gradient buffer (Denoiser 2,192,384 + MotionPrior 4,643,277)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--cpu", action="store_true", help="gloo on CPU tensors (protocol check without GPUs); skips the sampling step")
    ap.add_argument("--timeout", type=float, default=120.0, help="seconds per collective before the process group aborts")
    args = ap.parse_args()
    from amuse_amd import launch
    if not launch.launched_by_torchrun():      # the parent: never touches the GPU, starts the ranks as children, hands back their exit code
        argv = ["--gpus", str(args.gpus), "--timeout", str(args.timeout)] + (["--cpu"] if args.cpu else [])
        return launch.run_ranks(str(Path(__file__).resolve()), argv, args.gpus, timeout=20 * args.timeout)

    import torch
    import torch.distributed as dist
    world, rank, local = (int(os.environ[k]) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev = torch.device("cpu") if args.cpu else torch.device(f"cuda:{local}")
    steps, state = [], {"ok": True}

    def agree(ok: bool) -> bool:
        """every rank learns whether ALL ranks passed the step (a gloo/RCCL all-reduce of a flag; before init: the local flag)"""
        if not dist.is_initialized():
            return ok
        f = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() == 1.0)

    def step(name, fn):
        if not state["ok"]:
            steps.append({"step": name, "ok": None, "skipped": "an earlier step failed"})
            return
        t0, err, info = time.perf_counter(), None, None
        try:
            info = fn()
        except Exception as e:  # noqa: BLE001 - reported, agreed on, never swallowed into a hang
            err = f"{type(e).__name__}: {e}"
        ok = agree(err is None)
        steps.append({"step": name, "ok": ok, "ms": round((time.perf_counter() - t0) * 1e3, 2), **({"error_rank%d" % rank: err} if err else {}), **(info or {})})
        state["ok"] = ok

    def init():
        if args.cpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=args.timeout))
        else:
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.timeout))
        return {"backend": dist.get_backend(), "world": dist.get_world_size(), "device": str(dev)}

    def barrier():
        dist.barrier()
        if not args.cpu:
            torch.cuda.synchronize()

    def all_reduce():
        g = torch.full((GRAD_FLOATS,), float(rank + 1), device=dev)
        if not args.cpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.all_reduce(g)
        if not args.cpu:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        want = world * (world + 1) / 2
        assert float(g.min()) == want and float(g.max()) == want, (float(g.min()), float(g.max()), want)
        return {"floats": GRAD_FLOATS, "all_reduce_ms": round(dt * 1e3, 3), "bus_GBps": round(2 * (world - 1) / world * GRAD_FLOATS * 4 / dt / 1e9, 2)}

    def all_gather():
        mine = torch.arange(3 * 256, dtype=torch.float32, device=dev) + 1000 * rank
        parts = [torch.empty(3 * 256, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine)
        for r, p in enumerate(parts):
            assert float(p[0]) == 1000 * r and float(p[-1]) == 1000 * r + 767, (r, float(p[0]))
        return {"floats_per_rank": 768}

    def shard():
        from amuse_amd import scheduler as sch, shard as sh, weights as wts
        from amuse_amd.engine import HipEngine
        per, total = 32, 32 * world
        eng = HipEngine(wts.make_denoiser_weights(0), wts.make_prior_weights(0), dev)
        try:
            eng.set_schedule(sch.ddim_table())
            gen = torch.Generator().manual_seed(99)
            con, emo, sty = (torch.randn(total, 256, generator=gen).to(dev) for _ in range(3))      # every rank draws the same job
            fn = lambda bsz, c, e, s, clip_index0=0: eng.diffusion_backward(c, e, s, "fp32x", seed=7, clip_index0=clip_index0)
            out = sh.sample_sharded(fn, con, emo, sty, rank, world, set_clips_per_group=eng.set_clips_per_group, set_decode_path=eng.set_decode_path)
            lo, hi = sh.shard_range(total, rank, world, align=sh.job_plan(total)["clips_per_group"])
            assert hi - lo == per and out["poses"].shape[0] == per
            mine = [out["latents"].cpu(), out["poses"].cpu()]
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            info = {"clips_per_rank": per, "job_plan": sh.job_plan(total)}
            if rank == 0:   # rank 0 recomputes the WHOLE job on its GPU: every shard must be bitwise its rows
                whole = fn(total, con, emo, sty)
                for r, (lat, poses) in enumerate(gathered):
                    a, b = sh.shard_range(total, r, world, align=sh.job_plan(total)["clips_per_group"])
                    assert torch.equal(lat, whole["latents"][a:b].cpu()) and torch.equal(poses, whole["poses"][a:b].cpu()), f"rank {r}'s shard differs from the single-GPU job"
                info["bitwise_equal_to_single_gpu"] = True
            return info
        finally:
            eng.close()

    step("init", init)
    step("barrier", barrier)
    step("all_reduce", all_reduce)
    step("all_gather", all_gather)
    if not args.cpu:
        step("shard", shard)
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if rank == 0:
        print(json.dumps({"tool": "multigpu_preflight", "ok": state["ok"], "world": world, "cpu": args.cpu, "steps": steps}), flush=True)
    return 0 if state["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
