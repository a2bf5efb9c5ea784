#!/usr/bin/env python3
"""bench.py - SMPL-X frames/sec of the MI355X sampling path (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch of synthetic clips: initial noise ->
T-step DDPM loop (persistent HIP kernel) -> VAE decode -> 6D->axis-angle, inputs (three 256-d
condition vectors per clip) already resident in HBM.  Workload at every N: BASELINE configs[2] /
SURVEY.md 8d config 3 - 256 x 10 s clips IN TOTAL, 1000-step DDPM, bf16 operands - sharded 256 / N
contiguous clips per rank through amuse_amd/shard.py (tiling and decode kernel chosen from the job's
total, counter-based noise keyed by the global clip index: every sharding reproduces the single-GPU
result bitwise; no collective on the data path): `value` = those 256 clips x 300 frames / the slowest
rank's time, "scaling": "strong".  A 1000-step chain costs the same 34 ms for 32 clips as for 256, so
that figure is flat in N by construction; the shape in which the path does scale
- every rank its own 256 clips, global clip indices rank * 256 ... - is measured beside it and reported
as `weak_scaling`.

  python bench.py [--gpus N --steps K --warmup W]
N > 1: one process per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a rank;
typed directly, it starts the N ranks itself as children (amuse_amd/launch.py - before anything touches the GPU) and
passes rank 0's line through.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import statistics
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

FLOP_PER_CLIP_STEP = 19_120_640          # linears 19,005,440 + attention 115,200
FLOP_VAE_DECODE_PER_CLIP = 1.76e9
FLOP_VAE_ATTN_PER_CLIP = 9 * 4 * 2 * 2 * 300 * 300 * 32   # 414.7 MFLOP: nine blocks x four heads x (Q K^T + P V) at S = 300, dh = 32
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3, "fp32x": 2500.0}   # MI355X_MICROARCH.md chip-level parameters (dense; fp32x = fp16 MFMAs)
KERNEL_NAME = {"bf16": "k_sample8", "fp16": "k_sample8h", "fp32": "k_sample<fp32>", "fp32x": "k_sample8x"}
# L2 -> CU weight stream per denoising step and CU (every CU re-streams the network each step): bytes per parameter of the MFMA stream
STREAM_MB_PER_STEP = {"bf16": 3.80, "fp16": 3.80, "fp32": 7.60, "fp32x": 7.60}
CU_LOAD_BYTES_PER_CLK = 64.0


PMC_DIRS = {"bf16": ("profiles/r06_pmc", "profiles/r05_pmc"), "fp32x": ("profiles/r06_fp32x_pmc", "profiles/r05_fp32x_pmc")}
PMC_KERNEL = {"bf16": "k_sample8", "fp32x": "k_sample8x"}


def pmc_traffic_bytes(clips, T, precision):
    """HBM bytes per k_sample launch.  NOT measured by this run (PMC counters need rocprofv3 around the process): read from the newest committed
    rocprofv3 PMC passes (separate --pmc runs of tools/run_sample_once.py at the bench shape) - and only if the pass's kernel_id.json (tools/kernel_id.py:
    sha256 over the kernel's source, its headers and the Makefile, and over the kernel's machine code read out of the libamuse_hip.so that is loaded, written when the pass was taken) equals the identity of the sampler in THIS
    tree - the same build inputs or, failing that, the same object bits:
    a kernel change without a PMC retake returns (None, reason) instead of a stale figure.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled
    per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream).  -> (bytes or None, source / reason)."""
    if (clips, T) != (256, 1000) or precision not in PMC_DIRS:
        return None, "no committed PMC pass at this shape / precision"
    import csv
    sys.path.insert(0, str(REPO / "tools"))
    from kernel_id import kernel_id
    d = next((x for x in PMC_DIRS[precision] if (REPO / x / "FETCH_SIZE_counter_collection.csv").exists()), None)
    if d is None:
        return None, "no committed PMC pass"
    idf = REPO / d / "kernel_id.json"
    if not idf.exists():
        return None, f"{d} carries no kernel_id.json: cannot tell whether it counted the kernel of this tree"
    counted, mine = json.load(open(idf)).get(PMC_KERNEL[precision], {}), kernel_id(PMC_KERNEL[precision])
    want, have = counted.get("source_sha256"), mine["source_sha256"]
    # the same build inputs - or the same OBJECT: an edit to a shared header or to the Makefile's file list that leaves the kernel's translation unit compiling to the
    # same bits has not changed the kernel that was counted
    same_object = bool(counted.get("object_sha256")) and counted.get("object_sha256") == mine.get("object_sha256")
    if want != have and not same_object:
        return None, (f"{d} counted {PMC_KERNEL[precision]} built from sources {str(want)[:12]} (object {str(counted.get('object_sha256'))[:12]}), this tree's are {have[:12]} "
                      f"(object {str(mine.get('object_sha256'))[:12]}): retake the PMC pass (tools/run_round_measurements.sh)")
    ident = f"source id {have[:12]}" if want == have else f"object {mine['object_sha256'][:12]} (the counted pass's; source ids differ: {str(want)[:12]} then, {have[:12]} now)"
    tot = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        f = REPO / d / f"{name}_counter_collection.csv"
        if not f.exists():
            return None, f"{d}/{name}_counter_collection.csv missing"
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
             if "k_sample" in r["Kernel_Name"] and r["Counter_Name"] == name]
        if not v:
            return None, f"{d}: no k_sample rows"
        tot[name] = sum(v) / len(v)
    return int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), f"{d}/*_counter_collection.csv (committed rocprofv3 --pmc passes of the kernel with {ident}, not this run)"


def host_cpu_info():
    """(model name, physical cores, logical cpus) from lscpu / /proc/cpuinfo; physical falls back to logical // 2 (SMT) when neither says."""
    import subprocess
    logical = os.cpu_count() or 1
    model, phys = "unknown", None
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        model = kv.get("Model name", model)
        if "Core(s) per socket" in kv and "Socket(s)" in kv:
            phys = int(kv["Core(s) per socket"]) * int(kv["Socket(s)"])
    except Exception:
        pass
    if phys is None:
        try:
            ids = set()
            cur = {}
            for l in open("/proc/cpuinfo"):
                if ":" in l:
                    k, v = (x.strip() for x in l.split(":", 1))
                    cur[k] = v
                elif cur:
                    ids.add((cur.get("physical id"), cur.get("core id")))
                    cur = {}
            phys = len(ids) if len(ids) > 1 else None
        except Exception:
            phys = None
    return model, int(phys or max(1, logical // 2)), logical


def cpu_baseline(clips, T, wd, wp):
    """The oracle (CPU restatement of the reference's PyTorch path: unfused fp32 torch ops, same algorithm and op order) on this box's host
    cores - BASELINE.md section 4's three points, each at `physical cores` threads AND at 16 threads (these small ops do not scale to a
    big host's core count), medians after warm-ups:
      (i)   B = 1, DDIM-50 loop + decode + 6D -> axis-angle (config 1)           3 warm-ups + 10 repetitions
      (ii)  B = `clips`, DDPM: a 20-step slice of the 1000-step loop             3 + 10 slices; the job = T x step + (iii), stated as extrapolated
      (iii) VAE decode + 6D -> axis-angle of B = `clips` alone                   1 + 3
    `value` = point (ii)'s whole-job frames/s at the better of the two thread counts."""
    import torch
    from oracle import amuse_oracle as orc
    Wd, Wp = orc.to_torch(wd), orc.to_torch(wp)
    model, phys, logical = host_cpu_info()
    gen = torch.Generator().manual_seed(7)
    con, emo, sty, x = (torch.randn(clips, n, generator=gen) for n in (256, 256, 256, 128))
    nz = torch.randn(clips, 128, generator=gen)
    z = torch.randn(clips, 128, generator=gen)
    ddpm, ddim = orc.DDPM(T), orc.DDIM()
    SLICE = 20

    BUDGET_S = 8.0        # per point and thread count: 3 warm-ups + 10 repetitions where they fit, never fewer than 1 + 3 (a 128-thread run of these small
    reps_used = {}        # ops is ~10 x slower than the 16-thread one: the prescribed repetition count alone would take minutes there)

    def med(fn, warm, reps, tag=None):
        t0 = time.perf_counter()
        fn()
        first = time.perf_counter() - t0
        if first * (warm + reps) > BUDGET_S:
            warm, reps = 1, max(3, min(reps, int(BUDGET_S / max(first, 1e-9)) - 1))
        ts = []
        for i in range(warm - 1 + reps):           # (the probe above was the first warm-up)
            t0 = time.perf_counter()
            fn()
            if i >= warm - 1:
                ts.append(time.perf_counter() - t0)
        if tag:
            reps_used[tag] = f"{warm}+{reps}"
        return statistics.median(ts)

    def job1():
        lat = orc.sample_latents(Wd, ddim, con[:1], emo[:1], sty[:1], x[:1])
        orc.feats_to_smplx(orc.vae_decode(Wp, lat))

    def slice_ddpm():
        xx = x
        for i in range(SLICE):
            t = ddpm.timesteps[i]
            xx = ddpm.step(orc.denoiser_forward(Wd, xx, t, con, emo, sty), t, xx, nz)

    def decode_all():
        orc.feats_to_smplx(orc.vae_decode(Wp, z))

    pts = {}
    old = torch.get_num_threads()
    with torch.no_grad():
        for threads in dict.fromkeys((phys, min(16, logical))):
            torch.set_num_threads(threads)
            t1 = med(job1, 3, 10, f"b1@{threads}")
            ts = med(slice_ddpm, 3, 10, f"slice@{threads}") / SLICE
            td = med(decode_all, 1, 3, f"decode@{threads}")
            total = T * ts + td
            pts[threads] = {"threads": threads,
                            "b1_ddim50_ms_per_clip": round(t1 * 1e3, 2), "b1_ddim50_frames_per_s": round(300 / t1, 1),
                            f"b{clips}_ddpm_ms_per_step": round(ts * 1e3, 3), f"b{clips}_decode_s": round(td, 3),
                            f"b{clips}_ddpm{T}_frames_per_s": round(clips * 300 / total, 1), f"b{clips}_ddpm{T}_s_per_job_extrapolated": round(total, 2)}
    torch.set_num_threads(old)
    key = f"b{clips}_ddpm{T}_frames_per_s"
    best = max(pts.values(), key=lambda d: d[key])
    return {"value": best[key], "unit": "frames/s", "cores": best["threads"], "physical_cores": phys, "host_cores": logical, "cpu_model": model, "kind": "port",
            "points": list(pts.values()), "warmups_plus_repetitions": reps_used,
            "sample": f"oracle/amuse_oracle.py, fp32 torch CPU ops, medians: (i) B=1 DDIM-50 whole job, 3 warm-ups + 10 repetitions; (ii) B={clips} DDPM: "
                      f"{SLICE}-step slices of the {T}-step loop, 3 + 10 slices, job = {T} x step + (iii) (extrapolated); (iii) decode + 6D->axis-angle of "
                      f"{clips} clips, 1 + 3; each at {phys} threads (physical cores) and at {min(16, logical)}; a point whose repetitions would exceed {BUDGET_S:.0f} s runs 1 warm-up + >= 3 "
                      f"repetitions instead (`warmups_plus_repetitions`); `value` / `cores` = the faster of the two"}


def wellcond_pose_error(dev):
    """North star: "per-joint L2 vs reference < 1e-4".  The whole path (x_T -> DDIM-50 -> MotionPrior.decode -> 6D -> axis-angle, ONE amuse_diffusion_backward call)
    against the fixture the reference's own modules wrote for a second weight draw with a well-conditioned decoder (tests/golden/wellcond.npz: data, not the
    oracle; oracle/gen_golden.py --wellcond): max over EVERY joint of 4 clips x 300 frames x 55 joints, per mode; the 16-bit modes as rotation distances."""
    import numpy as np
    from amuse_amd import scheduler as sch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    g = np.load(REPO / "tests" / "golden" / "wellcond.npz")
    eng = HipEngine(wts.make_denoiser_weights(1), wts.make_wellcond_prior_weights(1), dev)
    try:
        eng.set_schedule(sch.ddim_table())
        out = {"fixture": "tests/golden/wellcond.npz full/* (reference Denoiser x DDIM-50 -> reference MotionPrior.decode -> rotation_6d_to_matrix -> p3d matrix_to_axis_angle)",
               "joints": int(g["full/poses"].size // 3)}
        for mode in ("fp32x", "fp32", "bf16", "fp16"):
            o = eng.diffusion_backward(g["con"], g["emo"], g["sty"], mode, x_init=g["x_T"])
            d = np.linalg.norm(o["poses"].cpu().numpy() - g["full/poses"], axis=-1)
            out[mode] = {"pose_l2_max": float(d.max()), "pose_l2_median": float(np.median(d)), "latents_max_err": float(np.abs(o["latents"].cpu().numpy() - g["full/latents"]).max())}
        return out
    finally:
        eng.close()


def diffusion_only_extra(dev, precision, peak):
    """The Denoiser's diffusion_only variant (denoiser.py:64-66,177-187; arch trans_enc): the transformer BASELINE's north star
    describes - self-attention over ~300 frames in EVERY denoising step (S = 304 = 4 condition tokens + 300 pose frames), sampled
    with the reference's DDIM-50.  Reports the job, the step kernel against the MFMA peak, and the attention-only figure
    (full launch minus the no-attention instantiation, as for `decode.attention`)."""
    import numpy as np
    import torch
    from amuse_amd import scheduler as sch
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine
    S = 304
    flop_step = 9 * S * 393216 + 4 * S * 65536 + 2 * (2 * 300 * 333 * 128)      # blocks + skip linears + pose_embd / pose_proj
    flop_attn = 9 * 4 * 2 * 2 * S * S * 32
    eng = HipEngine(wts.make_denoiser_weights(0, "trans_enc", True), None, dev, arch="trans_enc", diffusion_only=True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {"arch": "diffusion_only + trans_enc: S = 304 rows per clip and step (denoiser.py:177-187)", "sampler": "ddim-50", "precision": precision,
           "flop_per_clip_step": flop_step + flop_attn, "attention_flop_per_clip_step": flop_attn, "jobs": []}
    try:
        gold = np.load(REPO / "tests" / "golden" / "denoiser_variants.npz")
        x = gold["x_pose"].astype(np.float32)
        out["eps_err_vs_reference_golden"] = {
            m: max(float(np.abs(eng.denoise_step(x, t, gold["con"], gold["emo"], gold["sty"], m).cpu().numpy()[:, 0:300:6]
                                - gold[f"trans_enc_pose/eps_t{t}"]).max()) for t in (981, 501, 1)) for m in ("fp32x", precision)}
        eng.set_schedule(sch.ddim_table())
        gen = torch.Generator().manual_seed(4321)
        for B in (64, 256):
            con, emo, sty = (torch.randn(B, 256, generator=gen).to(dev) for _ in range(3))

            def job_ms(reps=3):
                ts = []
                for i in range(reps + 1):
                    ev0.record()
                    eng.sample(con, emo, sty, precision, seed=2024)
                    ev1.record()
                    ev1.synchronize()
                    if i >= 1:
                        ts.append(ev0.elapsed_time(ev1))
                return min(ts)
            ms = job_ms()
            j = {"clips": B, "ms_per_job": round(ms, 3), "ms_per_step": round(ms / 50, 4), "frames_per_s": round(B * 300 / ms * 1e3, 1),
                 "tflops": round(B * 50 * (flop_step + flop_attn) / (ms * 1e-3) / 1e12, 1),
                 "frac_of_mfma_peak": round(B * 50 * (flop_step + flop_attn) / (ms * 1e-3) / 1e12 / peak, 4),
                 "kernel": "k_den_fused (one persistent workgroup per clip and step)" if B >= 64 else "staged k_vae_rows / k_vae_attn"}
            if B >= 64:
                eng.set_ablation(1)
                ms_na = job_ms(2)
                eng.set_ablation(0)
                att = (ms - ms_na) / 50
                j["attention"] = {"ms_per_step": round(att, 4), "tflops": round(B * flop_attn / (att * 1e-3) / 1e12, 1),
                                  "frac_of_mfma_peak": round(B * flop_attn / (att * 1e-3) / 1e12 / peak, 4),
                                  "method": "step kernel minus its no-attention instantiation (amuse_debug_set_ablation), HIP events"}
            if B == 256 and precision != "fp32x":   # the parity mode's step at the same shape: DDIM-10, per step (k_den_fusedx)
                eng.set_schedule(sch.ddim_table(10))
                ts = []
                for i in range(3):
                    ev0.record()
                    eng.sample(con, emo, sty, "fp32x", seed=2024)
                    ev1.record()
                    ev1.synchronize()
                    if i >= 1:
                        ts.append(ev0.elapsed_time(ev1) / 10)
                eng.set_schedule(sch.ddim_table())
                j["fp32x_ms_per_step"] = round(min(ts), 4)
                j["fp32x_kernels"] = "k_den_fusedx: the whole step as one persistent workgroup per clip (pose_embd, nine encoder blocks, pose_proj + scheduler update)"
            out["jobs"].append(j)
    finally:
        eng.close()
    return out


def edit_batch_extra(eng, dev, rank, world, red_dev, share_gpu, precision="bf16", reps=5):
    """BASELINE config 5's shape on `world` ranks, the way `main.py --fn edit_gesture --all-pairs --gpus N` runs it (amuse_amd/trainer.py): 8 source + 8
    target waveforms (10 s, 16 kHz, resident in HBM) -> each rank embeds its contiguous share of the 16 WAVs (kaldi fbank + 3 x AST) -> the embeddings
    (3 x 256 floats per WAV) are all-gathered (RCCL; the path's one exchange) -> 8 x 8 = 64 jobs (content / style of source i, emotion of target j), this
    rank's contiguous range sampled with DDIM-50 (the reference's edit sampler) and decoded to SMPL-X.  Timed between barriers, max over ranks."""
    import torch
    import torch.distributed as dist
    from amuse_amd import audio_weights as aw
    from amuse_amd import scheduler as sch
    from amuse_amd import shard
    from amuse_amd.audio import AudioEngine
    aeng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS), device=dev)
    try:
        gen = torch.Generator().manual_seed(55)
        wav = (0.1 * torch.randn(16, 160000, generator=gen)).to(dev)          # every rank holds the same 16 waveforms and reads its share
        wlo, whi = shard.shard_range(16, rank, world)
        per = -(-16 // world)
        jobs = [(i, 8 + j) for i in range(8) for j in range(8)]
        g = shard.job_plan(64)["clips_per_group"]
        ja, jb = shard.job_range([1] * 64, rank, world, align=g)
        eng.set_schedule(sch.ddim_table())
        gdev = torch.device("cpu") if share_gpu else dev

        def once():
            emb = torch.zeros(per, 3, 256, device=dev)
            err = None
            try:
                if whi > wlo:
                    emb[:whi - wlo] = torch.stack(aeng.features(wav[wlo:whi]), dim=1)
            except Exception as e:  # noqa: BLE001 - a rank that fails on its share must not leave the others inside the all_gather: agree first
                err = e
            if not shard.all_ranks_ok(err is None, dev):
                raise err if err is not None else RuntimeError("another rank failed in its share of the audio front-end")
            if world > 1:
                parts = [torch.empty(per, 3, 256, device=gdev) for _ in range(world)]
                dist.all_gather(parts, emb.to(gdev))
                allv = torch.cat([parts[r][:b - a] for r in range(world) for a, b in [shard.shard_range(16, r, world)]]).to(dev)
            else:
                allv = emb
            if jb > ja:
                ii = torch.tensor([jobs[k][0] for k in range(ja, jb)], device=dev)
                jj = torch.tensor([jobs[k][1] for k in range(ja, jb)], device=dev)
                eng.set_clips_per_group(g)
                eng.set_decode_path(shard.job_plan(64)["decode_path"])
                try:
                    return eng.diffusion_backward(allv[ii, 0], allv[jj, 1], allv[ii, 2], precision, seed=2024, clip_index0=ja)
                finally:
                    eng.set_clips_per_group(0)
                    eng.set_decode_path("auto")

        def bar():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        once()
        ts = []
        for _ in range(reps):
            bar()
            t0 = time.perf_counter()
            o = once()
            bar()
            t = torch.tensor([time.perf_counter() - t0], device=red_dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ts.append(float(t.item()))
        assert o is None or bool(torch.isfinite(o["poses"]).all())
        ms = statistics.median(ts) * 1e3
        return {"workload": "16 x 10 s WAVs (8 sources + 8 targets) -> audio front-end -> 8 x 8 = 64 edit jobs, DDIM-50 + decode", "n_gpus": world,
                "precision": precision, "wavs_per_rank": [list(shard.shard_range(16, r, world)) for r in range(world)],
                "jobs_per_rank": [list(shard.job_range([1] * 64, r, world, align=g)) for r in range(world)],
                "exchange": "none (one rank)" if world == 1 else ("all_gather of 16 x 3 x 256 fp32 embeddings, " + ("gloo (ranks share a GPU)" if share_gpu else "RCCL")),
                "ms_per_batch": round(ms, 3), "frames_per_s": round(64 * 300 / (ms * 1e-3), 1), "reps": reps, "timing": "median, barrier to barrier, max over ranks"}
    finally:
        aeng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=256, help="clips of the job IN TOTAL (sharded over the ranks at N > 1; the weak-scaling companion gives every rank this many)")
    ap.add_argument("--T", type=int, default=1000, help="DDPM steps")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32", "fp32x"])
    ap.add_argument("--config", default="sample", choices=["sample", "train"],
                    help="sample: BASELINE configs[2] (the headline metric); train: configs[3] train_gesture data-parallel step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-audio", action="store_true", help="skip the audio front-end side measurement")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--train-batch", dest="train_batch", type=int, default=32, help="--config train: clips per GPU and iteration (BASELINE config 4: 32)")
    ap.add_argument("--no-torch-baseline", dest="no_torch_baseline", action="store_true", help="skip the PyTorch-ROCm eager restatement of the job (torch_eager_baseline, ~3 s)")
    ap.add_argument("--no-graph", dest="no_graph", action="store_true", help="--config train: the eager step instead of the two captured HIP graphs (A/B)")
    ap.add_argument("--edit-batch", action="store_true", help="with --no-extras: still run the edit_batch extra (BASELINE config 5's shape, every rank takes part)")
    args = ap.parse_args()
    from amuse_amd import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        # typed as `python bench.py --gpus N`: this process becomes the launcher of N ranks and never initialises the GPU
        raise SystemExit(launch.run_ranks(str(Path(__file__).resolve()), sys.argv[1:], args.gpus))
    if args.config == "train":
        from amuse_amd import train_gesture
        return train_gesture.bench_main(args)

    import torch
    import torch.distributed as dist
    from amuse_amd import scheduler as sch
    from amuse_amd import shard
    from amuse_amd import weights as wts
    from amuse_amd.engine import HipEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # AMUSE_BENCH_SHARE_GPU=1 (tests on a one-GPU box): all ranks on cuda:0, gloo for the two scalar reductions - exercises the N > 1
    # logic of this file end to end; the timings of ranks that share a GPU mean nothing
    share_gpu = world > 1 and os.environ.get("AMUSE_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if share_gpu else dev   # where the reduced scalars live (gloo reduces host tensors)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wd, wp = wts.make_denoiser_weights(0), wts.make_prior_weights(0)
    eng = HipEngine(wd, wp, dev)
    eng.set_schedule(sch.ddpm_table(args.T))
    total = args.clips
    gen = torch.Generator().manual_seed(1234)            # every rank draws the SAME global batch and takes its shard
    con, emo, sty = (torch.randn(total, 256, generator=gen).to(dev) for _ in range(3))
    g_job = shard.job_plan(total)["clips_per_group"]
    lo, hi = shard.shard_range(total, rank, world, align=g_job)
    B = hi - lo
    out = {"latents": torch.empty(max(B, 1), 128, device=dev), "poses": torch.empty(max(B, 1), 300, 55, 3, device=dev),
           "trans": torch.empty(max(B, 1), 300, 3, device=dev)}

    def sample_fn(bsz, c, e, s, clip_index0=0):
        return eng.diffusion_backward(c, e, s, args.precision, seed=2024, clip_index0=clip_index0, out=out)

    def step_strong():   # the product's sharding path: tiling from the job's total, aligned contiguous shard, global clip index
        shard.sample_sharded(sample_fn, con, emo, sty, rank, world, set_clips_per_group=eng.set_clips_per_group,
                             set_decode_path=eng.set_decode_path)

    step = step_strong   # the timed workload at every N: BASELINE config 3 - `total` clips in all, this rank's contiguous shard
    if world > 1:
        # weak-scaling companion: this rank's own `total` clips (another draw per rank), global clip indices rank * total ...; the
        # same code path as a whole single-GPU job (tiling and decode kernel chosen from the job's clip count)
        con_w, emo_w, sty_w = (torch.randn(total, 256, generator=torch.Generator().manual_seed(99 + rank)).to(dev) for _ in range(3))
        out_w = {"latents": torch.empty(total, 128, device=dev), "poses": torch.empty(total, 300, 55, 3, device=dev),
                 "trans": torch.empty(total, 300, 3, device=dev)}

        def sample_w(bsz, c, e, s, clip_index0=0):
            return eng.diffusion_backward(c, e, s, args.precision, seed=2024, clip_index0=rank * total + clip_index0, out=out_w)

        def step_weak():
            shard.sample_sharded(sample_w, con_w, emo_w, sty_w, 0, 1, set_clips_per_group=eng.set_clips_per_group,
                                 set_decode_path=eng.set_decode_path)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert B == 0 or bool(torch.isfinite(out["poses"][:B]).all())

    # ---- weak-scaling companion, N > 1 only: every rank its own `total` clips (the shape in which the path scales)
    weak = None
    if world > 1:
        step_weak()
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            step_weak()
        barrier()
        tw = torch.tensor([time.perf_counter() - t1], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak = {"clips_total": world * total, "clips_per_gpu": total, "frames_per_s": round(world * total * 300 * 3 / float(tw.item()), 1),
                "ms_per_job": round(float(tw.item()) / 3 * 1e3, 3),
                "clip_range_per_rank": [[r * total, (r + 1) * total] for r in range(world)],
                "note": "independent clip batches per rank, no collective: scales with N because nothing is exchanged"}
        assert bool(torch.isfinite(out_w["poses"]).all())
        del con_w, emo_w, sty_w, out_w

    # ---- BASELINE config 5 beside the headline, at every N (all ranks take part: audio shares, one all-gather, job ranges)
    edit_batch = None
    if (not args.no_extras or args.edit_batch) and not args.no_audio and args.precision in ("bf16", "fp16", "fp32x"):
        try:
            edit_batch = edit_batch_extra(eng, dev, rank, world, red_dev, share_gpu, args.precision)
        except Exception as e:          # the headline must not depend on an extra.  A rank that fails on ITS share agrees with the others before the exchange
            edit_batch = {"error": f"{type(e).__name__}: {e}"}   # (shard.all_ranks_ok inside): all ranks land here together; anything else runs into the group's timeout
        eng.set_schedule(sch.ddpm_table(args.T))

    # ---- dominant kernel (k_sample: the T-step loop) timed live with HIP events on its launch stream
    # (HipEngine launches on torch's current stream, which is the stream these events are recorded on)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kt = []
    eng.set_clips_per_group(g_job)
    for _ in range(max(3, min(args.steps, 10))):
        if B == 0:
            break
        ev0.record()
        eng.sample(con[lo:hi], emo[lo:hi], sty[lo:hi], args.precision, seed=2024, clip_index0=lo)
        ev1.record()
        ev1.synchronize()
        kt.append(ev0.elapsed_time(ev1) * 1e-3)
    eng.set_clips_per_group(0)
    k_avg = sum(kt) / len(kt) if kt else float("nan")

    line = None
    if rank == 0:
        value = total * 300 * args.steps / elapsed
        flop = B * args.T * FLOP_PER_CLIP_STEP
        achieved = flop / k_avg / 1e12
        peak = MFMA_PEAK_TFLOPS[args.precision]
        traffic, traffic_src = pmc_traffic_bytes(B, args.T, args.precision)
        us_step = k_avg / args.T * 1e6
        clk_ghz = getattr(torch.cuda.get_device_properties(dev), "clock_rate", 2.4e6) / 1e6   # kHz -> GHz (2.4 if torch does not say)
        stream_floor_us = STREAM_MB_PER_STEP[args.precision] * 1e6 / CU_LOAD_BYTES_PER_CLK / (clk_ghz * 1e9) * 1e6
        line = {
            "metric": "SMPL-X frames/sec (10 s clip, 1000-step DDPM)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "world_size_seen": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{total} x 10 s clips in total ({B} on rank 0), DDPM-{args.T} sampling loop + "
                                   f"VAE decode (300 frames) + 6D->axis-angle; random-init weights of the "
                                   f"diff_latent_v2 / prior_emotional_fing architecture",
                       "clips_total": total, "clips_rank0": B, "clips_per_tile": g_job, "sampler": f"ddpm-{args.T}",
                       "sharding": f"contiguous clip shards x{world} of ONE {total}-clip job (amuse_amd/shard.py; bitwise the single-GPU result), no collectives",
                       "clip_range_per_rank": [list(shard.shard_range(total, r, world, align=g_job)) for r in range(world)],
                       "mfma_operands": args.precision, "state_and_accumulate": "fp32",
                       "scheduler_arithmetic": "diffusers 0.17.1 DDPM (fixed_small) / DDIM restated (package not in the image); pinned in-kernel against the "
                                               "reference tree's own GaussianDiffusion / SpacedDiffusion (tests/test_gpu_pins.py, tests/golden/sched_ref.npz); "
                                               "unpinned conventions: DDIM last-step alpha_bar_prev = abar[0], un-clipped eps_hat in the DDIM direction term, "
                                               "fp32 cumprod (DESIGN.md section 2)"},
            "roofline": {"bound": "chain+l2_stream", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": KERNEL_NAME[args.precision] + " (persistent T-step denoising loop)",
                         "kernel_ms": round(k_avg * 1e3, 3),
                         "us_per_denoising_step": round(us_step, 3),
                         "stream_floor_us_per_step": round(stream_floor_us, 2),
                         "note": f"`bound` names what binds the kernel (the serial per-step dependency chain + the per-CU L2->CU weight stream); "
                                 f"`frac` stays achieved / dense MFMA peak. Algorithmic FLOPs = clips x T x 19,120,640 (rank 0's {B} clips); measured {us_step:.2f} us per "
                                 f"denoising step = {us_step / stream_floor_us:.2f} x the floor of its L2->CU weight stream "
                                 f"({STREAM_MB_PER_STEP[args.precision]:.2f} MB per CU and step at {CU_LOAD_BYTES_PER_CLK:.0f} B/clk = "
                                 f"{stream_floor_us:.1f} us at the {clk_ghz:.2f} GHz this device reports). The kernel is bound by the "
                                 f"serial per-step dependency chain + that stream, not by HBM or MFMA issue (DESIGN.md 4.1, 4.1b, 5)"},
        }
        if weak is not None:
            line["weak_scaling"] = weak
        if edit_batch is not None:
            line["edit_batch"] = edit_batch
    if rank == 0 and not args.no_extras:
        # single-clip latency (BASELINE configs[1]): B = 1, same sampler, 5 warm-ups, 50 HIP-event-timed repeats
        c1, e1, s1 = con[:1].contiguous(), emo[:1].contiguous(), sty[:1].contiguous()
        o1 = {"latents": torch.empty(1, 128, device=dev), "poses": torch.empty(1, 300, 55, 3, device=dev),
              "trans": torch.empty(1, 300, 3, device=dev)}
        lat = []
        for i in range(55):
            ev0.record()
            eng.diffusion_backward(c1, e1, s1, args.precision, seed=2024, out=o1)
            ev1.record()
            ev1.synchronize()
            if i >= 5:
                lat.append(ev0.elapsed_time(ev1))
        line["p50_clip_latency_ms"] = round(statistics.median(lat), 3)
        line["p50_clip_latency_samples"] = len(lat)
        # decode alone (MotionPrior.decode + 6D->axis-angle of rank 0's clips), HIP events
        if B > 0:
            z = torch.randn(B, 128, generator=gen).to(dev)
            dt = []
            for i in range(48):   # (the single-clip runs before this leave the GPU at a low clock: 0.64 ms kernels need tens of launches to ramp it)
                ev0.record()
                eng.vae_decode(z, None, args.precision)
                ev1.record()
                ev1.synchronize()
                if i >= 32:
                    dt.append(ev0.elapsed_time(ev1))
            flop_dec = B * FLOP_VAE_DECODE_PER_CLIP
            line["decode"] = {"clips": B, "ms": round(min(dt), 3), "tflops": round(flop_dec / (min(dt) * 1e-3) / 1e12, 1),
                              "frac_of_mfma_peak": round(flop_dec / (min(dt) * 1e-3) / 1e12 / peak, 4)}
            if args.precision in ("bf16", "fp16") and B >= 64:
                # attention-only roofline of the S = 300 self-attention (the "fraction of the attention roofline" of BASELINE's north star):
                # its time = the fused decode launch minus the same launch WITHOUT softmax(Q K^T) V (amuse_debug_set_ablation: a second
                # instantiation of the kernel - projections, K / V images, out_proj and every barrier stay), both timed here by HIP events
                eng.set_ablation(1)
                da = []
                for i in range(12):
                    ev0.record()
                    eng.vae_decode(z, None, args.precision)
                    ev1.record()
                    ev1.synchronize()
                    if i >= 4:
                        da.append(ev0.elapsed_time(ev1))
                eng.set_ablation(0)
                att_ms = min(dt) - min(da)
                line["decode"]["attention"] = {
                    "flop_per_clip": FLOP_VAE_ATTN_PER_CLIP, "ms": round(att_ms, 4), "tflops": round(B * FLOP_VAE_ATTN_PER_CLIP / (att_ms * 1e-3) / 1e12, 1),
                    "frac_of_mfma_peak": round(B * FLOP_VAE_ATTN_PER_CLIP / (att_ms * 1e-3) / 1e12 / peak, 4), "launch_without_attention_ms": round(min(da), 3),
                    "method": "k_vae_fused launch time minus its no-attention instantiation's, HIP events, same process",
                    "bound": "transcendental issue: 80 v_exp_f32 per 16-query tile and head against 40 useful MFMAs (dh = 32); lazy rescaling of the running maxima since round 6 "
                             "(0.27 -> 0.31); the 32x32x16 one-wave-per-SIMD regime probed at 0.32-0.37: profiles/r06_attention_lazy_rescale.txt, r06_attend32_probe.txt, DESIGN.md 4.2"}
        # the step time of k_sample does not depend on the clips per workgroup tile (1..3), so 3 clips per CU
        # cost the same loop time: report that saturating point too (not the headline workload)
        Bs = 3 * total
        cs, es, ss = (torch.randn(Bs, 256, generator=gen).to(dev) for _ in range(3))
        outs = {"latents": torch.empty(Bs, 128, device=dev), "poses": torch.empty(Bs, 300, 55, 3, device=dev),
                "trans": torch.empty(Bs, 300, 3, device=dev)}
        ts = []
        for i in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.diffusion_backward(cs, es, ss, args.precision, seed=2024, out=outs)
            torch.cuda.synchronize()
            if i >= 1:
                ts.append(time.perf_counter() - t1)
        flop_job = Bs * (args.T * FLOP_PER_CLIP_STEP + FLOP_VAE_DECODE_PER_CLIP)
        line["saturating_point"] = {"clips_per_gpu": Bs, "frames_per_s": round(Bs * 300 / min(ts), 1),
                                    "ms_per_job": round(min(ts) * 1e3, 3),
                                    "frac_of_mfma_peak": round(flop_job / min(ts) / 1e12 / MFMA_PEAK_TFLOPS[args.precision], 4),
                                    "note": "three clips per 16-row tile on all 256 CUs: every further 768 clips add one more round of the same ~35 ms (a 4,096-clip job = 6 rounds, "
                                            "the last one a third full: 5.7 M frames/s measured: `at_4096_clips`); the two-tile kernel that would lift this was costed at 1.24 x and not built "
                                            "(profiles/r05_k_sample8_two_tile_ablation.txt)"}
        del cs, es, ss, outs
        if world == 1:   # SURVEY 8d also asks for a point at >= 4,096 clips per GPU: six rounds of 768 clips, the last a third full
            B4 = 4096
            c4, e4, s4 = (torch.randn(B4, 256, generator=gen).to(dev) for _ in range(3))
            o4 = {"latents": torch.empty(B4, 128, device=dev), "poses": torch.empty(B4, 300, 55, 3, device=dev), "trans": torch.empty(B4, 300, 3, device=dev)}
            t4 = []
            for i in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                eng.diffusion_backward(c4, e4, s4, args.precision, seed=2024, out=o4)
                torch.cuda.synchronize()
                if i >= 1:
                    t4.append(time.perf_counter() - t1)
            line["saturating_point"]["at_4096_clips"] = {"frames_per_s": round(B4 * 300 / min(t4), 1), "ms_per_job": round(min(t4) * 1e3, 3),
                                                         "frac_of_mfma_peak": round(B4 * (args.T * FLOP_PER_CLIP_STEP + FLOP_VAE_DECODE_PER_CLIP) / min(t4) / 1e12 / MFMA_PEAK_TFLOPS[args.precision], 4)}
            del c4, e4, s4, o4
        if world == 1 and B > 0:
            # parity mode: the SAME job in the fp32x mode (split-fp16 MFMA operands, fp32 everything else) - the mode that
            # meets the north-star tolerance; eps_err = teacher-forced eps_hat against the reference modules' golden
            # vectors (tests/golden/denoiser_steps.npz: data, not the oracle), bar 1e-5
            import numpy as np
            gold = np.load(REPO / "tests" / "golden" / "denoiser_steps.npz")
            eps_err = {}
            for mode in ("fp32x", "fp32", "bf16", "fp16"):
                eps_err[mode] = max(float(np.abs(eng.denoise_step(gold["x_t"], t, gold["con"], gold["emo"], gold["sty"], mode).cpu().numpy()
                                                 - gold[f"eps_t{t}"]).max()) for t in (981, 501, 1))

            def time_job(mode, reps):
                ts_ = []
                for i in range(reps + 1):
                    ev0.record()
                    eng.diffusion_backward(con[lo:hi], emo[lo:hi], sty[lo:hi], mode, seed=2024, clip_index0=lo, out=out)
                    ev1.record()
                    ev1.synchronize()
                    if i >= 1:
                        ts_.append(ev0.elapsed_time(ev1))
                return min(ts_)
            eng.set_clips_per_group(g_job)
            ms_x, ms_f = time_job("fp32x", 3), time_job("fp32", 2)
            # its sampling kernel alone (k_sample8x), HIP events; counters from the committed PMC passes of the same shape
            ktx = []
            for _ in range(3):
                ev0.record()
                eng.sample(con[lo:hi], emo[lo:hi], sty[lo:hi], "fp32x", seed=2024, clip_index0=lo)
                ev1.record()
                ev1.synchronize()
                ktx.append(ev0.elapsed_time(ev1))
            kx = min(ktx)
            tfx = B * args.T * FLOP_PER_CLIP_STEP / (kx * 1e-3) / 1e12
            trx, trx_src = pmc_traffic_bytes(B, args.T, "fp32x")
            try:
                wc = wellcond_pose_error(dev)
            except Exception as e:  # noqa: BLE001 - an extra must not take the headline down
                wc = {"error": f"{type(e).__name__}: {e}"}
            line["parity_mode"] = {"precision": "fp32x", "ms_per_job": round(ms_x, 3), "frames_per_s": round(B * 300 / ms_x * 1e3, 1),
                                   "pose_l2_max": wc.get("fp32x", {}).get("pose_l2_max"), "pose_l2_bar": 1e-4, "end_to_end_vs_reference_modules": wc,
                                   "roofline": {"kernel": "k_sample8x (split-fp16 operands: 3 MFMAs per product)", "kernel_ms": round(kx, 3),
                                                "us_per_denoising_step": round(kx / args.T * 1e3, 2),
                                                "achieved_algorithmic_tflops": round(tfx, 1), "mfma_tflops_issued": round(3 * tfx, 1),
                                                "frac_of_f16_mfma_peak_issued": round(3 * tfx / MFMA_PEAK_TFLOPS["fp32x"], 4),
                                                "stream_floor_us_per_step": round(STREAM_MB_PER_STEP["fp32x"] * 1e6 / CU_LOAD_BYTES_PER_CLK / (clk_ghz * 1e9) * 1e6, 2),
                                                "traffic": trx, "traffic_source": trx_src,
                                                "traffic_note": "7.6 MB of weights per step exceed the 4 MB L2 of an XCD: every XCD re-fetches the stream each "
                                                                "step (8 x 7.6 MB x T = 61 GB per launch, fabric / Infinity-Cache side, ~0.9 TB/s) - not the bound "
                                                                "(the per-CU load path is); the bf16 stream (3.8 MB) stays L2-resident"},
                                   "us_per_denoising_step_incl_decode": round(ms_x / args.T * 1e3, 2),
                                   "eps_err": eps_err["fp32x"], "eps_bar": 1e-5,
                                   "fp32_mode": {"ms_per_job": round(ms_f, 3), "frames_per_s": round(B * 300 / ms_f * 1e3, 1), "eps_err": eps_err["fp32"]},
                                   "bf16_eps_err": eps_err["bf16"],
                                   "what": "same clips, DDPM-%d + decode + 6D->axis-angle; eps_err = max |eps_hat - reference golden| at t = 981, 501, 1" % args.T}
            # fp16 throughput mode: the bf16 mode's kernels built for fp16 MFMA operands (same bytes, same MFMA rate, 11 significand
            # bits instead of 8) - the same job and the same eps_err measure
            ms_h, ms_b = time_job("fp16", 3), time_job("bf16", 3)
            line["fp16_mode"] = {"precision": "fp16", "ms_per_job": round(ms_h, 3), "frames_per_s": round(B * 300 / ms_h * 1e3, 1),
                                 "eps_err": eps_err["fp16"], "bf16_ms_per_job": round(ms_b, 3), "bf16_eps_err": eps_err["bf16"],
                                 "what": "k_sample8h + k_vae_fusedh: fp16 operands, fp32 accumulate / softmax / LayerNorm / scheduler; "
                                         "same job and error measure as parity_mode"}
            # the sampler the reference ships (infer_ldm.py:116-125): DDIM-50, same clips
            eng.set_schedule(sch.ddim_table())
            dd = {mode: time_job(mode, 5) for mode in ("bf16", "fp16", "fp32x")}
            line["ddim50"] = {"clips": B, **{f"{m}_ms_per_job": round(v, 3) for m, v in dd.items()},
                              **{f"{m}_frames_per_s": round(B * 300 / v * 1e3, 1) for m, v in dd.items()}}
            # ... and as the reference runs it: ONE clip per call (infer_gesture on one 10 s WAV, BASELINE configs[0] / [1] with DDIM-50)
            eng.set_clips_per_group(0)
            for mode in ("bf16", "fp32x"):
                ts1 = []
                for i in range(25):
                    ev0.record()
                    eng.diffusion_backward(c1, e1, s1, mode, seed=2024, out=o1)
                    ev1.record()
                    ev1.synchronize()
                    if i >= 5:
                        ts1.append(ev0.elapsed_time(ev1))
                line["ddim50"][f"{mode}_single_clip_ms"] = round(statistics.median(ts1), 3)
            eng.set_schedule(sch.ddpm_table(args.T))
            eng.set_clips_per_group(0)
        if world == 1 and args.precision in ("bf16", "fp16"):
            try:   # the headline must not depend on an extra
                line["diffusion_only"] = diffusion_only_extra(dev, args.precision, peak)
            except Exception as e:
                line["diffusion_only"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_audio:
            # side measurement, not part of `value` (whose inputs are the three 256-d embeddings): the
            # audio front-end that produces them from 10 s of 16 kHz audio - kaldi fbank + 3 x AST, 778 GFLOP per clip
            from amuse_amd import audio_weights as aw
            from amuse_amd.audio import AudioEngine
            aeng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS), device=dev)
            Ba = 32
            wav = 0.1 * torch.randn(Ba, 160000, generator=gen).to(dev)
            ta = []
            for i in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                aeng.features(wav)
                torch.cuda.synchronize()
                if i >= 1:
                    ta.append(time.perf_counter() - t1)
            flop = 3 * 12 * (2 * 1214 * 768 * (2304 + 768 + 2 * 3072) + 4 * 1214 * 1214 * 768)
            ms_clip = min(ta) * 1e3 / Ba
            line["audio_frontend"] = {"clips": Ba, "ms_per_clip": round(ms_clip, 3),
                                      "tflops": round(flop / (ms_clip * 1e-3) / 1e12, 1),
                                      "frames_per_s_wav_to_smplx": round(300.0 / (ms_clip * 1e-3 + elapsed / args.steps / total), 1)}
            aeng.close()
            del aeng, wav
        if world == 1:
            # BASELINE config 4 beside the headline: the train_gesture iteration at batch 32 on this GPU - `--config train` in a child process of its own
            # (the step is host-bound: inside this process, behind the engines above, it measures ~25 % slower than alone); DESIGN.md 4.6
            import subprocess
            r = None
            try:
                r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--config", "train", "--steps", "40", "--warmup", "15"], capture_output=True, text=True,
                                   timeout=600)
                tl = json.loads(r.stdout.strip().splitlines()[-1])
                line["train_gesture"] = {"batch_per_gpu": 32, "it_per_s": tl["value"], "ms_per_iteration": tl["ms_per_step"], "how": "python bench.py --config train --steps 40 "
                                         "--warmup 15 as a child process; transformer layers (incl. their fp32 attention) on the library's layer-level entry points (amuse_train_*), DESIGN.md 4.6"}
            except Exception as e:   # the headline must not depend on the extra
                line["train_gesture"] = {"error": f"{type(e).__name__}: {e}; stderr tail: {r.stderr[-300:] if r is not None else ''}"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(total, args.T, wd, wp)
        if world == 1 and not args.no_torch_baseline and args.T == 1000:
            # the reference's own FORM of this job on this GPU: PyTorch-ROCm eager modules in a Python denoising loop (tools/gpu_torch_eager_baseline.py - the build's torch
            # twins of the reference modules on torch's stock layers; the reference itself cannot travel to the GPU box).  ~2 s per job + a 20-step warm-up.
            try:
                sys.path.insert(0, str(REPO / "tools"))
                from gpu_torch_eager_baseline import WHAT, measure
                tb = measure(total, dev, samplers=("DDPM-1000",), precisions=("fp32",), repeats=1)[0]
                line["torch_eager_baseline"] = dict(tb, what=WHAT, speedup_of_value=round(line["value"] / tb["frames_per_s"], 1))
            except Exception as e:   # the headline must not depend on the extra
                line["torch_eager_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if line is not None:
        # every headline carries its parity-grade twin at the TOP level (the driver's parsed block keeps top-level scalars and `config`):
        # `value` is the configuration BASELINE.json names (bf16 operands); the mode that meets the north star's "< 1e-4" is fp32x
        pm, att = line.get("parity_mode", {}), line.get("decode", {}).get("attention", {})
        twin = {"headline_precision": args.precision, "headline_eps_err_vs_reference": pm.get("bf16_eps_err") if args.precision == "bf16" else None,
                "parity_precision": pm.get("precision"), "parity_frames_per_s": pm.get("frames_per_s"), "parity_ms_per_job": pm.get("ms_per_job"),
                "parity_eps_err_vs_reference": pm.get("eps_err"), "parity_pose_l2_max_vs_reference": pm.get("pose_l2_max"),
                "attention_frac_of_mfma_peak": att.get("frac_of_mfma_peak"),
                "vs_torch_eager_same_gpu": line.get("torch_eager_baseline", {}).get("speedup_of_value")}
        line.update({k: v for k, v in twin.items()})
        line["config"]["parity_twin"] = twin
    barrier()
    if world > 1:
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
