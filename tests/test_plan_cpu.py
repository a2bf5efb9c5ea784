"""CPU: ONE source of truth for the launch plan.  `amuse_plan` (include/amuse_hip.h; csrc/amuse_host.hpp plan_*) is the only statement of which kernels a job
takes - amuse_amd/shard.py and the C client call it.  tests/host_asan/plan_sweep.cpp runs the library's host code on the stubbed HIP runtime (no GPU) and, for EVERY
clip count 1..8192 x four precisions x 3 / 4 / 5 tokens, checks that what amuse_sample / amuse_vae_decode / amuse_vae_encode / a pose-space Denoiser step actually took
on AUTO (amuse_debug_last_plan) is what amuse_plan returned, and that a job's pinned plan overrides a shard's own clip count."""
import os
import subprocess


def test_auto_takes_exactly_what_amuse_plan_returns(host_asan_build):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([str(host_asan_build / "plan_sweep")], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0 and "PLAN SWEEP OK" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
    assert int(run.stdout.split("(")[1].split()[0]) > 160000
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr


def test_python_has_no_restatement_of_the_plan():
    """shard.job_plan is a call into the library; the names of the round-5 restatements are gone from the package."""
    import inspect
    from amuse_amd import _lib, shard
    for name in ("job_clips_per_group", "FUSED_DECODE_MIN_CLIPS", "fusedx_rule", "job_decode_path"):
        assert not hasattr(shard, name), name
    assert "_lib.plan" in inspect.getsource(shard.job_plan)
    for n in (1, 63, 64, 159, 160, 256, 257, 420, 4096):
        for prec in (_lib.PREC_F32, _lib.PREC_BF16, _lib.PREC_F32X, _lib.PREC_F16):
            assert shard.job_plan(n, 5, prec) == _lib.plan(n, prec, 5)
