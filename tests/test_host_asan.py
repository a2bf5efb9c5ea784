"""CPU: the library's host C++ (amuse_api.hip + amuse_audio_api.hip: context construction, MFMA-fragment weight packing, in-place
re-packing, workspaces, argument checks, teardown) under AddressSanitizer + UndefinedBehaviorSanitizer.  The HIP runtime and the
kernel launchers are stubbed with host memory / no-ops (tests/host_asan/hip_stub.cpp), so it runs without a GPU; the driver
(tests/host_asan/main.cpp) goes through the C ABI only and ends with a leak check of the "device" allocations."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent / "host_asan"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_host_code_is_clean_under_asan_and_ubsan(host_asan_build):
    tmp_path = host_asan_build
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([str(tmp_path / "host_asan"), "1"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert run.returncode == 0 and "HOST ASAN OK (audio 1)" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
