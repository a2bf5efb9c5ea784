#!/bin/bash
# builds tests/host_asan/host_asan (the library's host code + stubbed runtime, -fsanitize=address,undefined): build.sh <out dir>
set -e
here="$(cd "$(dirname "$0")" && pwd)"
out=${1:-/tmp/amuse_host_asan}
mkdir -p "$out"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
for f in amuse_api amuse_variants amuse_audio_api; do
  $HIPCC --offload-host-only -std=c++17 $SAN -Wno-unused-function -c "$here/../../amuse_amd/csrc/$f.hip" -o "$out/$f.o"
done
$HIPCC --offload-host-only -std=c++17 $SAN -c "$here/hip_stub.cpp" -x hip -o "$out/hip_stub.o" 2>/dev/null || $HIPCC --offload-host-only -std=c++17 $SAN -x hip -c "$here/hip_stub.cpp" -o "$out/hip_stub.o"
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 $SAN -c "$here/main.cpp" -o "$out/main.o"
/opt/rocm/lib/llvm/bin/clang++ $SAN "$out/main.o" "$out/hip_stub.o" "$out/amuse_api.o" "$out/amuse_variants.o" "$out/amuse_audio_api.o" -o "$out/host_asan"
# the launch-plan sweep (tests/test_plan_cpu.py) on the same objects
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 $SAN -c "$here/plan_sweep.cpp" -o "$out/plan_sweep.o"
/opt/rocm/lib/llvm/bin/clang++ $SAN "$out/plan_sweep.o" "$out/hip_stub.o" "$out/amuse_api.o" "$out/amuse_variants.o" "$out/amuse_audio_api.o" -o "$out/plan_sweep"
