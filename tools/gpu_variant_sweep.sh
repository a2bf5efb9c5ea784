#!/bin/bash
# time every libamuse_hip_*.so variant: sampling time at 256 clips + phase sums of wave 0 / wave 4
for lib in amuse_amd/libamuse_hip*.so; do
  echo "== $lib"
  AMUSE_HIP_LIB=$PWD/$lib timeout 120 python tools/gpu_sample_smoke.py 2>&1 | grep "B=256 bf16"
  AMUSE_HIP_LIB=$PWD/$lib timeout 60 python tools/gpu_phase_profile8.py 2>&1 | grep "^wave [04]"
done
