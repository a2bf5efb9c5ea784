#!/bin/bash
# build.sh: the probe in the filler placements that were measured (B1_V0 = VALU instructions behind the first score MFMA of a tile)
set -e
cd "$(dirname "$0")"
for tau in 0 6; do for v in 12; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 -DB1_V0=$v -DTAU=$tau attend32_probe.hip -o attend32_v${v}_t${tau}_probe
done; done
