#!/bin/bash
# build libamuse_hip variants with extra -D flags for A/B timing: tools/build_variant.sh NAME "-DX=1 ..."
set -e
cd "$(dirname "$0")/../amuse_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $@ -c k_sampler8.hip -o /tmp/k_sampler8_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libamuse_hip_$name.so amuse_api.o amuse_audio_api.o k_sampler.o /tmp/k_sampler8_$name.o k_vae.o k_misc.o k_audio.o
