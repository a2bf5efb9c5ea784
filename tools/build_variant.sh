#!/bin/bash
# build libamuse_hip variants with extra -D flags for A/B timing: tools/build_variant.sh NAME SOURCE.hip "-DX=1 ..."
# (SOURCE = the one translation unit that is recompiled, e.g. k_sampler8.hip or k_vae_fused.hip; the other objects are
# taken from the regular build, which must be up to date)
set -e
cd "$(dirname "$0")/../amuse_amd/csrc"
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $@ -c $src -o /tmp/${src%.hip}_$name.o
objs=""
for o in amuse_api amuse_variants amuse_audio_api k_sampler k_sampler_dec k_sampler8 k_sampler8h k_sampler8x k_vae k_vae_rows8 k_vae_fused k_vae_fusedh k_vae_fusedx k_den_fused k_den_fusedh k_misc k_train k_train_attn k_train_gemm k_audio k_audio_gemm; do
  if [ "$o.hip" == "$src" ]; then objs="$objs /tmp/${o}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libamuse_hip_$name.so $objs
