#!/bin/bash
# devasm.sh <csrc dir> <out dir>: device-only gfx950 ISA of every translation unit (for bitwise comparisons of refactors)
src=$1; out=$2; mkdir -p $out
cd $src
for f in k_sampler k_sampler_dec k_sampler8 k_sampler8h k_sampler8x k_vae k_vae_rows8 k_vae_fused k_vae_fusedh k_vae_fusedx k_den_fused k_den_fusedh k_misc k_train k_train_attn k_train_gemm k_audio k_audio_gemm; do
  extra=""
  case $f in k_audio) extra="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans";; k_vae_fused|k_vae_fusedh|k_vae_fusedx|k_den_fused|k_den_fusedh) extra="-fno-honor-nans";; esac
  echo "/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -S $extra $f.hip -o $out/$f.s 2>/dev/null"
done | xargs -P 8 -I{} bash -c "{}"
cd $out && for f in *.s; do grep -v "^\s*\.\(file\|ident\|section\|loc\)\|^\s*;" $f | sed 's/\s*;.*$//' | sha256sum | cut -c1-16 | tr '\n' ' '; echo $f; done
