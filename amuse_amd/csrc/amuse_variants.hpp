// The Denoiser variants behind the C ABI (include/amuse_hip.h AMUSE_ARCH_*): what amuse_api.hip forwards to amuse_variants.hip when
// a context was created with amuse_create_arch(arch != AMUSE_ARCH_ENC).
#pragma once
#include "amuse_host.hpp"

size_t variant_param_count(int arch);   // floats of the variant's state dict, 0 = unknown arch
inline size_t variant_state_dim(int arch) { return (arch & 2) ? (size_t)AMUSE_POSE_STATE : (size_t)AMUSE_D_MODEL; }
// packs and uploads the variant's weight streams, parameter vectors and projection tables (first call allocates, later calls overwrite)
int variant_build(amuse_ctx* c, const float* den, int what);
void variant_destroy(amuse_ctx* c);
// after amuse_set_schedule has put timesteps / coefficients on the device: the per-step time tokens (and their K / V tables)
int variant_set_schedule(amuse_ctx* c, hipStream_t st);
int variant_sample(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int precision, uint64_t seed,
                   uint64_t clip0, const float* x_init, const float* step_noise, float* out, float* traj_out, hipStream_t st);
// teacher-forced step: timesteps host [per_clip ? B : 1]; lengths host [B] or null (pose-space variants only)
int variant_denoise(amuse_ctx* c, const float* x_t, const int* timesteps, bool per_clip, const float* con, const float* emo,
                    const float* sty, const int* lengths, int B, int precision, float* eps_out, float* tap_out, hipStream_t st);
