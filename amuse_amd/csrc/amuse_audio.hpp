// Argument blocks + launchers of the audio front-end kernels (k_audio.hip), used by amuse_audio_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amuse {

constexpr int kAstDim = 768, kAstHeads = 12, kAstLayers = 12, kAstMlp = 3072, kAstFeat = 256;
constexpr int kAstMel = 128, kAstFrames = 1024;
constexpr int kAstF = 12, kAstT = 101, kAstPatches = kAstF * kAstT, kAstTokens = 2 + kAstPatches;   // 1214
constexpr int kAstPoolSplit = 16;    // workgroups per clip in k_ast_pool (row slices, added in slice order by k_ast_head)
constexpr int kAstKeysPad = 1216;   // V^T rows padded to whole 64-key chunks (the pad columns stay zero)

enum { EPI_BF16 = 0, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32, EPI_PATCH, EPI_QKV };
struct GemmArgs {
    const unsigned short* A;     // bf16 [M padded to 128][K]
    const unsigned short* W;     // bf16, PACKED in MFMA-fragment order (amuse_audio_api.hip pack_w): for every 64-feature
                                 // span, fragment x = 2 p + q (row i <-> feature 32 p + 8 (i >> 2) + 4 q + (i & 3), so that
                                 // a lane's accumulators of a fragment pair are 8 consecutive features), k-step ks (32 k):
                                 // unit [64 lanes][8 bf16], lane (g, i) = W[feature][32 ks + 8 g + e]
    const float* bias;           // [N]
    int M, N, K;                 // N % 128 == 0, K % 64 == 0
    unsigned short* out_bf16;    // EPI_BF16 / EPI_GELU_BF16: [M][N]; EPI_QKV: q | k as [M][1536]
    float* out_f32;              // EPI_RESID_F32 (+=) / EPI_F32: [M][N]; EPI_PATCH: token matrix [B * 1214][768]
    const float* pos;            // EPI_PATCH: pos_embed [1214][768]
    unsigned short* vt;          // EPI_QKV: V^T [B][768][kAstKeysPad]
};
hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
hipError_t launch_fbank(const float* wave, int n_samples, int B, const float* window, const float* melw_t, const int* mel_range, float mean, float std,
                        float* out, hipStream_t s);
hipError_t launch_im2col(const float* fbank, unsigned short* patches, int B, hipStream_t s);
hipError_t launch_ast_tokens(const float* cls, const float* dist, const float* pos, float* X, int B, hipStream_t s);
hipError_t launch_ln_bf16(const float* X, const float* gamma, const float* beta, float eps, unsigned short* out, int M, hipStream_t s);
hipError_t launch_ast_attn(const unsigned short* QK, const unsigned short* Vt, unsigned short* O, int B, hipStream_t s);
hipError_t launch_ast_pool(const float* X, const float* gamma, const float* beta, int frame_based, float* pooled, int B, hipStream_t s);
hipError_t launch_ast_head(const float* pooled, int frame_based, const float* gamma, const float* beta, const unsigned short* W, const float* bias,
                           float* out, int B, hipStream_t s);

}  // namespace amuse
