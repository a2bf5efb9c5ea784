// Argument blocks + launchers of the audio front-end kernels (k_audio.hip), used by amuse_audio_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amuse {

constexpr int kAstDim = 768, kAstHeads = 12, kAstLayers = 12, kAstMlp = 3072, kAstFeat = 256;
constexpr int kAstMel = 128, kAstFrames = 1024;
constexpr int kAstF = 12, kAstT = 101, kAstPatches = kAstF * kAstT, kAstTokens = 2 + kAstPatches;   // 1214
constexpr int kAstPoolSplit = 16;    // workgroups per clip in k_ast_pool (row slices, added in slice order by k_ast_head)
constexpr int kAstRows = 1216;      // row stride of a clip in every activation matrix: 1214 tokens + 2 pad rows = 76 tiles of 16 = 19 of 64,
                                    // so no 16-row tile, no 64-row wave span and no 64-key attention chunk straddles two clips
constexpr int kAstKeysPad = kAstRows;

// ---- Tile-major activation layout.  Every matrix the GEMM reads or writes as an operand is stored as 16-row x 32-feature
// tiles, row tiles outer: tile (row / 16, f / 32) of an [M][F] matrix is 512 consecutive elements.
//   bf16: [64 lanes][8]: lane (g, j) = row j of the tile, features 8 g .. 8 g + 7 - exactly one MFMA B-operand fragment, so the
//         GEMM's LDS-DMA copies tiles verbatim (1 KiB contiguous per instruction, full cache lines) and its epilogue - a lane
//         holds 8 consecutive features of a row - stores a whole tile per wave instruction.
//   fp32: two halves [features 8 g + 0..3 | 8 g + 4..7] of [64 lanes][4] floats: each of a lane's two 16-byte accesses belongs to a
//         wave-contiguous 1 KiB.
// Rows are padded to kGemmTM; pad rows hold finite garbage that never reaches a valid row (every row is computed on its own).
constexpr int kGemmTM = 128, kGemmTN = 256;   // output tile of k_gemm_tm: tokens x features
__host__ __device__ inline size_t tm_bf16(size_t row, int f, int F) {
    return ((row >> 4) * (size_t)(F >> 5) + (f >> 5)) * 512 + ((((f & 31) >> 3) << 4) + (row & 15)) * 8 + (f & 7);
}
__host__ __device__ inline size_t tm_f32(size_t row, int f, int F) {
    return ((row >> 4) * (size_t)(F >> 5) + (f >> 5)) * 512 + ((f >> 2) & 1) * 256 + ((((f & 31) >> 3) << 4) + (row & 15)) * 4 + (f & 3);
}

enum { EPI_BF16 = 0, EPI_GELU_BF16, EPI_RESID_F32, EPI_F32, EPI_PATCH, EPI_QKV };
struct GemmArgs {
    const unsigned short* A;     // bf16 tile-major [M padded to 128][K]
    const unsigned short* W;     // bf16, PACKED in MFMA-fragment order (amuse_audio_api.hip pack_w): for every 64-feature
                                 // span, fragment x = 2 p + q (row i <-> feature 32 p + 8 (i >> 2) + 4 q + (i & 3), so that
                                 // a lane's accumulators of a fragment pair are 8 consecutive features), k-step ks (32 k):
                                 // unit [64 lanes][8 bf16], lane (g, i) = W[feature][32 ks + 8 g + e]
    const float* bias;           // [N]
    int M, N, K;                 // N % 256 == 0, K % 64 == 0
    unsigned short* out_bf16;    // EPI_BF16 / EPI_GELU_BF16: tile-major [M padded][N]; EPI_QKV: q (pre-scaled) | k tile-major [M padded][1536]
    float* out_f32;              // EPI_RESID_F32 (+=) / EPI_F32: tile-major [M padded][N]; EPI_PATCH: tile-major token matrix [B * 1216 padded][768]
    const float* pos;            // EPI_PATCH: pos_embed [1214][768]
    unsigned short* vt;          // EPI_QKV: V^T as a tile-major matrix [B * 768 rows][1216 key slots] (k_ast_attn has the row / slot order)
};
hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s);
hipError_t launch_fbank(const float* wave, int n_samples, int B, const float* window, const float* melw_t, const int* mel_range, float mean, float std,
                        float* out, hipStream_t s);
hipError_t launch_im2col(const float* fbank, unsigned short* patches, int B, hipStream_t s);
hipError_t launch_ast_tokens(const float* cls, const float* dist, const float* pos, float* X, int B, hipStream_t s);
hipError_t launch_ln_bf16(const float* X, const float* gamma, const float* beta, float eps, unsigned short* out, int M, hipStream_t s);
// row-major <-> tile-major copies (hidden-state tap, amuse_debug_gemm): rows >= M of a tile-major destination are zeroed
hipError_t launch_tile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s);
hipError_t launch_untile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s);
hipError_t launch_untile_f32(const float* src, float* dst, int M, int F, int rows_in, int rows_out, hipStream_t s);
hipError_t launch_ast_attn(const unsigned short* QK, const unsigned short* Vt, unsigned short* O, int B, hipStream_t s);
hipError_t launch_ast_pool(const float* X, const float* gamma, const float* beta, int frame_based, float* pooled, int B, hipStream_t s);
hipError_t launch_ast_head(const float* pooled, int frame_based, const float* gamma, const float* beta, const unsigned short* W, const float* bias,
                           float* out, int B, hipStream_t s);

}  // namespace amuse
