// Audio front-end: PretrainedLPDM_v1.process_single_seq (reference
// models/latent_diffusion/infer_ldm.py:180-193) = kaldi fbank -> pad / normalise -> 3 x ASTModel.forward
// (models/audio/audio_main_new.py:174-204: DeiT-B distilled ViT over 2 + 12 x 101 tokens) -> feature_head.
// 260 GFLOP per encoder per clip - 37 x the 1000-step sampler - almost all of it in four GEMM shapes, so this file is
// a conventional MFMA pipeline rather than the register-resident design of the sampler:
//   k_fbank        one workgroup per frame: DC removal, pre-emphasis, Hann window, 512-point FFT in LDS, 128 mel bins
//   k_im2col       16 x 16 stride-10 patches of the [128 x 1024] spectrogram as bf16 rows (K = 256)
//   k_gemm_tm      (k_audio_gemm.hip) C = A . W^T (+ fused epilogue): 256 x 128 x 32 stages by LDS-DMA, fp32 accumulation.
//                  Weights are the A operand, so a lane ends up with 8 consecutive FEATURES of one token row: bias,
//                  GELU, residual and the q / k / v^T split are applied in registers.  Its operands and outputs are
//                  TILE-MAJOR (amuse_audio.hpp): so are the residual stream and every activation in this file.
//   k_ln_bf16      LayerNorm of the fp32 residual stream -> bf16 GEMM operand (one workgroup per 16-row tile row)
//   k_ast_attn     flash attention, S = 1214, d = 64: S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_16x16x32_bf16 with
//                  the softmax along registers (the layout of the S = 300 decoder attention, k_vae.hip), K and V^T tiles
//                  streamed through LDS by DMA in 64-key chunks; V is written TRANSPOSED by the qkv epilogue.
// Row space: a clip owns kAstRows = 1216 rows of every activation matrix (1214 tokens + 2 pad rows).
//   k_ast_pool / k_ast_head   final LayerNorm + mean over the patch tokens, feature_head (LayerNorm + Linear 768 -> 256)
// Arithmetic: bf16 GEMM / attention operands, fp32 accumulation, fp32 residual stream, LayerNorm, softmax and GELU.
#include <cstdlib>

#include "amuse_dev.hpp"
#include "amuse_audio.hpp"

namespace amuse {
namespace {

typedef unsigned short bf16raw;
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ uint2 pack4(f32x4 v) { return uint2{pack2(v[0], v[1]), pack2(v[2], v[3])}; }
// LDS-DMA: 64 lanes x 16 B from (wave-uniform base + 32-bit lane offset) to LDS [dst, dst + 1 KiB), lane-linear (k_audio_gemm.hip)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// ---------------------------------------------------------------------------------------------- fbank
// grid (kAstFrames, B), 256 threads.  Frames beyond the waveform are the padding rows of infer_ldm.py:185-188.
__global__ __launch_bounds__(256) void k_fbank(const float* __restrict__ wave, int n_samples, const float* __restrict__ window,
                                               const float* __restrict__ melw_t /*[257][128]: transposed*/,
                                               const int* __restrict__ mel_range /*[128][2]: first bin, end bin of the filter*/,
                                               float norm_mean, float inv_2std,
                                               float* __restrict__ out /*[B][1024][128]*/) {
    __shared__ float re[512], im[512], red[8];
    const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    float* dst = out + ((size_t)b * kAstFrames + f) * kAstMel;
    const int n_frames = n_samples < 400 ? 0 : 1 + (n_samples - 400) / 160;
    if (f >= n_frames) {
        if (t < kAstMel) dst[t] = (0.0f - norm_mean) * inv_2std;
        return;
    }
    const float* src = wave + (size_t)b * n_samples + (size_t)f * 160;
    const float x0 = t < 400 ? src[t] : 0.f, x1 = t + 256 < 400 ? src[t + 256] : 0.f;
    float s = x0 + x1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / 400.0f);
    re[t] = x0 - mean;
    re[t + 256] = (t + 256 < 400) ? x1 - mean : 0.f;
    __syncthreads();
    // pre-emphasis against the previous sample (the first against itself), window, bit-reversed scatter for the FFT
    float y[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = t + 256 * h;
        y[h] = 0.f;
        if (i < 400) y[h] = (re[i] - 0.97f * re[i > 0 ? i - 1 : 0]) * window[i];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = t + 256 * h;
        const int r = __brev((unsigned)i) >> 23;   // 9-bit reversal
        re[r] = y[h];
        im[r] = 0.f;
    }
    __syncthreads();
    for (int len = 2; len <= 512; len <<= 1) {
        const int half = len >> 1, k = t & (half - 1), base = ((t - k) << 1) + k;
        float sn, cs;
        sincospif(-2.0f * (float)k / (float)len, &sn, &cs);
        const float ur = re[base], ui = im[base], vr = re[base + half], vi = im[base + half];
        const float tr = vr * cs - vi * sn, ti = vr * sn + vi * cs;
        __syncthreads();
        re[base] = ur + tr; im[base] = ui + ti;
        re[base + half] = ur - tr; im[base + half] = ui - ti;
        __syncthreads();
    }
    // power spectrum (bins 0..256) in place, then the mel filters
    const float p0 = re[t] * re[t] + im[t] * im[t];
    const float p256 = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    re[t] = p0;
    if (t == 0) re[256] = p256;
    __syncthreads();
    if (t < kAstMel) {
        // bins in ascending order, as the dense product; the filter's zero bins outside [k0, k1) add nothing.  The
        // transposed weights make a bin's 128 loads one contiguous 512 B row.
        const int k0 = mel_range[2 * t], k1 = mel_range[2 * t + 1];
        float e = 0.f;
        for (int k = k0; k < k1; ++k) e += re[k] * melw_t[k * kAstMel + t];
        dst[t] = (logf(fmaxf(e, 1.1920929e-07f)) - norm_mean) * inv_2std;
    }
}

// ---------------------------------------------------------------------------------------------- im2col
// patches[b * 1212 + fh * 101 + tw][kh * 16 + kw] = fbank[b][10 tw + kw][10 fh + kh]   (x.unsqueeze(1).transpose(2, 3)
// then Conv2d(1, 768, 16, stride 10): audio_main_new.py:180-184, 92-96); tile-major (amuse_audio.hpp)
__global__ __launch_bounds__(256) void k_im2col(const float* __restrict__ fbank, bf16raw* __restrict__ patches, int B) {
    const size_t row = (size_t)blockIdx.x;           // b * 1212 + p
    const int b = (int)(row / kAstPatches), p = (int)(row - (size_t)b * kAstPatches);
    const int fh = p / kAstT, tw = p - fh * kAstT;
    const int kh = threadIdx.x >> 4, kw = threadIdx.x & 15;
    const float v = fbank[((size_t)b * kAstFrames + 10 * tw + kw) * kAstMel + 10 * fh + kh];
    typedef __bf16 bf;
    patches[tm_bf16(row, threadIdx.x, 256)] = __builtin_bit_cast(bf16raw, (bf)v);
}

// cls / distillation rows of the token matrix (audio_main_new.py:185-188); the clip's two pad rows (1214, 1215) restart from zero with
// every call (they are read-modify-written by every block and would otherwise drift from call to call - finite, never reaching a valid
// row, but not reproducible)
__global__ __launch_bounds__(256) void k_ast_tokens(const float* __restrict__ cls, const float* __restrict__ dist,
                                                    const float* __restrict__ pos, float* __restrict__ X) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * kAstDim; i += 256) {
        const int r = i / kAstDim, c = i - r * kAstDim;
        X[tm_f32((size_t)b * kAstRows + r, c, kAstDim)] = (r == 0 ? cls[c] : dist[c]) + pos[(size_t)r * kAstDim + c];
        X[tm_f32((size_t)b * kAstRows + kAstTokens + r, c, kAstDim)] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm rows
// fp32 tile-major in -> bf16 tile-major out.  One workgroup per 16-row tile row; wave w owns the feature tiles 6 w .. 6 w + 5, so
// lane (g, j) holds 48 values of row j (features 32 t + 8 g .. + 7) and every load / store of a wave is one contiguous 1 KiB.
// Row statistics: lane partials -> the four g lanes of a row (two shuffles) -> the four waves through LDS, two passes (mean,
// then the centred second moment), always in the same order - a row's result does not depend on where in the batch it sits.
__global__ __launch_bounds__(256) void k_ln_bf16(const float* __restrict__ X, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, bf16raw* __restrict__ out) {
    __shared__ float red[2][4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t tile0 = (size_t)blockIdx.x * (kAstDim / 32) + 6 * wave;
    const float* x = X + tile0 * 512 + lane * 4;
    f32x4 v[6][2];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        v[t][0] = ld4(x + t * 512);
        v[t][1] = ld4(x + t * 512 + 256);
        s += ((v[t][0][0] + v[t][0][1]) + (v[t][0][2] + v[t][0][3])) + ((v[t][1][0] + v[t][1][1]) + (v[t][1][2] + v[t][1][3]));
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (g == 0) red[0][wave][j] = s;
    __syncthreads();
    const float mean = ((red[0][0][j] + red[0][1][j]) + (red[0][2][j] + red[0][3][j])) * (1.0f / kAstDim);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float d = v[t][h][m] - mean;
                q += d * d;
            }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (g == 0) red[1][wave][j] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0][j] + red[1][1][j]) + (red[1][2][j] + red[1][3][j])) * (1.0f / kAstDim) + eps);
    bf16raw* o = out + tile0 * 512 + lane * 8;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int f = 32 * (6 * wave + t) + 8 * g;
        uint2 pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 ga = ld4(gamma + f + 4 * h), be = ld4(beta + f + 4 * h);
            f32x4 y;
#pragma unroll
            for (int m = 0; m < 4; ++m) y[m] = (v[t][h][m] - mean) * rstd * ga[m] + be[m];
            pk[h] = pack4(y);
        }
        *reinterpret_cast<uint4*>(o + t * 512) = uint4{pk[0].x, pk[0].y, pk[1].x, pk[1].y};
    }
}

// ---------------------------------------------------------------------------------------------- row-major <-> tile-major
// one workgroup per 16-row tile row; thread = (tile-relative feature octet walk): rows >= M read as zero / are not written
__global__ __launch_bounds__(256) void k_tile_bf16(const bf16raw* __restrict__ src, bf16raw* __restrict__ dst, int M, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    for (int ft = wave; ft < F / 32; ft += 4) {
        uint4 u = uint4{0, 0, 0, 0};
        if (row < (size_t)M) u = *reinterpret_cast<const uint4*>(src + row * F + 32 * ft + 8 * g);
        *reinterpret_cast<uint4*>(dst + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 8) = u;
    }
}
__global__ __launch_bounds__(256) void k_untile_bf16(const bf16raw* __restrict__ src, bf16raw* __restrict__ dst, int M, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    if (row >= (size_t)M) return;
    for (int ft = wave; ft < F / 32; ft += 4)
        *reinterpret_cast<uint4*>(dst + row * F + 32 * ft + 8 * g) =
            *reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 8);
}
// rows_in / rows_out: row stride of a clip in the source / number of its rows that are copied (rows_in == rows_out: a flat matrix)
__global__ __launch_bounds__(256) void k_untile_f32(const float* __restrict__ src, float* __restrict__ dst, int M, int F, int rows_in, int rows_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const size_t row = (size_t)blockIdx.x * 16 + j;
    if (row >= (size_t)M) return;
    const size_t b = row / rows_in, r = row - b * rows_in;
    if (r >= (size_t)rows_out) return;
    float* d = dst + (b * rows_out + r) * F;
    for (int ft = wave; ft < F / 32; ft += 4) {
        const float* t = src + ((size_t)blockIdx.x * (F / 32) + ft) * 512 + lane * 4;
        st4(d + 32 * ft + 8 * g, ld4(t));
        st4(d + 32 * ft + 8 * g + 4, ld4(t + 256));
    }
}

// ---------------------------------------------------------------------------------------------- attention
// Flash attention over one clip's 1214 tokens, head_dim 64; every operand is a tile of a tile-major matrix (amuse_audio.hpp), i.e.
// a ready MFMA fragment that LDS-DMA copies verbatim:
//   Q, K   tiles of QK [B * 1216][1536] (q pre-scaled by the qkv epilogue): S^T = K . Q^T, lane (g, query j) holds S[j][key 16 u + 4 g + m]
//   V^T    tiles of Vt [B * 768][1216 key slots], written by the qkv epilogue's swapped MFMAs: rows in W-fragment order (row 16 F + i of
//          fragment F <-> feature 32 (F >> 1) + 8 (i >> 2) + 4 (F & 1) + (i & 3)), key slots in the order P comes out of the S^T MFMA
//          (key 16 a + 4 g + m of a group of 32 -> slot 8 g + 4 a + m): O^T = V^T . P^T needs no data movement at all, and a lane's
//          accumulators of a fragment pair are 8 consecutive features - O is stored one whole tile per wave instruction.
// Workgroup = 256 queries of one head (8 waves x 2 query tiles: a K / V^T fragment read feeds two MFMAs); keys in chunks of 64
// (8 K tiles + 8 V^T tiles = 16 KiB) through a ring of three LDS stages, two chunks in flight, one barrier per chunk.  One
// online-softmax step per chunk: the per-step fixed costs - two cross-lane reductions, the rescale of the accumulators, exp2 of the
// running-max shift - are paid once per 64 keys, log2(e) rides in the exp2 argument's fma, only the last chunk masks keys.
// max of three.  Built with -fno-honor-nans (Makefile): hipcc then drops the v_max_f32 x, x canonicalisation it otherwise puts in
// front of every fmaxf operand and fuses pairs into v_max3_f32.  NOT inline asm: the hazard recogniser does not see through asm, and
// a VALU read of an MFMA result needs software wait states - an asm v_max3_f32 right behind the score MFMAs read stale registers.
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
constexpr int kAttnQ = 256;                         // queries per workgroup: 8 waves x 2 query tiles (NQ = 1: 128 - a launch of ONE clip, where
                                                    // 60 workgroups leave most of the chip idle: twice the workgroups, the same bits per query tile)
constexpr int kAttnStage = 16 * 1024;
constexpr int kAttnLds = 3 * kAttnStage;            // 48 KiB; two workgroups per CU (registers: four waves per SIMD)
constexpr int kAttnChunks = kAstRows / 64;          // 19
template <int NQ>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ast_attn(const bf16raw* __restrict__ QK, const bf16raw* __restrict__ Vt,
                                                                                            bf16raw* __restrict__ O) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    constexpr int kRowTiles = kAstRows / 16, kQkTiles = 2 * kAstDim / 32, kSlotTiles = kAstRows / 32;   // 76, 48, 38
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const unsigned voff = lane * 16;
    // this wave's two DMA pieces of chunk c: waves 0..3 the K tiles of key tile 4 c + wave (k-steps 0, 1), waves 4..7 the V^T tiles of row
    // tile 4 h + wave - 4 (slot tiles 2 c, 2 c + 1); stage layout: K tile (u, s) at 2 u + s, V^T tile (td, pr) at 8 + 2 td + pr
    const bool kwave = wave < 4;
    const char* src = kwave ? reinterpret_cast<const char*>(QK) + (((size_t)b * kRowTiles + wave) * kQkTiles + kAstDim / 32 + 2 * h) * 1024
                            : reinterpret_cast<const char*>(Vt) + ((size_t)b * (kAstDim / 16) + 4 * h + wave - 4) * kSlotTiles * 1024;
    const size_t cstep = kwave ? (size_t)4 * kQkTiles * 1024 : (size_t)2 * 1024;
    auto fetch = [&](int c, int slot) {
        c = c < kAttnChunks ? c : kAttnChunks - 1;   // past the end: the last chunk again (lands in a free slot, never read)
        const unsigned d = lds0 + slot * kAttnStage + 2 * wave * 1024;
        glds16s(src + c * cstep, voff, d);
        glds16s(src + c * cstep + 1024, voff, d + 1024);
    };
    fetch(0, 0);
    fetch(1, 1);
    const int qt0 = 8 * NQ * qb + NQ * wave;        // this wave's query tiles qt0 (, qt0 + 1) of the clip's 76
    bf16x8 qf[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 u = uint4{0, 0, 0, 0};
            if (qt0 + q < kRowTiles)
                u = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(QK) + (((size_t)b * kRowTiles + qt0 + q) * kQkTiles + 2 * h + s) * 1024 + voff);
            qf[q][s] = __builtin_bit_cast(bf16x8, u);
        }
    // the q fragments are waited for HERE (they sit behind the first DMA pieces): an asm that redefines them makes hipcc put its
    // s_waitcnt in front of the loop instead of a vmcnt(0) in front of the first MFMA of every chunk, which would drain the DMA queue
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(qf[q][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // from here on only DMA is counted
    // Scores arrive in the log2 domain (the qkv epilogue scales q by log2(e) / 8) and RELATIVE to the row's running maximum: the
    // score MFMAs start from C = -m_run, so in the common case - the maximum did not move - p = exp2 of the MFMA result, no
    // subtraction, no rescale.  m_run is the true running maximum from chunk 0 on (chunk 0 starts from C = 0 and takes its own).
    // The row sums ride the matrix pipe (as in k_vae_fused.hip): a constant "V^T" fragment whose row d = 0 is all ones makes
    // O^T[0][j] = sum_key P[j][key] - two MFMAs per query tile and chunk instead of fifteen adds and a butterfly on a VALU that is
    // the busier pipe here (5.6 VALU instructions per MFMA).  What is summed is the bf16 P the PV product uses.  Lane (g = 0, j)
    // collects query j's sum in os[q][0]; the other three lanes of the row hold 0 there and ONE butterfly at the end broadcasts it.
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (lane & 15) == 0 ? uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u} : uint4{0u, 0u, 0u, 0u});
    float m_run[NQ];
    f32x4 os[NQ];
    f32x4 o[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        m_run[q] = 0.f;
        os[q] = splat4(0.f);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int td = 0; td < 4; ++td) o[q][td] = splat4(0.f);
    int slot = 0, fslot = 2;
#pragma unroll 1
    for (int c = 0; c < kAttnChunks; ++c) {
        // this wave's pieces of chunk c have landed (chunk c + 1 may still fly), its reads of chunk c - 1 are done; behind the
        // barrier chunk c is complete and the slot of chunk c - 1 is free for chunk c + 2
        if (c == 0) asm volatile("s_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fetch(c + 2, fslot);
        fslot = fslot == 2 ? 0 : fslot + 1;
        const char* sl = smem + slot * kAttnStage + lane * 16;
        slot = slot == 2 ? 0 : slot + 1;
        f32x4 st[NQ][4];
        f32x4 c0[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) c0[q] = splat4(-m_run[q]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(sl + (2 * u) * 1024), k1 = *reinterpret_cast<const bf16x8*>(sl + (2 * u + 1) * 1024);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                st[q][u] = mfma_bf16(k0, qf[q][0], c0[q]);
                st[q][u] = mfma_bf16(k1, qf[q][1], st[q][u]);
            }
        }
        // lane (g, query j): log2-domain S[j][key = 64 c + 16 u + 4 g + m] - m_run[j]
        if (c == kAttnChunks - 1) {   // keys 1214, 1215 are the clip's pad rows
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (64 * c + 16 * u + 4 * g + m >= kAstTokens) st[q][u][m] = -INFINITY;
        }
        bf16x8 pb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float mx = max3(max3(st[q][0][0], st[q][0][1], st[q][0][2]), max3(st[q][0][3], st[q][1][0], st[q][1][1]), max3(st[q][1][2], st[q][1][3], st[q][2][0]));
            mx = max3(mx, max3(st[q][2][1], st[q][2][2], st[q][2][3]), max3(st[q][3][0], st[q][3][1], st[q][3][2]));
            mx = fmaxf(mx, st[q][3][3]);   // this lane's 16 scores; every chunk holds a valid key
            // the running maximum moves in the first chunks and then hardly ever: the row's maximum (four lanes) is formed, and the
            // scores shifted and the accumulators rescaled, only when SOME lane of the wave holds a positive score (wave-uniform branch)
            if (c == 0 || __builtin_amdgcn_ballot_w64(mx > kAttnTau) != 0) {   // (lazy rescaling: amuse_dev.hpp kAttnTau)
                mx = allreduce_g_max(mx);   // the same in the four lanes of a row
                const float d = c == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) st[q][u] -= splat4(d);
                const float alpha = c == 0 ? 0.f : __builtin_amdgcn_exp2f(-d);   // (chunk 0: o = l = 0, and exp2(-d) may overflow)
                os[q] *= alpha;
#pragma unroll
                for (int td = 0; td < 4; ++td) o[q][td] *= alpha;
                m_run[q] += d;
            }
            f32x4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 4; ++m) p[u][m] = __builtin_amdgcn_exp2f(st[q][u][m]);   // masked keys: exp2(-inf) = 0
            // k-slots (g, e) of key group pr: e < 4 -> tile 2 pr key 4 g + e, else tile 2 pr + 1 key 4 g + e - 4: the V^T slot order
            pb[q][0] = pack_bf16(p[0], p[1]);
            pb[q][1] = pack_bf16(p[2], p[3]);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int td = 0; td < 4; ++td) {
                const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sl + (8 + 2 * td + pr) * 1024);
#pragma unroll
                for (int q = 0; q < NQ; ++q) o[q][td] = mfma_bf16(vf, pb[q][pr], o[q][td]);
            }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int q = 0; q < NQ; ++q) os[q] = mfma_bf16(ones, pb[q][pr], os[q]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus fetches must not outlive the workgroup's LDS
    // o[q][td][m] = O[query j][feature 64 h + 32 (td >> 1) + 8 g + 4 (td & 1) + m]: the pair td = 2 t, 2 t + 1 is this lane's slot of tile 2 h + t
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (qt0 + q >= kRowTiles) continue;
        const float inv = 1.0f / allreduce_g_sum(os[q][0]);
        char* dst = reinterpret_cast<char*>(O) + (((size_t)b * kRowTiles + qt0 + q) * (kAstDim / 32) + 2 * h) * 1024 + voff;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const uint2 lo = pack4(o[q][2 * tt] * inv), hi = pack4(o[q][2 * tt + 1] * inv);
            *reinterpret_cast<uint4*>(dst + tt * 1024) = uint4{lo.x, lo.y, hi.x, hi.y};
        }
    }
}

// ---------------------------------------------------------------------------------------------- pooling + head
// v.norm on every token, then the mean over the 1212 patch tokens (frame_based_feats) or (cls + dist) / 2
// grid (B), 256 threads = 4 waves striding over the rows; partial sums combined through LDS
// grid (B, kAstPoolSplit): workgroup y normalises and sums its slice of the rows; k_ast_head adds the slices in a fixed
// order (one workgroup per clip walking all 1212 rows serially left the pooling at 0.3 ms, whatever the batch).
__global__ __launch_bounds__(256) void k_ast_pool(const float* __restrict__ X, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int frame_based, float* __restrict__ pooled) {
    __shared__ float part[4][kAstDim];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x, y = blockIdx.y;
    const int r0 = frame_based ? 2 : 0, r1 = frame_based ? kAstTokens : 2;
    const int chunk = (r1 - r0 + kAstPoolSplit - 1) / kAstPoolSplit;
    const int c0 = r0 + y * chunk, c1 = min(r1, c0 + chunk);
    f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
    f32x4 ga[3], be[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ga[i] = ld4(gamma + 256 * i + 4 * lane); be[i] = ld4(beta + 256 * i + 4 * lane); }
    for (int r = c0 + wave; r < c1; r += 4) {
        const size_t row = (size_t)b * kAstRows + r;
        f32x4 v[3];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[i] = ld4(X + tm_f32(row, 256 * i + 4 * lane, kAstDim)); s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / kAstDim);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) { const float d = v[i][m] - mean; q += d * d; }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * (1.0f / kAstDim) + 1e-6f);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[i][m] += (v[i][m] - mean) * rstd * ga[i][m] + be[i][m];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) st4(&part[wave][256 * i + 4 * lane], acc[i]);
    __syncthreads();
    for (int c = threadIdx.x; c < kAstDim; c += 256)
        pooled[((size_t)b * kAstPoolSplit + y) * kAstDim + c] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// feature_head: LayerNorm(768, eps 1e-5) -> Linear(768 -> 256) with bf16-rounded operands, fp32 accumulation
__global__ __launch_bounds__(256) void k_ast_head(const float* __restrict__ pooled /*[B][kAstPoolSplit][768] row sums*/,
                                                  float inv_rows, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const bf16raw* __restrict__ W /*[256][768]*/,
                                                  const float* __restrict__ bias, float* __restrict__ out /*[B][256]*/) {
    __shared__ float h[kAstDim];
    __shared__ float red[2][4];
    const int t = threadIdx.x, b = blockIdx.x;
    const float* x = pooled + (size_t)b * kAstPoolSplit * kAstDim;
    float v[3], s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // mean over the pooled rows: the slices of k_ast_pool, added in slice order
        float a = 0.f;
        for (int y = 0; y < kAstPoolSplit; ++y) a += x[y * kAstDim + t + 256 * i];
        v[i] = a * inv_rows;
        s += v[i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((t & 63) == 0) red[0][t >> 6] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / kAstDim);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = v[i] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((t & 63) == 0) red[1][t >> 6] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / kAstDim) + 1e-5f);
    typedef __bf16 bf;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = t + 256 * i;
        h[c] = (float)(bf)((v[i] - mean) * rstd * gamma[c] + beta[c]);   // GEMM operand rounding
    }
    __syncthreads();
    const bf16raw* w = W + (size_t)t * kAstDim;
    float acc = 0.f;
    for (int c = 0; c < kAstDim; c += 8) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(w + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)wv[e] * h[c + e];
    }
    out[(size_t)b * kAstFeat + t] = acc + bias[t];
}

}  // namespace

hipError_t launch_fbank(const float* wave, int n_samples, int B, const float* window, const float* melw, const int* mel_range, float mean, float std,
                        float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_fbank, dim3(kAstFrames, B), dim3(256), 0, s, wave, n_samples, window, melw, mel_range, mean, 1.0f / (2.0f * std), out);
    return hipGetLastError();
}
hipError_t launch_im2col(const float* fbank, unsigned short* patches, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_im2col, dim3(B * kAstPatches), dim3(256), 0, s, fbank, patches, B);
    return hipGetLastError();
}
hipError_t launch_ast_tokens(const float* cls, const float* dist, const float* pos, float* X, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ast_tokens, dim3(B), dim3(256), 0, s, cls, dist, pos, X);
    return hipGetLastError();
}
hipError_t launch_ln_bf16(const float* X, const float* gamma, const float* beta, float eps, unsigned short* out, int M, hipStream_t s) {
    hipLaunchKernelGGL(k_ln_bf16, dim3((M + 15) / 16), dim3(256), 0, s, X, gamma, beta, eps, out);
    return hipGetLastError();
}
hipError_t launch_tile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_bf16, dim3((M + kGemmTM - 1) / kGemmTM * (kGemmTM / 16)), dim3(256), 0, s, src, dst, M, F);
    return hipGetLastError();
}
hipError_t launch_untile_bf16(const unsigned short* src, unsigned short* dst, int M, int F, hipStream_t s) {
    hipLaunchKernelGGL(k_untile_bf16, dim3((M + 15) / 16), dim3(256), 0, s, src, dst, M, F);
    return hipGetLastError();
}
hipError_t launch_untile_f32(const float* src, float* dst, int M, int F, int rows_in, int rows_out, hipStream_t s) {
    hipLaunchKernelGGL(k_untile_f32, dim3((M + 15) / 16), dim3(256), 0, s, src, dst, M, F, rows_in, rows_out);
    return hipGetLastError();
}
hipError_t launch_ast_attn(const unsigned short* QK, const unsigned short* Vt, unsigned short* O, int B, hipStream_t s) {
    // one query tile per wave for a single clip (120 instead of 60 workgroups: 2.06 -> 2.02 ms for the whole front-end; at two clips already slower: profiles/r04_audio_attn_nq_ab.txt)
    const int nq = B == 1 ? 1 : 2;
    if (nq == 1) hipLaunchKernelGGL(k_ast_attn<1>, dim3((kAstRows + kAttnQ / 2 - 1) / (kAttnQ / 2), kAstHeads, B), dim3(512), kAttnLds, s, QK, Vt, O);
    else hipLaunchKernelGGL(k_ast_attn<2>, dim3((kAstRows + kAttnQ - 1) / kAttnQ, kAstHeads, B), dim3(512), kAttnLds, s, QK, Vt, O);
    return hipGetLastError();
}
hipError_t launch_ast_pool(const float* X, const float* gamma, const float* beta, int frame_based, float* pooled, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ast_pool, dim3(B, kAstPoolSplit), dim3(256), 0, s, X, gamma, beta, frame_based, pooled);
    return hipGetLastError();
}
hipError_t launch_ast_head(const float* pooled, int frame_based, const float* gamma, const float* beta, const unsigned short* W,
                           const float* bias, float* out, int B, hipStream_t s) {
    const float inv_rows = 1.0f / (float)(frame_based ? kAstTokens - 2 : 2);
    hipLaunchKernelGGL(k_ast_head, dim3(B), dim3(256), 0, s, pooled, inv_rows, gamma, beta, W, bias, out);
    return hipGetLastError();
}

}  // namespace amuse
