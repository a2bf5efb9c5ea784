// The training step's TALL fp32 GEMMs (config 4: ldm.py:59-116 forward + autograd backward; every projection of the 27 transformer layers over 9,600 rows):
//     out[M][N] (+)= a[M][K] . b^T      b row-major [N][K]: forward projections  (x W^T, torch.nn.functional.linear)
//     out[M][N] (+)= a[M][K] . b        b row-major [K][N]: input gradients      (dy W)
// with M ~ 9,600, K and N in {128, 256, 384, 512}.  rocBLAS runs these at 12-30 us each, 2-3 x their MFMA time (a 64 x 64 macro tile with a four-step k loop is
// mostly prologue), and a first own kernel that read both operands straight from global memory as MFMA fragments was slower still (half-used cache lines, W
// re-read per wave: profiles/r04_train_wgrad_kernel_ab.txt).  Here both operands go through LDS in full 128-byte lines:
//   * workgroup = 48 (or 32) rows x 128 columns, four MFMA waves side by side along N: a wave owns all row tiles x 32 columns = MT x 2 accumulator tiles of
//     v_mfma_f32_16x16x4_f32 (fp32 products, fp32 accumulate: the arithmetic class of the library GEMM it replaces);
//   * the k loop runs in chunks of 32: a chunk of a (MT x 16 rows x 128 B) and of b (16 KiB) is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, one
//     1 KiB piece = 64 lanes x 16 B per instruction) into a ring of three buffers by four waves that do nothing else, two chunks ahead of the MFMAs;
//   * the LDS images are lane-linear per piece (all LDS-DMA can write), so the bank-conflict-free order is put into the SOURCE addresses: a 16-byte slot of an
//     image row sits at slot ^ f(row) (f = (row >> 1) & 7 for the 128-byte rows of a and of b^T; bit 3 of the slot ^ bit 2 of k for the 512-byte rows of b);
//   * k order inside a 16-wide step: lane (g, r) reads ONE 16-byte slot = k {16 j + 4 g + s}, s = 0..3, of its row and feeds MFMA s with element s - the k index
//     of an MFMA is {16 j + 4 g + s : g}, the same on both operands, so a fragment costs one ds_read_b128 per four MFMA k-steps;
//   * columns are dealt to a wave's two accumulator tiles alternately (tile t holds columns 2 r + t), so a lane ends up with PAIRS of adjacent columns: 8-byte
//     stores, 128 contiguous bytes per 16 lanes, and (b [K][N]) one ds_read_b64 per k row and lane for both tiles.
// Bias (forward) and the previous contents of out (accumulating calls) are loaded ahead of the k loop and added behind it (beta C last, as the library GEMM).
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"
#include <algorithm>
#include <cstdio>

namespace amuse {
namespace {

constexpr int kGN = 128, kGK = 32;
// (a chunk of b is 16 KiB either way: 128 rows (n) x 128 B, or 32 rows (k) x 512 B)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {   // (k_vae_fused.hip: LDS-DMA outside hipcc's waitcnt bookkeeping)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int n>
__device__ __forceinline__ void wait_vm_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); }
// waves 4 .. of a workgroup only copy: an LDS-DMA instruction costs its wave 100-190 cycles of issue here (cycle stamps of round 5, profiles/r05_train_gemm_ab.txt; MI355X_MICROARCH.md has 60-185) -
// six of them per chunk in front of 48 MFMAs held the matrix pipe at 60 % when the four MFMA waves copied their own pieces, and TWO copying waves (11 pieces each per
// chunk = ~2,150 cycles against the chunk's 1,536 of MFMAs) were the critical path of every chunk: four (6 x ~150) are not; measured 6-7 % faster than two
constexpr int kGemmCopyWaves = 4;
// MT = 16-row tiles per workgroup (3: 48 rows, 2: 32 rows); TB: b is [N][K] (out = a b^T), else [K][N]
// kGemmBufs = LDS buffers of the chunk ring (3: two chunks in flight under the MFMAs of a third; 2: one - and room for four workgroups per CU)
// (Round 6: a workgroup walking SEVERAL row tiles as one stream of chunks - 512 resident workgroups, the next tile's first chunks fetched under the current tile's
// last MFMAs and epilogue - measured no faster: 20.2 against 19.6 us at 9,600 x 512 x 128, 15.1 against 15.6 at 384 wide; the one-tile launch's dynamic balance is
// worth what the saved prologues are.)
template <int MT, bool TB, int kGemmBufs>
__global__ __launch_bounds__(64 * (4 + kGemmCopyWaves)) void k_train_gemm_tall(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias, float* __restrict__ out,
                                                         int M, int N, int K, int accumulate) {
    constexpr int kAPieces = 2 * MT;                // 1 KiB pieces of a chunk of a (8 rows each)
    constexpr int kPieces = kAPieces + 16;          // + the chunk of b
    constexpr int kBuf = kPieces * 1024;
    constexpr int kPerWave = (kPieces + kGemmCopyWaves - 1) / kGemmCopyWaves;   // pieces of a copying wave: piece p belongs to copying wave p % kGemmCopyWaves (the last ones may hold one less)
    static_assert(kPerWave * kGemmCopyWaves - kPieces < kGemmCopyWaves && (kGemmBufs - 1) * kPerWave < 64, "pieces per copying wave; vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(1024))) char smem[];   // kGemmBufs x kBuf
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, r = lane & 15;
    const int m0 = blockIdx.x * (16 * MT), nb0 = blockIdx.y * kGN;
    const int nchunks = K / kGK;
    [[maybe_unused]] const bool prof_on = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == 4);
    if (wave >= 4) {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
        const int cw = wave - 4;
        const bool full = cw + kGemmCopyWaves * (kPerWave - 1) < kPieces;   // this wave holds kPerWave pieces (else one less)
        // this wave's pieces cw, cw + 2, ...: per-lane source address of chunk 0 (advanced by one chunk per fetch)
        const float* src[kPerWave];
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            const int p = cw + kGemmCopyWaves * i;
            if (p < kAPieces) {
                const int row = 8 * p + (lane >> 3), slot = (lane & 7) ^ ((row >> 1) & 7);
                src[i] = a + (size_t)min(m0 + row, M - 1) * K + 4 * slot;
            } else if constexpr (TB) {
                const int rho = 8 * (p - kAPieces) + (lane >> 3), slot = (lane & 7) ^ ((rho >> 1) & 7);   // image row rho = 32 w + 16 t + r  <-  column 32 w + 2 r + t
                src[i] = b + (size_t)(nb0 + (rho & ~31) + 2 * (rho & 15) + ((rho >> 4) & 1)) * K + 4 * slot;
            } else {
                const int k = 2 * (p - kAPieces) + (lane >> 5), slot = (lane & 31) ^ (((k >> 2) & 1) << 3);
                src[i] = b + (size_t)k * N + nb0 + 4 * slot;
            }
        }
        auto fetch = [&](int buf) {
            const unsigned d = __builtin_amdgcn_readfirstlane(lds0 + buf * kBuf + cw * 1024);
#pragma unroll
            for (int i = 0; i < kPerWave; ++i) {
                const int p = cw + kGemmCopyWaves * i;
                if (p >= kPieces) continue;   // (wave-uniform)
                glds16(src[i], d + i * (kGemmCopyWaves * 1024));
                src[i] += (p < kAPieces || TB) ? kGK : (size_t)kGK * N;
            }
        };
#pragma unroll
        for (int c = 0; c < kGemmBufs - 1; ++c)
            if (c < nchunks) fetch(c);
        int buf = 0;
        for (int c = 0; c < nchunks; ++c) {
            // chunk c has landed: only the chunks issued behind it (up to c + kGemmBufs - 2) may still be in flight.  One barrier per chunk: behind it every MFMA wave
            // has finished chunk c - 1 (its fragment reads have returned), so that chunk's buffer takes chunk c + kGemmBufs - 1 right away
            const int behind = min(nchunks - 1, c + kGemmBufs - 2) - c;
            if (behind == 0) wait_vm_le<0>();
            else if (behind == 1) { if (full) wait_vm_le<kPerWave>(); else wait_vm_le<kPerWave - 1>(); }
            else { if (full) wait_vm_le<2 * kPerWave>(); else wait_vm_le<2 * kPerWave - 2>(); }
            static_assert(kGemmBufs == 2 || kGemmBufs == 3, "the wait counts above");
            __builtin_amdgcn_s_barrier();
            if (c + kGemmBufs - 1 < nchunks) fetch(buf == 0 ? kGemmBufs - 1 : buf - 1);
            buf = buf == kGemmBufs - 1 ? 0 : buf + 1;
        }
        return;
    }
    // bias / previous contents of out: loaded now (ahead of every LDS-DMA: vmcnt retires in order), added behind the k loop as the library GEMM does (beta C last).
    // Lane (g, r), tile (mt, t), element v  ->  out[m0 + 16 mt + 4 g + v][nb0 + 32 wave + 2 r + t]
    const int col = nb0 + 32 * wave + 2 * r;
    float2 init[MT][4];
    {
        const float2 b2 = bias ? *reinterpret_cast<const float2*>(bias + col) : float2{0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0 + 16 * mt + 4 * g + v;
                init[mt][v] = b2;
                if (accumulate && m < M) {
                    const float2 old = *reinterpret_cast<const float2*>(out + (size_t)m * N + col);
                    init[mt][v].x += old.x;
                    init[mt][v].y += old.y;
                }
            }
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = splat4(0.f);
    int buf = 0;
    for (int c = 0; c < nchunks; ++c) {
        __builtin_amdgcn_s_barrier();   // chunk c is in its buffer (the copying waves waited for it)
        const char* A = smem + buf * kBuf;
        const char* B = A + kAPieces * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 af[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = 16 * mt + r;
                af[mt] = *reinterpret_cast<const f32x4*>(A + row * 128 + (((4 * j + g) ^ ((row >> 1) & 7)) << 4));
            }
            float bv[2][4];
            if constexpr (TB) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rho = 32 * wave + 16 * t + r;
                    const f32x4 f = *reinterpret_cast<const f32x4*>(B + rho * 128 + (((4 * j + g) ^ ((rho >> 1) & 7)) << 4));
#pragma unroll
                    for (int s = 0; s < 4; ++s) bv[t][s] = f[s];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = 16 * j + 4 * g + s, u = 16 * wave + r;
                    const float2 f = *reinterpret_cast<const float2*>(B + k * 512 + (((u >> 1) ^ ((g & 1) << 3)) << 4) + (u & 1) * 8);
                    bv[0][s] = f.x;
                    bv[1][s] = f.y;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[mt][t] = mfma_f32(af[mt][s], bv[t][s], acc[mt][t]);
        }
        buf = buf == kGemmBufs - 1 ? 0 : buf + 1;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = m0 + 16 * mt + 4 * g + v;
            if (m < M) *reinterpret_cast<float2*>(out + (size_t)m * N + col) = float2{acc[mt][0][v] + init[mt][v].x, acc[mt][1][v] + init[mt][v].y};
        }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Every OTHER GEMM of the step (the ones rocBLAS ran through round 5): the 333-wide embedding / output layers (skel_embedding, final_layer: K or N = 333 - no
// 16-byte rows, no multiple of 32), the 32-row memory projections of the decoder's one-key cross-attention, the Denoiser's 160-row layers, and every weight
// gradient the chunked kernel of k_train.hip does not take.  One kernel for any shape and both transposes:
//     out[M][N] (+)= opA . opB (+ bias),   opA(m, k) = TA ? a[k][m] : a[m][k],   opB(k, n) = TB ? b[n][k] : b[k][n],   fp32 operands, v_mfma_f32_16x16x4_f32
// 64 x 64 output tile per workgroup (four waves, 32 x 32 each = 2 x 2 accumulator tiles), k in steps of 16 through LDS: every thread loads 4 + 4 elements per step
// (bounds-checked, zero beyond the edges; along the operand's contiguous axis, so a step reads whole 64-byte segments) into [k][row] images with a 65-float
// stride, and a fragment is one ds_read_b32 per MFMA.  Long reductions over few output tiles (weight gradients: K = 9,600 rows into 333 x 128) are cut along k
// over gridDim.z into partial tiles that k_gemm_any_sum adds up in a fixed order (deterministic).  These are the small GEMMs of the step - tens of microseconds
// in total - so the kernel is built for coverage and determinism, not for the last percent: the tall projections stay on k_train_gemm_tall above.
// T = output tile side of a workgroup: 64 (four waves of 32 x 32, k steps of 16) or - for problems whose 64-tiles could not fill the chip: the 32- and 160-row
// GEMMs - 32 (four waves of 16 x 16, k steps of 32: four times the workgroups, a quarter of the dependent MFMAs per step).  Either way a thread moves 4 + 4
// elements per step, and the loads of step i + 1 are in flight while the MFMAs of step i run (these GEMMs are latency chains: 4 .. 32 steps).
template <bool TA, bool TB, int T>
__global__ __launch_bounds__(256) void k_train_gemm_any(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ bias, float* __restrict__ out,
                                                        int M, int N, int K, int kchunk, int accumulate, float* __restrict__ part) {
    constexpr int KS = 1024 / T, kStride = T + 1, NW = T / 32;   // k step; LDS row stride; accumulator tiles per wave and side
    __shared__ float As[KS * kStride], Bs[KS * kStride];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, r = lane & 15;
    const int m0 = blockIdx.x * T, n0 = blockIdx.y * T;
    const int k_lo = blockIdx.z * kchunk, k_hi = min(K, k_lo + kchunk);
    const int wm = (T / 2) * (wave >> 1), wn = (T / 2) * (wave & 1);
    f32x4 acc[NW][NW];
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) acc[i][j] = splat4(0.f);
    // an operand whose contiguous axis is k (a [M][K], b [N][K]): thread = (row, four k's); contiguous along m / n (a [K][M], b [K][N]): thread = (k, four rows)
    constexpr int kPerRow = KS / 4, rPerK = T / 4;
    const int kc_row = tid / kPerRow, kc_k = (tid % kPerRow) * 4, mc_k = tid / rPerK, mc_row = (tid % rPerK) * 4;
    float va[4], vb[4];
    auto load = [&](int k0) {
        if constexpr (!TA) {
            const int m = m0 + kc_row;
#pragma unroll
            for (int e = 0; e < 4; ++e) va[e] = (m < M && k0 + kc_k + e < k_hi) ? a[(size_t)m * K + k0 + kc_k + e] : 0.f;
        } else {
            const int k = k0 + mc_k;
#pragma unroll
            for (int e = 0; e < 4; ++e) va[e] = (k < k_hi && m0 + mc_row + e < M) ? a[(size_t)k * M + m0 + mc_row + e] : 0.f;
        }
        if constexpr (TB) {
            const int n = n0 + kc_row;
#pragma unroll
            for (int e = 0; e < 4; ++e) vb[e] = (n < N && k0 + kc_k + e < k_hi) ? b[(size_t)n * K + k0 + kc_k + e] : 0.f;
        } else {
            const int k = k0 + mc_k;
#pragma unroll
            for (int e = 0; e < 4; ++e) vb[e] = (k < k_hi && n0 + mc_row + e < N) ? b[(size_t)k * N + n0 + mc_row + e] : 0.f;
        }
    };
    if (k_lo < k_hi) load(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += KS) {
        __syncthreads();   // (the previous step's fragments have been read)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (!TA) As[(kc_k + e) * kStride + kc_row] = va[e];
            else As[mc_k * kStride + mc_row + e] = va[e];
            if constexpr (TB) Bs[(kc_k + e) * kStride + kc_row] = vb[e];
            else Bs[mc_k * kStride + mc_row + e] = vb[e];
        }
        __syncthreads();
        if (k0 + KS < k_hi) load(k0 + KS);   // in flight under this step's MFMAs
#pragma unroll
        for (int s = 0; s < KS / 4; ++s) {   // MFMA s: k = k0 + 4 s + g on both operands
            float af[NW], bf[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                af[i] = As[(4 * s + g) * kStride + wm + 16 * i + r];
                bf[i] = Bs[(4 * s + g) * kStride + wn + 16 * i + r];
            }
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int j = 0; j < NW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // C / D layout of the 16 x 16 tile: lane (g, r) holds rows 4 g + v, column r
    float* dst = gridDim.z > 1 ? part + (size_t)blockIdx.z * M * N : out;
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int n = n0 + wn + 16 * j + r;
            if (n >= N) continue;
            const float bv = (bias && gridDim.z == 1) ? bias[n] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m = m0 + wm + 16 * i + 4 * g + v;
                if (m >= M) continue;
                float* o = dst + (size_t)m * N + n;
                *o = (gridDim.z == 1 && accumulate ? *o : 0.f) + acc[i][j][v] + bv;
            }
        }
}
// out (+)= sum over the k chunks' partial tiles, chunk 0 first (+ bias per column)
__global__ __launch_bounds__(256) void k_gemm_any_sum(const float* __restrict__ part, int chunks, size_t mn, int N, const float* __restrict__ bias, int accumulate, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < mn; i += (size_t)gridDim.x * 256) {
        float s = part[i];
        for (int c = 1; c < chunks; ++c) s += part[(size_t)c * mn + i];
        out[i] = (accumulate ? out[i] : 0.f) + s + (bias ? bias[i % N] : 0.f);
    }
}
float* g_any_ws[64][2] = {};   // per device and lane (k_train.hip train_lane)
constexpr size_t kAnyWsFloats = (size_t)4 << 20;   // 16 MB of partial tiles per device

}  // namespace
int train_lane();   // (k_train.hip: the calling thread's scratch lane)

// Which of the step's projections run here: every tall one the tiling covers.  Back to back on hot operands rocBLAS is 20-25 % faster on the un-biased forward
// projections with 384 / 512 outputs (and slower on everything else: biased calls, every input gradient dy W); inside the training step - operands produced by the
// kernel in front, the library's code objects alternating with ours - the step is fastest with ALL of them here: 13.0-13.1 ms of device time per iteration against
// 13.6-15.0 with rocBLAS and 13.9-14.2 with a per-shape mix (profiles/r05_train_gemm_ab.txt).
bool train_gemm_tall_takes(long M, long N, long K, bool /*tb*/, bool /*bias*/) { return M >= 1024 && !(N & 127) && !(K & 31) && N <= 4096 && K <= 4096; }

// out[M][N] = (bias | accumulate: out) + a . (tb ? b^T : b); the caller has checked train_gemm_tall_takes
hipError_t launch_train_gemm_tall(const float* a, const float* b, const float* bias, float* out, long M, long N, long K, bool tb, bool accumulate, hipStream_t stream) {
    // 48-row workgroups unless 32-row ones fill the chip better (a launch is MFMA-bound: its time is the busiest CU's tiles)
    const long wg3 = ((M + 47) / 48) * (N / kGN), wg2 = ((M + 31) / 32) * (N / kGN);
    const long cost3 = ((wg3 + 255) / 256) * 3, cost2 = ((wg2 + 255) / 256) * 2;
    const bool mt3 = cost3 <= cost2;
    const dim3 grid((unsigned)((M + (mt3 ? 47 : 31)) / (mt3 ? 48 : 32)), (unsigned)(N / kGN)), block(64 * (4 + kGemmCopyWaves));
    const int acc = accumulate ? 1 : 0;
    const size_t lds = (size_t)3 * ((mt3 ? 6 : 4) + 16) * 1024;   // three chunk buffers: two chunks in flight under the MFMAs of a third (two measured slower, r05_train_gemm_ab.txt)
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        for (const void* k : {reinterpret_cast<const void*>(&k_train_gemm_tall<3, true, 3>), reinterpret_cast<const void*>(&k_train_gemm_tall<3, false, 3>),
                              reinterpret_cast<const void*>(&k_train_gemm_tall<2, true, 3>), reinterpret_cast<const void*>(&k_train_gemm_tall<2, false, 3>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 22 * 1024);
            if (e != hipSuccess) return e;
        }
        once.set(dev_);
    }
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, lds, stream, a, b, bias, out, (int)M, (int)N, (int)K, acc); };
    if (mt3 && tb) go(k_train_gemm_tall<3, true, 3>);
    else if (mt3) go(k_train_gemm_tall<3, false, 3>);
    else if (tb) go(k_train_gemm_tall<2, true, 3>);
    else go(k_train_gemm_tall<2, false, 3>);
    return hipGetLastError();
}

// out[M][N] (+)= op(a) . op(b) (+ bias[N]) for ANY shape (see above).  Workspace of partial tiles: allocated on the device's first long-reduction call.
hipError_t launch_train_gemm_any(const float* a, const float* b, const float* bias, float* out, long M, long N, long K, bool ta, bool tb, bool accumulate, hipStream_t stream) {
    // 32 x 32 tiles where 64 x 64 ones could not fill the chip (the 32- and 160-row GEMMs)
    const bool small = ((M + 63) / 64) * ((N + 63) / 64) < 128;
    const int T = small ? 32 : 64, KS = 1024 / T;
    const unsigned gx = (unsigned)((M + T - 1) / T), gy = (unsigned)((N + T - 1) / T);
    // cut the reduction when the output tiles alone cannot fill the chip: chunks of >= 256 k's (multiples of the k step), at most 64, bounded by the workspace
    int chunks = 1;
    if ((long)gx * gy < 128 && K >= 1024) {
        chunks = (int)std::min<long>({64L, K / 256, (long)(512 / ((long)gx * gy)), (long)(kAnyWsFloats / (size_t)(M * N))});
        if (chunks < 1) chunks = 1;
    }
    int kchunk = (int)((K + chunks - 1) / chunks);
    kchunk = (kchunk + KS - 1) / KS * KS;
    chunks = (int)((K + kchunk - 1) / kchunk);
    float* part = nullptr;
    if (chunks > 1) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        dev &= 63;
        const int lane = train_lane() & 1;
        if (!g_any_ws[dev][lane]) {
            e = hipMalloc((void**)&g_any_ws[dev][lane], kAnyWsFloats * sizeof(float));
            if (e != hipSuccess) return e;
        }
        part = g_any_ws[dev][lane];
    }
    const dim3 grid(gx, gy, (unsigned)chunks), block(256);
    const int acc = accumulate ? 1 : 0;
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, stream, a, b, bias, out, (int)M, (int)N, (int)K, kchunk, acc, part); };
    if (small) {
        if (ta && tb) go(k_train_gemm_any<true, true, 32>);
        else if (ta) go(k_train_gemm_any<true, false, 32>);
        else if (tb) go(k_train_gemm_any<false, true, 32>);
        else go(k_train_gemm_any<false, false, 32>);
    } else {
        if (ta && tb) go(k_train_gemm_any<true, true, 64>);
        else if (ta) go(k_train_gemm_any<true, false, 64>);
        else if (tb) go(k_train_gemm_any<false, true, 64>);
        else go(k_train_gemm_any<false, false, 64>);
    }
    if (chunks > 1) {
        const size_t mn = (size_t)M * N;
        hipLaunchKernelGGL(k_gemm_any_sum, dim3((unsigned)std::min<size_t>((mn + 255) / 256, 2048)), block, 0, stream, part, chunks, mn, (int)N, bias, acc, out);
    }
    return hipGetLastError();
}

}  // namespace amuse
