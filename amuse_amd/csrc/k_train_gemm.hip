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

}  // namespace

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

}  // namespace amuse
