// The audio front-end's GEMM (reference models/audio/audio_main_new.py:174-204: the four Linear shapes of a DeiT-B block and the
// patch embedding): C = A . W^T + bias with a fused epilogue, bf16 operands, fp32 accumulation.
//
// Operand delivery is what bounds a bf16 GEMM on this chip (tools/probes/gemm_dma_probe.hip has the experiments), so
// the kernel is built around it:
//   * BOTH operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write.  Activations are
//     kept TILE-MAJOR in HBM (amuse_audio.hpp: a 16-row x 32-k tile is one MFMA fragment, 1 KiB contiguous) and the weights are
//     packed the same way on the host, so every DMA instruction copies 1 KiB of whole cache lines straight into fragment order
//     and every fragment read is a lane-linear ds_read_b128.  (Gathering fragments from row-major rows - sixteen 64-byte pieces per
//     instruction - delivered 8.4 TB/s over the chip and capped the kernel at 650 TFLOP/s, where every earlier variant had stalled.)
//   * 256 features x 128 tokens per workgroup (128 x 128 with a deeper ring for launches of few tiles), k-steps of 32: a stage is
//     24 fragments (24 KiB), a ring of three stages per workgroup, TWO persistent workgroups per CU (one's barrier and epilogue run under the other's MFMAs).  Four waves of
//     128 x 64: 128 accumulator registers, 12 fragment reads per 32 MFMAs.
//   * the fragments of stage s + 1 are read into registers WHILE stage s is multiplied (a W fragment right behind its four MFMAs,
//     the X fragments into a second set), so no LDS round trip is exposed.  The one barrier per stage certifies that stage s + 1 has
//     landed and that every wave holds stage s in registers; that slot is refilled at once - three stages in flight.
//   * a wave blocks at the issue of a vector-memory instruction while the CU's 64 B/clk path is busy, so the six DMA instructions
//     of a stage are issued one per MFMA group, not in a burst behind the barrier.
//   * epilogue in registers (a lane holds 8 consecutive features of a token row); tile-major outputs are written one whole tile
//     per wave instruction, non-temporal (a 240 MB output stream through the L2 evicted the weights).  Stores are
//     fire-and-forget: the two stages after an epilogue wait with vmcnt(6 + stores) - vmcnt retires in order, and the DMA pieces
//     they need are older than the stores.  The bias tile travels by DMA too (an epilogue load would drain the queue).
// Measured on MI355X, M = 38,848 (32 clips), TFLOP/s with epilogue: see DESIGN.md 4.4.
#include <cstdlib>
#include <type_traits>

#include "amuse_dev.hpp"
#include "amuse_audio.hpp"
#include "amuse_kernels.hpp"   // DeviceOnce

namespace amuse {
namespace {

typedef unsigned short bf16raw;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kGemmColGroup = 3;   // feature tiles per column group of the tile walk (measured: 3 beats 6 and the plain row-major walk by 1.5 %)
constexpr int TM = kGemmTM, BK = 32, XFR = TM / 16;
// Two shapes of the same kernel (identical arithmetic per output element: k-steps of 32 in order):
//   FX = 8  256-feature tiles, a wave holds 8 W fragments; stage = 16 W + 8 X fragments (24 KiB), ring of 3 = 74,752 B with the bias
//           tile - two workgroups per CU: the production shape;
//   FX = 4  128-feature tiles, stage = 8 + 8 fragments (16 KiB), ring of 9 = 148,480 B - for launches of few tiles (one or two
//           clips): every workgroup has a CU to itself and walks its whole K extent alone, so what paces it is the fetch latency;
//           twice the workgroups and eight stages in flight instead of two.
constexpr int gemm_stage(int fx) { return (2 * fx + XFR) * 1024; }
constexpr int gemm_lds(int nslot, int fx) { return nslot * gemm_stage(fx) + 1024; }
constexpr int kDeepTiles = 128;   // launches of at most this many 256-feature tiles use the narrow, deep shape

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ u32x4 pack8(f32x4 lo, f32x4 hi) {
    return u32x4{pack2(lo[0], lo[1]), pack2(lo[2], lo[3]), pack2(hi[0], hi[1]), pack2(hi[2], hi[3])};
}
// LDS-DMA: 64 lanes x 16 B from (wave-uniform base + 32-bit lane offset) to LDS [dst, dst + 1 KiB), lane-linear.  Inline asm: the
// compiler does not count it in its s_waitcnt bookkeeping (its own waits can only become longer, never too short: vmcnt retires
// in order); the stage protocol does the counting.
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}

// vector-memory operations of an epilogue that may still be in flight when the next tile's first two stages wait (exact and
// unconditional per wave; 0 = those stages wait strictly)
template <int EPI, int FX>
constexpr int epi_stores() {
    return ((EPI == EPI_BF16 || EPI == EPI_GELU_BF16) ? 16 : (EPI == EPI_RESID_F32 || EPI == EPI_F32) ? 32 : 0) * FX / 8;
}

template <int EPI, int NSLOT, int FX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_tm(GemmArgs a) {
    constexpr int TN = 32 * FX, WFR = 2 * FX, STAGE = gemm_stage(FX);   // features per tile, W fragments per stage
    constexpr int WP = FX / 2, PW = WP + 2;                             // this wave's W pieces / all its DMA pieces per stage
    constexpr int kOffBias = NSLOT * STAGE;                             // [TN] float: the bias of the current tile's features
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), g = lane >> 4, j = lane & 15;
    const int wf = wave >> 1, wr = wave & 1;   // feature half (16 FX), token half (64) of the tile
    const int N = a.N;
    const int tiles_n = N / TN, tiles_m = (a.M + TM - 1) / TM, n_tiles = tiles_n * tiles_m, nk = a.K / BK;
    // PERSISTENT: workgroup w computes tiles w, w + grid, ...; consecutive tiles walk the N tiles of one M tile.  Workgroups are dealt
    // round-robin to the 8 XCDs: give each XCD a contiguous range of w so that tiles that share operand rows share an L2.
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (wg >= n_tiles) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)smem;
    const unsigned voff = lane * 16;
    // fetch cursor (wave-uniform): the stage fetched next = k-step f_k of tile f_tile into slot f_slot.  This wave's PW DMA pieces
    // of a stage: W fragment rows WP wave .. + WP - 1, X fragment rows 2 wave, + 1.
    int f_tile = wg, f_k = 0, f_slot = 0;
    const size_t frag_row = (size_t)nk * 1024;   // bytes between consecutive fragment rows of a packed / tile-major operand
    const char *fw, *fx;
    // tile index -> (row tile, feature tile): feature tiles in groups of kGemmColGroup, row tiles walked inside a group - the
    // workgroups of an XCD (a contiguous range of tile indices) then share the W rows of one group in their L2
    constexpr int CGS = kGemmColGroup;
    const int cg = tiles_n % CGS == 0 ? CGS : tiles_n;
    auto tile_tm = [&](int tile) { return (tile % (tiles_m * cg)) / cg; };
    auto tile_tn = [&](int tile) { return (tile / (tiles_m * cg)) * cg + tile % cg; };
    auto cursor = [&]() {
        fw = reinterpret_cast<const char*>(a.W) + (size_t)(tile_tn(f_tile) * WFR + WP * wave) * frag_row;
        fx = reinterpret_cast<const char*>(a.A) + (size_t)(tile_tm(f_tile) * XFR + 2 * wave) * frag_row;
    };
    cursor();
    auto fetch_piece = [&](int i) {
        const unsigned d = lds0 + f_slot * STAGE;
        if (i < WP) glds16s(fw + i * frag_row + (size_t)f_k * 1024, voff, d + (WP * wave + i) * 1024);
        else glds16s(fx + (i - WP) * frag_row + (size_t)f_k * 1024, voff, d + (WFR + 2 * wave + i - WP) * 1024);
    };
    auto fetch_advance = [&]() {
        f_slot = f_slot == NSLOT - 1 ? 0 : f_slot + 1;
        if (++f_k == nk) {
            f_k = 0;
            const int nt = f_tile + gridDim.x;
            f_tile = nt < n_tiles ? nt : f_tile;   // past the end: the last tile again (lands in a free slot, never read)
            cursor();
        }
    };
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
#pragma unroll
        for (int i = 0; i < PW; ++i) fetch_piece(i);
        fetch_advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PW * (NSLOT - 1)) : "memory");
    bf16x8 wc[FX], xa[4], xb[4];
    {
        const char* sl = smem + lane * 16;
#pragma unroll
        for (int y = 0; y < 4; ++y) xa[y] = *reinterpret_cast<const bf16x8*>(sl + (WFR + 4 * wr + y) * 1024);
#pragma unroll
        for (int x = 0; x < FX; ++x) wc[x] = *reinterpret_cast<const bf16x8*>(sl + (FX * wf + x) * 1024);
    }
    int r_slot = 1;
    constexpr int SN = epi_stores<EPI, FX>();
    f32x4 acc[FX][4];   // [feature fragment][token fragment]
    // one k-step: xc = this stage's X fragments (registers), xn receives the next stage's
    // swapped: the X fragment is the MFMA's A operand, so a lane ends up with 4 consecutive TOKENS of one feature (the V^T tiles)
    // relaxed (wave-uniform, runtime): the stage waits with the epilogue's stores still in flight.  A branch around the two forms of
    // the wait and nothing else: as separate instantiations of the whole k-step (strict for the first tile, relaxed behind an
    // epilogue) the paths met at the k loop with the 32 W-fragment registers allocated differently, and hipcc reconciled them
    // through scratch - 18 to 34 spilled registers per kernel, reloaded behind a full vmcnt wait once per tile.
    auto half = [&](bool relaxed, auto swapped, bf16x8 (&xc)[4], bf16x8 (&xn)[4], int bias_tile) {
        constexpr int WAITN = PW * (NSLOT - 2);
        static_assert(WAITN + SN < 64, "vmcnt is a 6-bit counter");
        // this wave's pieces of the NEXT stage have landed (the NSLOT - 2 younger stages may still fly) and its reads of this stage are in
        // registers; behind the barrier that holds for every wave: the next stage is complete, this stage's slot is free
        if (SN > 0 && relaxed) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITN + SN) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(WAITN) : "memory");
        if (bias_tile >= 0 && wave == 0)   // (every wave's epilogue reads of the previous tile's bias are in front of this barrier)
            glds16s(a.bias + (size_t)bias_tile * TN, (lane % (TN / 4)) * 16, lds0 + kOffBias);   // (TN = 128: the upper lanes repeat the lower half)
        const char* sl = smem + r_slot * STAGE + lane * 16;
#pragma unroll
        for (int y = 0; y < 4; ++y) xn[y] = *reinterpret_cast<const bf16x8*>(sl + (WFR + 4 * wr + y) * 1024);
#pragma unroll
        for (int x = 0; x < FX; ++x) {
#pragma unroll
            for (int y = 0; y < 4; ++y)
                acc[x][y] = decltype(swapped)::value ? mfma_bf16(xc[y], wc[x], acc[x][y]) : mfma_bf16(wc[x], xc[y], acc[x][y]);
            wc[x] = *reinterpret_cast<const bf16x8*>(sl + (FX * wf + x) * 1024);
            if (x < PW) fetch_piece(x);
            if (x == FX - 1) fetch_advance();
            __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from hoisting every read to the top: 48 more live registers)
        }
        r_slot = r_slot == NSLOT - 1 ? 0 : r_slot + 1;
    };
    bool first = true;
    for (int tile = wg; tile < n_tiles; tile += gridDim.x) {
        const int tm = tile_tm(tile), tn = tile_tn(tile);
#pragma unroll
        for (int x = 0; x < FX; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = splat4(0.f);
        const bool vt_tile = EPI == EPI_QKV && tn >= 2 * kAstDim / TN;   // (uniform)
        if (EPI == EPI_QKV && vt_tile) {
#pragma unroll 1
            for (int kp = 0; kp < nk; kp += 2) {
                half(false, std::true_type{}, xa, xb, kp == 0 ? tn : -1);
                half(false, std::true_type{}, xb, xa, -1);
            }
        } else {
#pragma unroll 1
            for (int kp = 0; kp < nk; kp += 2) {
                const bool relaxed = kp == 0 && !first;   // the two stages behind an epilogue
                half(relaxed, std::false_type{}, xa, xb, kp == 0 ? tn : -1);
                half(relaxed, std::false_type{}, xb, xa, -1);
            }
        }
        first = false;
        // ---- epilogue: lane (g, j): token row m0 + 16 y + j, features n0 + 32 p + 8 g .. + 7 = acc[2 p][y], acc[2 p + 1][y]
        const int m0 = tm * TM + 64 * wr, n0 = tn * TN + 16 * FX * wf;
        if (EPI == EPI_QKV && vt_tile) {
            // V^T (swapped MFMAs): lane (g, j) holds row j of feature fragment x - V^T row 16 F + j, F = the fragment's index among
            // the 48 of v - and tokens 16 y + 4 g + m of the wave's 64-row span.  Key SLOT order inside 32 keys: key 16 a + 4 g + m ->
            // slot 8 g + 4 a + m (the order the attention's P operand comes out of its S^T MFMA in), so a lane's values of the
            // fragment pair y = 2 Y, 2 Y + 1 are 8 consecutive slots and the wave's store is one whole tile of the matrix
            // [B * 768][1216].  A 64-row span lies inside one clip (1216 = 19 * 64).
            const float* bw = reinterpret_cast<const float*>(smem + kOffBias) + 16 * FX * wf;
            const int b = m0 / kAstRows, tok0 = m0 - b * kAstRows;
            if (m0 >= a.M) continue;   // (the pad half of the last row tile: no clip owns it)
#pragma unroll
            for (int x = 0; x < FX; ++x) {
                const float bv = bw[32 * (x >> 1) + 8 * (j >> 2) + 4 * (x & 1) + (j & 3)];   // fragment row j <-> this feature (pack_w)
                const size_t rt = (size_t)b * (kAstDim / 16) + ((n0 - 2 * kAstDim) >> 4) + x;   // row tile of the V^T matrix
#pragma unroll
                for (int Y = 0; Y < 2; ++Y) {
                    const f32x4 v0 = acc[x][2 * Y] + splat4(bv), v1 = acc[x][2 * Y + 1] + splat4(bv);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.vt) + (rt * (kAstRows / 32) + ((tok0 >> 5) + Y)) * 1024 + voff) = pack8(v0, v1);
                }
            }
            continue;
        }
        const float* bl = reinterpret_cast<const float*>(smem + kOffBias) + 16 * FX * wf + 8 * g;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const size_t tile0 = (size_t)((m0 >> 4) + y) * (N >> 5) + (n0 >> 5);   // tile-major outputs: tile index of p = 0
            const size_t row = (size_t)m0 + 16 * y + j;
#pragma unroll
            for (int p = 0; p < FX / 2; ++p) {
                const int n = n0 + 32 * p + 8 * g;
                f32x4 v0 = acc[2 * p][y] + ld4(bl + 32 * p), v1 = acc[2 * p + 1][y] + ld4(bl + 32 * p + 4);
                if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_BF16) {
                    if constexpr (EPI == EPI_GELU_BF16) {
                        // erf-GELU through the clamped odd polynomial of amuse_dev.hpp (|error| <= 1.9e-4, below the bf16 rounding that
                        // follows): 6.5 issue slots per element against ~11 for the rcp / exp2 form - the epilogue's VALU work runs with
                        // the matrix pipes idle (profiles/r02_audio_pmc), so it is paid in full
                        v0 = gelu_poly4(v0);
                        v1 = gelu_poly4(v1);
                    }
                    __builtin_nontemporal_store(pack8(v0, v1), reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out_bf16) + (tile0 + p) * 1024 + voff));
                } else if constexpr (EPI == EPI_RESID_F32) {
                    float* c = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out_f32) + (tile0 + p) * 2048 + voff);
                    v0 += ld4(c);
                    v1 += ld4(c + 256);
                    st4(c, v0);
                    st4(c + 256, v1);
                } else if constexpr (EPI == EPI_F32) {
                    float* c = reinterpret_cast<float*>(reinterpret_cast<char*>(a.out_f32) + (tile0 + p) * 2048 + voff);
                    st4(c, v0);
                    st4(c + 256, v1);
                } else if constexpr (EPI == EPI_PATCH) {
                    // patch row = b * 1212 + q  ->  token row b * 1214 + 2 + q, + pos_embed[2 + q]
                    if (row < (size_t)a.M) {
                        const size_t b = row / kAstPatches, q = row - b * kAstPatches;
                        float* dst = a.out_f32 + tm_f32(b * kAstRows + 2 + q, n, kAstDim);
                        const float* ps = a.pos + (2 + q) * kAstDim + n;
                        st4(dst, v0 + ld4(ps));
                        st4(dst + 256, v1 + ld4(ps + 4));
                    }
                } else {  // EPI_QKV, q | k tiles, tile-major [M][1536]; q pre-scaled by head_dim ** -0.5 * log2(e): the attention's scores are
                          // exp2 arguments as they leave its MFMAs
                    const float sc = n < kAstDim ? 0.125f * 1.44269504088896340736f : 1.0f;
                    const size_t qk0 = (size_t)((m0 >> 4) + y) * (2 * kAstDim / 32) + (n0 >> 5);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.out_bf16) + (qk0 + p) * 1024 + voff) = pack8(v0 * sc, v1 * sc);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus fetches must not outlive the workgroup's LDS
}

template <int EPI, int NSLOT, int FX>
hipError_t launch_gemm_n(const GemmArgs& a, int n_tiles, hipStream_t s) {
    static DeviceOnce once;
    int dev_;
    if (!once.done(&dev_)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tm<EPI, NSLOT, FX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           gemm_lds(NSLOT, FX));
        if (e != hipSuccess) return e;
        once.set(dev_);
    }
    const int resident = 2 * 256;   // two workgroups per CU, 256 CUs
    hipLaunchKernelGGL((k_gemm_tm<EPI, NSLOT, FX>), dim3(n_tiles < resident ? n_tiles : resident), dim3(256), gemm_lds(NSLOT, FX), s, a);
    return hipGetLastError();
}
template <int EPI>
hipError_t launch_gemm_t(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + TM - 1) / TM, n_tiles = tiles_m * (a.N / kGemmTN);
    constexpr int deep_tiles = kDeepTiles;
    if (n_tiles <= deep_tiles) return launch_gemm_n<EPI, 9, 4>(a, 2 * n_tiles, s);
    return launch_gemm_n<EPI, 3, 8>(a, n_tiles, s);
}

}  // namespace

hipError_t launch_gemm(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.N % kGemmTN || a.K % (2 * BK) || a.M < 1) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BF16: return launch_gemm_t<EPI_BF16>(a, s);
        case EPI_GELU_BF16: return launch_gemm_t<EPI_GELU_BF16>(a, s);
        case EPI_RESID_F32: return launch_gemm_t<EPI_RESID_F32>(a, s);
        case EPI_F32: return launch_gemm_t<EPI_F32>(a, s);
        case EPI_PATCH: return launch_gemm_t<EPI_PATCH>(a, s);
        default: return launch_gemm_t<EPI_QKV>(a, s);
    }
}

}  // namespace amuse
