// Host-side pieces shared by the library's translation units (amuse_api.hip, amuse_variants.hip): state-dict index, MFMA-fragment
// weight packers (amuse_dev.hpp has the device side of the layouts), upload helper, and the context struct.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/amuse_hip.h"
#include "amuse_dev.hpp"
#include "amuse_kernels.hpp"

// error slot of the C ABI (thread-local text behind amuse_last_error; defined in amuse_api.hip)
__attribute__((visibility("hidden"), format(printf, 2, 3))) int amuse_failf(int code, const char* fmt, ...);
#define fail(...) amuse_failf(__VA_ARGS__)

namespace amuse {
// amuse_update_weights_device learns the packed images' gather maps by running the builders on probe parameters with upload()
// redirected into host memory, keyed by the context slot the image belongs to
struct Capture {
    std::map<void**, std::vector<unsigned char>> bufs;
};
__attribute__((visibility("hidden"))) inline thread_local Capture* g_capture = nullptr;
__attribute__((visibility("hidden"))) inline bool g_probe_f16 = false;   // build_repack_maps: the lo units carry the probe value too (amuse_update_weights_device)
}  // namespace amuse

using namespace amuse;

namespace {
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AMUSE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------- state-dict index (order = reference)
struct ParamIndex {
    std::map<std::string, std::pair<size_t, size_t>> m;  // name -> (offset, numel)
    size_t total = 0;
    void add(const std::string& n, size_t numel) { m[n] = {total, numel}; total += numel; }
};

void enc_layer(ParamIndex& P, const std::string& p) {
    P.add(p + ".self_attn.in_proj_weight", 384 * 128); P.add(p + ".self_attn.in_proj_bias", 384);
    P.add(p + ".self_attn.out_proj.weight", 128 * 128); P.add(p + ".self_attn.out_proj.bias", 128);
    P.add(p + ".linear1.weight", 512 * 128); P.add(p + ".linear1.bias", 512);
    P.add(p + ".linear2.weight", 128 * 512); P.add(p + ".linear2.bias", 128);
    P.add(p + ".norm1.weight", 128); P.add(p + ".norm1.bias", 128);
    P.add(p + ".norm2.weight", 128); P.add(p + ".norm2.bias", 128);
}
void dec_layer(ParamIndex& P, const std::string& p) {
    P.add(p + ".self_attn.in_proj_weight", 384 * 128); P.add(p + ".self_attn.in_proj_bias", 384);
    P.add(p + ".self_attn.out_proj.weight", 128 * 128); P.add(p + ".self_attn.out_proj.bias", 128);
    P.add(p + ".multihead_attn.in_proj_weight", 384 * 128); P.add(p + ".multihead_attn.in_proj_bias", 384);
    P.add(p + ".multihead_attn.out_proj.weight", 128 * 128); P.add(p + ".multihead_attn.out_proj.bias", 128);
    P.add(p + ".linear1.weight", 512 * 128); P.add(p + ".linear1.bias", 512);
    P.add(p + ".linear2.weight", 128 * 512); P.add(p + ".linear2.bias", 128);
    for (const char* n : {"norm1", "norm2", "norm3"}) { P.add(p + "." + n + ".weight", 128); P.add(p + "." + n + ".bias", 128); }
}
std::string blk_name(const std::string& prefix, int blk) {
    if (blk < 4) return prefix + ".input_blocks." + std::to_string(blk);
    if (blk == 4) return prefix + ".middle_block";
    return prefix + ".output_blocks." + std::to_string(blk - 5);
}
void skip_stack(ParamIndex& P, const std::string& prefix, bool dec) {
    P.add(prefix + ".norm.weight", 128); P.add(prefix + ".norm.bias", 128);
    for (int b = 0; b < 9; ++b) dec ? dec_layer(P, blk_name(prefix, b)) : enc_layer(P, blk_name(prefix, b));
    for (int i = 0; i < 4; ++i) {
        P.add(prefix + ".linear_blocks." + std::to_string(i) + ".weight", 128 * 256);
        P.add(prefix + ".linear_blocks." + std::to_string(i) + ".bias", 128);
    }
}
ParamIndex denoiser_index() {
    ParamIndex P;
    P.add("time_embedding.linear_1.weight", 128 * 256); P.add("time_embedding.linear_1.bias", 128);
    P.add("time_embedding.linear_2.weight", 128 * 128); P.add("time_embedding.linear_2.bias", 128);
    for (const char* n : {"con", "emo", "sty"}) {
        P.add(std::string("emb_proj_") + n + ".1.weight", 128 * 256);
        P.add(std::string("emb_proj_") + n + ".1.bias", 128);
    }
    P.add("query_pos.pe", 500 * 128); P.add("mem_pos.pe", 500 * 128);
    skip_stack(P, "encoder", false);
    return P;
}
ParamIndex prior_index() {
    ParamIndex P;
    P.add("global_motion_token", 2 * 128);
    P.add("query_pos_encoder.pe", 500 * 128); P.add("query_pos_decoder.pe", 500 * 128);
    skip_stack(P, "encoder", false);
    skip_stack(P, "decoder", true);
    P.add("skel_embedding.weight", 128 * 333); P.add("skel_embedding.bias", 128);
    P.add("final_layer.weight", 333 * 128); P.add("final_layer.bias", 333);
    return P;
}
struct Params {
    const ParamIndex& idx;
    const float* base;
    const float* get(const std::string& n) const { return base + idx.m.at(n).first; }
};

// ---------------------------------------------------------------- MFMA-fragment packing (see amuse_dev.hpp)
uint16_t f2bf(float f) {  // round-to-nearest-even, as v_cvt_pk_bf16_f32
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}
// fp32 -> fp16 bits, round-to-nearest-even with gradual underflow (what v_cvt_pk_f16_f32 / (_Float16) do), and back (exact)
uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          // NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);         // >= 65520 rounds to infinity
    if (x < 0x38800000u) {                                            // below 2^-14: the result is subnormal (or 2^-14)
        float a;
        memcpy(&a, &x, 4);
        return (uint16_t)(sign | (uint16_t)nearbyintf(a * 16777216.0f));   // units of 2^-24, ties to even
    }
    x -= 0x38000000u;                                                 // re-bias the exponent (127 -> 15)
    x += 0xfffu + ((x >> 13) & 1u);
    return (uint16_t)(sign | (x >> 13));
}
float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float f;
    if (e == 0) {
        f = (float)m * 5.9604644775390625e-8f;                        // m * 2^-24
        uint32_t u;
        memcpy(&u, &f, 4);
        u |= sign;
        memcpy(&f, &u, 4);
        return f;
    }
    const uint32_t u = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
    memcpy(&f, &u, 4);
    return f;
}
// W: [n_out x K] row-major.  Appends units for (k-tile outer, out-tile inner); PREC_F16X2: k-tile pair outer, out-tile inner,
// two units each - hi = rn16(w), lo = rn16(w - hi) (amuse_dev.hpp gemm_ring_s).
void pack_gemm(std::vector<uint4>& out, int prec, const float* W, int n_out, int K, const std::vector<int>& otiles,
               const std::vector<int>& ktiles) {
    // (amuse_update_weights calls this once per training iteration: the destination is sized once and filled through a
    // pointer, rows / columns inside the matrix skip the bounds checks)
    const size_t nunits = is_op16(prec) ? (ktiles.size() / 2) * otiles.size() : ktiles.size() * otiles.size();
    uint16_t (*const cv16)(float) = prec == PREC_F16 ? f2h : f2bf;   // the one-piece 16-bit formats differ in the conversion only
    const size_t base = out.size();
    out.resize(base + nunits * 64);
    uint4* dst = out.data() + base;
    auto at = [&](int row, int col) -> float { return (row < n_out && col < K) ? W[(size_t)row * K + col] : 0.f; };
    if (prec == PREC_F32) {
        for (int t : ktiles)
            for (int o : otiles) {
                const bool inside = 16 * o + 16 <= n_out && 16 * t + 16 <= K;
                for (int lane = 0; lane < 64; ++lane, ++dst) {
                    const int g = lane >> 4, i = lane & 15;
                    float v[4];
                    if (inside) memcpy(v, W + (size_t)(16 * o + i) * K + 16 * t + 4 * g, 16);
                    else
                        for (int m = 0; m < 4; ++m) v[m] = at(16 * o + i, 16 * t + 4 * g + m);
                    memcpy(dst, v, 16);
                }
            }
    } else if (prec == PREC_F16X2) {
        for (size_t c = 0; c + 1 < ktiles.size(); c += 2) {
            const int t0 = ktiles[c], t1 = ktiles[c + 1];
            for (int o : otiles) {
                for (int lane = 0; lane < 64; ++lane, ++dst) {
                    const int g = lane >> 4, i = lane & 15;
                    uint16_t hi[8], lo[8];
                    for (int e = 0; e < 8; ++e) {
                        const float w = at(16 * o + i, 16 * (e < 4 ? t0 : t1) + 4 * g + (e & 3));
                        hi[e] = f2h(w);
                        lo[e] = g_probe_f16 ? hi[e] : f2h(w - h2f(hi[e]));
                    }
                    memcpy(dst, hi, 16);
                    memcpy(dst + 64, lo, 16);
                }
                dst += 64;
            }
        }
    } else {
        for (size_t c = 0; c + 1 < ktiles.size(); c += 2) {
            const int t0 = ktiles[c], t1 = ktiles[c + 1];
            for (int o : otiles) {
                const bool inside = 16 * o + 16 <= n_out && 16 * t0 + 16 <= K && 16 * t1 + 16 <= K;
                for (int lane = 0; lane < 64; ++lane, ++dst) {
                    const int g = lane >> 4, i = lane & 15;
                    uint16_t v[8];
                    if (inside) {
                        const float* r0 = W + (size_t)(16 * o + i) * K + 16 * t0 + 4 * g;
                        const float* r1 = W + (size_t)(16 * o + i) * K + 16 * t1 + 4 * g;
                        for (int e = 0; e < 4; ++e) { v[e] = cv16(r0[e]); v[4 + e] = cv16(r1[e]); }
                    } else {
                        for (int e = 0; e < 4; ++e) {
                            v[e] = cv16(at(16 * o + i, 16 * t0 + 4 * g + e));
                            v[4 + e] = cv16(at(16 * o + i, 16 * t1 + 4 * g + e));
                        }
                    }
                    memcpy(dst, v, 16);
                }
            }
        }
    }
}
std::vector<int> range(int a, int b) { std::vector<int> r; for (int i = a; i < b; ++i) r.push_back(i); return r; }

void fill_block_pvec(float* pv, const Params& P, const std::string& p, bool dec) {
    memcpy(pv + PV_IN_B, P.get(p + ".self_attn.in_proj_bias"), 384 * 4);
    memcpy(pv + PV_OUT_B, P.get(p + ".self_attn.out_proj.bias"), 128 * 4);
    memcpy(pv + PV_L1_B, P.get(p + ".linear1.bias"), 512 * 4);
    memcpy(pv + PV_L2_B, P.get(p + ".linear2.bias"), 128 * 4);
    memcpy(pv + PV_LN1_W, P.get(p + ".norm1.weight"), 128 * 4); memcpy(pv + PV_LN1_B, P.get(p + ".norm1.bias"), 128 * 4);
    memcpy(pv + PV_LN2_W, P.get(p + ".norm2.weight"), 128 * 4); memcpy(pv + PV_LN2_B, P.get(p + ".norm2.bias"), 128 * 4);
    if (dec) { memcpy(pv + PV_LN3_W, P.get(p + ".norm3.weight"), 128 * 4); memcpy(pv + PV_LN3_B, P.get(p + ".norm3.bias"), 128 * 4); }
}
std::vector<float> build_pvec(const Params& P, const std::string& prefix, bool dec) {
    std::vector<float> pv(PV_TOTAL, 0.f);
    for (int b = 0; b < 9; ++b) fill_block_pvec(pv.data() + b * PV_BLOCK, P, blk_name(prefix, b), dec);
    for (int i = 0; i < 4; ++i)
        memcpy(pv.data() + PV_SKIP_B + i * 128, P.get(prefix + ".linear_blocks." + std::to_string(i) + ".bias"), 128 * 4);
    memcpy(pv.data() + PV_FINAL_W, P.get(prefix + ".norm.weight"), 128 * 4);
    memcpy(pv.data() + PV_FINAL_B, P.get(prefix + ".norm.bias"), 128 * 4);
    return pv;
}
// the per-wave pieces shared by encoder and decoder blocks
void pack_qkv(std::vector<uint4>& s, int prec, const float* in_w, int h, bool v_separate) {
    if (v_separate) {  // sampler: q,k tiles as one 4-tile GEMM, then v (operand-swapped on the device)
        pack_gemm(s, prec, in_w, 384, 128, {2 * h, 2 * h + 1, 8 + 2 * h, 8 + 2 * h + 1}, range(0, 8));
        pack_gemm(s, prec, in_w, 384, 128, {16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
    } else {
        pack_gemm(s, prec, in_w, 384, 128, {2 * h, 2 * h + 1, 8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
    }
}
void pack_outproj_ffn(std::vector<uint4>& s, int prec, const Params& P, const std::string& p, int w) {
    pack_gemm(s, prec, P.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * w, 2 * w + 1});
    pack_gemm(s, prec, P.get(p + ".linear1.weight"), 512, 128, range(8 * w, 8 * w + 8), range(0, 8));
    pack_gemm(s, prec, P.get(p + ".linear2.weight"), 128, 512, range(0, 8), range(8 * w, 8 * w + 8));
}
// sampler order: out_proj, then the FFN in four software-pipelined quarters (k_sampler.hip encoder_block)
void pack_outproj_ffn_quarters(std::vector<uint4>& s, int prec, const Params& P, const std::string& p, int w) {
    pack_gemm(s, prec, P.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * w, 2 * w + 1});
    // software-pipelined order of k_sampler.hip: F1q0 F1q1 F2q0 F1q2 F2q1 F1q3 F2q2 F2q3
    auto f1 = [&](int q) { const int h0 = 8 * w + 2 * q; pack_gemm(s, prec, P.get(p + ".linear1.weight"), 512, 128, {h0, h0 + 1}, range(0, 8)); };
    auto f2 = [&](int q) { const int h0 = 8 * w + 2 * q; pack_gemm(s, prec, P.get(p + ".linear2.weight"), 128, 512, range(0, 8), {h0, h0 + 1}); };
    f1(0); f1(1); f2(0); f1(2); f2(1); f1(3); f2(2); f2(3);
}
void pack_skiplin(std::vector<uint4>& s, int prec, const Params& P, const std::string& prefix, int i, int w) {
    pack_gemm(s, prec, P.get(prefix + ".linear_blocks." + std::to_string(i) + ".weight"), 128, 256, range(0, 8),
              range(4 * w, 4 * w + 4));
}

// first call allocates; later calls (amuse_update_weights: same architecture, same sizes) overwrite in place
template <typename T>
int upload(T** dst, const void* src, size_t bytes) {
    if (g_capture) {
        const unsigned char* b = static_cast<const unsigned char*>(src);
        g_capture->bufs[reinterpret_cast<void**>(dst)].assign(b, b + bytes);
        return 0;
    }
    if (!*dst) HIP_TRY(hipMalloc((void**)dst, bytes));
    HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
std::vector<float> transpose(const float* w, int rows, int cols) {  // [rows][cols] -> [cols][rows]
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
    return t;
}

}  // namespace

struct amuse_variant;   // the Denoiser variants' streams and tables (amuse_variants.hip)
// ---- The launch plan: ONE statement of which kernels a job takes (exported as amuse_plan, include/amuse_hip.h; amuse_amd/shard.py and
// tests/c_client call that - nothing restates these rules).  Every choice is keyed by the clip count of the CALL / the JOB, never of a chunk or a shard.
constexpr int kFusedMinClips = 64;   // the per-clip 16-bit kernels occupy one CU per clip: they win once the clips fill a good part of the chip (profiles/r03_decode_perf.txt:
                                     // fused 0.61 ms for any B <= 128; staged 0.51 ms at 32 clips, 0.66 ms at 64, 1.07 ms at 128)
// Do the clips fill rounds of the chip's 256 CUs well enough for the fp32x per-clip kernels (k_vae_fusedx.hip)?  A clip takes ~1.5 ms on its CU whatever the batch: from 160
// clips in the first round, in round r >= 2 with at least 164 - 50 (r - 2) clips in it (profiles/r05_fusedx_decode.txt, ms at 160 / 256 / 384 / 512 / 768 / 1024 clips:
// 1.49 1.65 3.05 3.18 4.76 6.32 against 1.67 1.86 2.84 3.88 5.81 7.75 on the row / attention launches)
inline bool fills_rounds(int B) {
    if (B < 160) return false;
    const int r = (B + 255) / 256, in_last = B - 256 * (r - 1);
    return r == 1 || in_last >= 164 - 50 * (r - 2);
}
// clips per 16-row tile of the latent trans_enc sampler (k_sampler*.hip); the other Denoiser variants have no such choice (1).  One clip per tile up to 128 clips, then fatter
// tiles: a step costs the same for 1..16/tokens clips per tile, and 128 busy CUs run it 6-7 % faster than 256 - with every CU re-streaming the whole network each step the
// 256-workgroup launch sits at the L2's delivery limit (profiles/r01_batch_sweep.txt: 256 clips 36.0 ms with one clip per tile, 33.6 ms with two or three).
inline int plan_clips_per_group(int arch, int B, int tokens) {
    if (arch != AMUSE_ARCH_ENC) return 1;
    const int gmax = 16 / tokens;
    int g = (B + 127) / 128;
    if (g > gmax) g = gmax;
    return g < 1 ? 1 : g;
}
// MotionPrior.decode: fp32 has one kernel family (STAGED); bf16 / fp16: the fused per-clip kernel (k_vae_fused.hip) from kFusedMinClips; fp32x: the per-clip kernel
// (k_vae_fusedx.hip, CLIP) where the clips fill rounds, else the no-split-K row kernel (k_vae_rows8.hip, FUSED) from kFusedMinClips, else the split-K row kernel (STAGED)
inline int plan_decode_path(int precision, int B) {
    if (precision == AMUSE_PREC_F32) return AMUSE_DECODE_STAGED;
    if (precision == AMUSE_PREC_F32X && fills_rounds(B)) return AMUSE_DECODE_CLIP;
    return B >= kFusedMinClips ? AMUSE_DECODE_FUSED : AMUSE_DECODE_STAGED;
}
// MotionPrior.encode: only the fp32x mode has more than the staged kernels (the decode's rule)
inline int plan_encode_path(int precision, int B) { return precision == AMUSE_PREC_F32X ? plan_decode_path(precision, B) : AMUSE_DECODE_STAGED; }
// one Denoiser step of the pose-space trans_enc variant (S = 304 rows per clip): the decode's rule on k_den_fused / k_vae_rows8x<ENC> / k_den_fusedx; the other variants: STAGED
inline int plan_step_path(int arch, int precision, int B) { return arch == AMUSE_ARCH_ENC_POSE ? plan_decode_path(precision, B) : AMUSE_DECODE_STAGED; }
// a pin (amuse_set_decode_path) against what the precision has: FUSED = the fused kernel of the 16-bit modes / the row kernel without split-K of fp32x; CLIP = fp32x's
// per-clip kernel, FUSED in the modes that have no third kernel; fp32 has one family
inline int resolve_path(int pin, int planned, int precision) {
    if (pin == AMUSE_DECODE_AUTO) return planned;
    if (precision == AMUSE_PREC_F32 || pin == AMUSE_DECODE_STAGED) return AMUSE_DECODE_STAGED;
    return (pin == AMUSE_DECODE_CLIP && precision == AMUSE_PREC_F32X) ? AMUSE_DECODE_CLIP : AMUSE_DECODE_FUSED;
}

struct amuse_ctx {
    int device = 0;
    int arch = AMUSE_ARCH_ENC;         // Denoiser variant (amuse_create_arch); anything but AMUSE_ARCH_ENC runs through `var`
    amuse_variant* var = nullptr;
    bool has_prior = true;             // pose-space variants may be created without MotionPrior weights
    int clips_per_group = 0;
    int decode_path = AMUSE_DECODE_AUTO;
    int last_plan[4] = {0, 0, 0, 0};   // amuse_debug_last_plan: clips per tile / decode / encode / step path the last call of each kind actually took
    float* decode_tap = nullptr;       // amuse_debug_set_decode_tap
    int ablate = 0;                    // amuse_debug_set_ablation
    // denoiser
    uint4* den_w[3] = {nullptr, nullptr, nullptr};   // 4-wave kernel streams: fp32 | bf16 | split-fp16 (fp32x)
    uint32_t den_wave_units[3] = {0, 0, 0};
    uint4* den_w8 = nullptr;           // bf16 streams of the 8-wave kernel (k_sampler8.hip)
    uint32_t den_w8_units[2] = {0, 0}; // per-step units of a group-A / group-B wave
    uint4* den_w8h = nullptr;          // fp16 streams of the same kernel built for fp16 operands (k_sampler8h.hip, AMUSE_PREC_F16)
    uint4* den_w8x = nullptr;          // split-fp16 streams of the 8-wave fp32x kernel (k_sampler8x.hip)
    uint32_t den_w8x_units[2] = {0, 0};
    float* den_pvec = nullptr;
    float* den_pe = nullptr;           // [500][128]
    float* den_freqs = nullptr;        // [128]
    float *te_w1t = nullptr, *te_b1 = nullptr, *te_w2t = nullptr, *te_b2 = nullptr;
    float* cond_wt[3] = {nullptr, nullptr, nullptr};
    float* cond_b[3] = {nullptr, nullptr, nullptr};
    // prior decoder
    uint4* vae_w[4] = {nullptr, nullptr, nullptr, nullptr};   // staged decode streams: fp32 | bf16 | split-fp16 (fp32x) | fp16
    uint32_t vae_stage_base[4][kVaeStages];
    uint32_t vae_stage_units[4][kVaeStages];
    uint4* vae_wf = nullptr;           // bf16 stream of the fused decode kernel (k_vae_fused.hip)
    uint4* vae_wfh = nullptr;          // its fp16 twin (k_vae_fusedh.hip, AMUSE_PREC_F16)
    float* vae_c1[4] = {nullptr, nullptr, nullptr, nullptr};   // block 0's self-attention half of the fused decoder, bf16 | fp16 build: [300][128] (+ the tap scratch behind it); [2]: of the fp32x row stages
    bool vae_c1_valid[4] = {false, false, false, false};   // (re)computed by the next fused decode after a weight change
    hipEvent_t vae_c1_ev[4] = {nullptr, nullptr, nullptr, nullptr};     // recorded behind the launches that produced vae_c1[i] ...
    hipStream_t vae_c1_stream[4] = {nullptr, nullptr, nullptr, nullptr}; // ... on this stream: a decode on ANOTHER stream waits on the event first
    uint4* vae_w8x = nullptr;          // fp32x row stages without split-K (k_vae_rows8.hip): one stream per stage, consumption order
    uint32_t vae_w8x_base[kVaeStages];
    uint4* vaee_wfx = nullptr;         // fp32x encoder as one per-clip kernel (k_vae_fusedx.hip k_den_fusedx<encode>): one stream for the whole network
    uint4* vae_wfx = nullptr;          // fp32x fused decoder (k_vae_fusedx.hip): one stream of unit pairs for the whole network, consumption order
    uint4* vaee_w8x = nullptr;         // the same for MotionPrior.encode's stages 1..9 (AMUSE_UPD_ENCODER | AMUSE_UPD_F32X)
    uint32_t vaee_w8x_base[kVaeStages];
    uint4* vae_skip = nullptr; size_t vae_skip_cap = 0;   // clips
    float* vae_ca_ws = nullptr; size_t vae_ca_cap = 0;    // clips
    float *vae_pvec = nullptr, *vae_final_bias = nullptr, *vae_pe = nullptr;
    float *vae_wv_t = nullptr, *vae_bv = nullptr, *vae_wo_t = nullptr, *vae_bo = nullptr;
    // prior encoder (MotionPrior.encode)
    uint4* vaee_w[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t vaee_stage_base[4][kVaeStages];
    uint32_t vaee_stage_units[4][kVaeStages];
    float *vaee_pvec = nullptr, *vaee_pe = nullptr, *vaee_tok = nullptr, *vaee_emb_bias = nullptr;
    // schedule
    int T = 0;
    int* d_timesteps = nullptr;
    float *d_coef = nullptr, *d_time_tok = nullptr;
    int* d_ts1 = nullptr;
    float *d_tt1 = nullptr, *d_coef1 = nullptr;
    // workspaces
    float* cond_tok = nullptr; size_t cond_cap = 0;
    float* lat_tmp = nullptr; size_t lat_cap = 0;
    float* fwd_ws = nullptr; size_t fwd_cap = 0;
    float* vae_ws = nullptr; size_t vae_cap = 0;  // clips
    int* d_lengths = nullptr; size_t len_cap = 0;
    std::vector<void*> owned;
    // amuse_update_weights_device: one entry per packed image (built on the first call)
    struct Repack { void** slot; int* map; size_t n; int prior, kind, cls; };
    std::vector<Repack> repack;
};

namespace {
int ensure(float** p, size_t* cap, size_t need_floats) {
    if (*cap >= need_floats) return 0;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *cap = 0;
    HIP_TRY(hipMalloc((void**)p, need_floats * sizeof(float)));
    *cap = need_floats;
    return 0;
}
}  // namespace
