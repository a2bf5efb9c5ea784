// C ABI of the audio front-end (include/amuse_hip.h, "Audio front-end"): context, bf16 weight images, workspace and
// the launch sequence of one AST encoder.  Host code only - kernels live in k_audio.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/amuse_hip.h"
#include "amuse_audio.hpp"

using namespace amuse;

int amuse_fail_msg(int code, const char* msg);   // amuse_api.hip: the library's thread-local error slot

namespace {

int failf(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    char buf[400];
    snprintf(buf, sizeof(buf), fmt, a, b, c);
    return amuse_fail_msg(code, buf);
}
#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return failf(AMUSE_EHIP, "%s (line %ld)", hipGetErrorString(e_), __LINE__); \
    } while (0)

unsigned short f2bf(float f) {  // round-to-nearest-even, as v_cvt_pk_bf16_f32
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x >> 16) | 0x40);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (unsigned short)(x >> 16);
}

struct Block {
    float *n1w, *n1b, *n2w, *n2b, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    unsigned short *qkv_w, *proj_w, *fc1_w, *fc2_w;
};
struct Encoder {
    float *cls, *dist, *pos, *patch_b, *norm_w, *norm_b, *fh_ln_w, *fh_ln_b, *fh_b;
    unsigned short *patch_w, *fh_w;
    Block blk[kAstLayers];
};
constexpr int kChunk = 32;   // clips per pass over the network (about 22 MB of workspace per clip)
// amuse_audio_features runs the three encoders (independent networks over the same fbank) concurrently, each on its own
// stream and workspace: at small batches one encoder's launches leave most of the chip idle (10 row tiles of 128 tokens per
// clip against 512 persistent workgroup slots), at large ones the other encoders' work fills the tail of every launch.
// activations of one encoder pass over `cap` clips
struct Workspace {
    int cap = 0;
    float *X = nullptr, *pooled = nullptr;
    unsigned short *H = nullptr, *QK = nullptr, *Vt = nullptr, *O = nullptr, *F = nullptr, *P = nullptr;
};

}  // namespace

struct amuse_audio_ctx {
    int device = 0;
    int frame_based = 1;
    float norm_mean = 0.f, norm_std = 1.f;
    float *melw = nullptr, *window = nullptr;   // melw: the mel filter bank TRANSPOSED, [257 bins][128 filters]
    int* mel_range = nullptr;                   // [128][2]: first / end bin of each filter's support
    Encoder enc[3];
    std::vector<void*> owned;
    Workspace ws[3];             // one per encoder stream (amuse_audio_encode uses [0])
    float* fbank = nullptr;      // fbanks of one chunk
    int fbank_cap = 0;
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
};

namespace {

int up_f32(amuse_audio_ctx* c, float** dst, const float* src, size_t n) {
    HIP_TRY(hipMalloc((void**)dst, n * sizeof(float)));
    c->owned.push_back(*dst);
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}
// torch Linear weight [N][K] fp32 -> the GEMM's fragment order (amuse_audio.hpp GemmArgs::W)
std::vector<unsigned short> pack_w(const float* W, int N, int K) {
    std::vector<unsigned short> out((size_t)N * K);
    size_t o = 0;
    for (int sp = 0; sp < N / 64; ++sp)
        for (int x = 0; x < 4; ++x)
            for (int ks = 0; ks < K / 32; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15;
                    const int f = 64 * sp + 32 * (x >> 1) + 8 * (i >> 2) + 4 * (x & 1) + (i & 3);
                    for (int e = 0; e < 8; ++e) out[o++] = f2bf(W[(size_t)f * K + 32 * ks + 8 * g + e]);
                }
    return out;
}
int up_packed(amuse_audio_ctx* c, unsigned short** dst, const float* src, int N, int K) {
    const std::vector<unsigned short> h = pack_w(src, N, K);
    HIP_TRY(hipMalloc((void**)dst, h.size() * 2));
    c->owned.push_back(*dst);
    HIP_TRY(hipMemcpy(*dst, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    return 0;
}
int up_bf16(amuse_audio_ctx* c, unsigned short** dst, const float* src, size_t n) {
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(src[i]);
    HIP_TRY(hipMalloc((void**)dst, n * 2));
    c->owned.push_back(*dst);
    HIP_TRY(hipMemcpy(*dst, h.data(), n * 2, hipMemcpyHostToDevice));
    return 0;
}

int build_encoder(amuse_audio_ctx* c, Encoder& E, const float* p) {
    const size_t D = kAstDim;
    auto take = [&](size_t n) { const float* q = p; p += n; return q; };
    if (up_f32(c, &E.cls, take(D), D) || up_f32(c, &E.dist, take(D), D) ||
        up_f32(c, &E.pos, take((size_t)kAstTokens * D), (size_t)kAstTokens * D) ||
        up_packed(c, &E.patch_w, take(D * 256), (int)D, 256) || up_f32(c, &E.patch_b, take(D), D))
        return AMUSE_EHIP;
    for (int l = 0; l < kAstLayers; ++l) {
        Block& b = E.blk[l];
        if (up_f32(c, &b.n1w, take(D), D) || up_f32(c, &b.n1b, take(D), D) ||
            up_packed(c, &b.qkv_w, take(3 * D * D), (int)(3 * D), (int)D) || up_f32(c, &b.qkv_b, take(3 * D), 3 * D) ||
            up_packed(c, &b.proj_w, take(D * D), (int)D, (int)D) || up_f32(c, &b.proj_b, take(D), D) ||
            up_f32(c, &b.n2w, take(D), D) || up_f32(c, &b.n2b, take(D), D) ||
            up_packed(c, &b.fc1_w, take((size_t)kAstMlp * D), kAstMlp, (int)D) || up_f32(c, &b.fc1_b, take(kAstMlp), kAstMlp) ||
            up_packed(c, &b.fc2_w, take(D * kAstMlp), (int)D, kAstMlp) || up_f32(c, &b.fc2_b, take(D), D))
            return AMUSE_EHIP;
    }
    if (up_f32(c, &E.norm_w, take(D), D) || up_f32(c, &E.norm_b, take(D), D) || up_f32(c, &E.fh_ln_w, take(D), D) ||
        up_f32(c, &E.fh_ln_b, take(D), D) || up_bf16(c, &E.fh_w, take((size_t)kAstFeat * D), (size_t)kAstFeat * D) ||
        up_f32(c, &E.fh_b, take(kAstFeat), kAstFeat))
        return AMUSE_EHIP;
    return 0;
}

size_t pad128(size_t m) { return (m + 127) / 128 * 128; }

void free_ws(Workspace& w) {
    void* old[] = {w.X, w.pooled, w.H, w.QK, w.Vt, w.O, w.F, w.P};
    for (void* p : old)
        if (p) (void)hipFree(p);
    w = Workspace{};
}
int ensure_ws(Workspace& w, int nb) {
    if (w.cap >= nb) return 0;
    free_ws(w);
    const size_t Mp = pad128((size_t)nb * kAstRows);
    HIP_TRY(hipMalloc((void**)&w.X, Mp * kAstDim * 4));
    HIP_TRY(hipMalloc((void**)&w.pooled, (size_t)nb * kAstPoolSplit * kAstDim * 4));
    HIP_TRY(hipMalloc((void**)&w.H, Mp * kAstDim * 2));
    HIP_TRY(hipMalloc((void**)&w.QK, Mp * 2 * kAstDim * 2));
    HIP_TRY(hipMalloc((void**)&w.Vt, (size_t)nb * kAstDim * kAstKeysPad * 2));
    HIP_TRY(hipMalloc((void**)&w.O, Mp * kAstDim * 2));
    HIP_TRY(hipMalloc((void**)&w.F, Mp * kAstMlp * 2));
    HIP_TRY(hipMalloc((void**)&w.P, pad128((size_t)nb * kAstPatches) * 256 * 2));
    // every activation is tile-major (amuse_audio.hpp), rows padded to the GEMM's 128-token tile.  Pad rows are read by the GEMM
    // tiles and the LayerNorm (their results stay in pad rows) and the V^T pad columns by the attention (masked): they only
    // have to be finite
    HIP_TRY(hipMemset(w.X, 0, Mp * kAstDim * 4));
    HIP_TRY(hipMemset(w.H, 0, Mp * kAstDim * 2));
    HIP_TRY(hipMemset(w.O, 0, Mp * kAstDim * 2));
    HIP_TRY(hipMemset(w.F, 0, Mp * kAstMlp * 2));
    HIP_TRY(hipMemset(w.P, 0, pad128((size_t)nb * kAstPatches) * 256 * 2));
    HIP_TRY(hipMemset(w.Vt, 0, (size_t)nb * kAstDim * kAstKeysPad * 2));
    HIP_TRY(hipMemset(w.QK, 0, Mp * 2 * kAstDim * 2));
    w.cap = nb;
    return 0;
}
int ensure_fbank(amuse_audio_ctx* c, int nb) {
    if (c->fbank_cap >= nb) return 0;
    if (c->fbank) HIP_TRY(hipFree(c->fbank));
    c->fbank = nullptr; c->fbank_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->fbank, (size_t)nb * kAstFrames * kAstMel * 4));
    c->fbank_cap = nb;
    return 0;
}
int ensure_side_streams(amuse_audio_ctx* c) {
    if (c->ev_fork) return 0;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    return 0;
}

// one encoder over nb <= cap clips whose fbanks are at `fbank`
int run_encoder(const amuse_audio_ctx* c, const Workspace& w, const Encoder& E, const float* fbank, int nb, float* feat_out,
                float* hidden_out, int tap_block, hipStream_t st) {
    const int M = nb * kAstRows;   // a clip owns 1216 rows: 1214 tokens + 2 pad rows
    HIP_TRY(launch_im2col(fbank, w.P, nb, st));
    GemmArgs g{};
    g.A = w.P; g.W = E.patch_w; g.bias = E.patch_b; g.M = nb * kAstPatches; g.N = kAstDim; g.K = 256;
    g.out_f32 = w.X; g.pos = E.pos;
    HIP_TRY(launch_gemm(g, EPI_PATCH, st));
    HIP_TRY(launch_ast_tokens(E.cls, E.dist, E.pos, w.X, nb, st));
    for (int l = 0; l < kAstLayers; ++l) {
        const Block& b = E.blk[l];
        HIP_TRY(launch_ln_bf16(w.X, b.n1w, b.n1b, 1e-6f, w.H, M, st));
        g = GemmArgs{};
        g.A = w.H; g.W = b.qkv_w; g.bias = b.qkv_b; g.M = M; g.N = 3 * kAstDim; g.K = kAstDim; g.out_bf16 = w.QK; g.vt = w.Vt;
        HIP_TRY(launch_gemm(g, EPI_QKV, st));
        HIP_TRY(launch_ast_attn(w.QK, w.Vt, w.O, nb, st));
        g = GemmArgs{};
        g.A = w.O; g.W = b.proj_w; g.bias = b.proj_b; g.M = M; g.N = kAstDim; g.K = kAstDim; g.out_f32 = w.X;
        HIP_TRY(launch_gemm(g, EPI_RESID_F32, st));
        HIP_TRY(launch_ln_bf16(w.X, b.n2w, b.n2b, 1e-6f, w.H, M, st));
        g = GemmArgs{};
        g.A = w.H; g.W = b.fc1_w; g.bias = b.fc1_b; g.M = M; g.N = kAstMlp; g.K = kAstDim; g.out_bf16 = w.F;
        HIP_TRY(launch_gemm(g, EPI_GELU_BF16, st));
        g = GemmArgs{};
        g.A = w.F; g.W = b.fc2_w; g.bias = b.fc2_b; g.M = M; g.N = kAstDim; g.K = kAstMlp; g.out_f32 = w.X;
        HIP_TRY(launch_gemm(g, EPI_RESID_F32, st));
        if (hidden_out && l == tap_block)
            HIP_TRY(launch_untile_f32(w.X, hidden_out, M, kAstDim, kAstRows, kAstTokens, st));
    }
    HIP_TRY(launch_ast_pool(w.X, E.norm_w, E.norm_b, c->frame_based, w.pooled, nb, st));
    HIP_TRY(launch_ast_head(w.pooled, c->frame_based, E.fh_ln_w, E.fh_ln_b, E.fh_w, E.fh_b, feat_out, nb, st));
    return 0;
}

}  // namespace

extern "C" {

amuse_audio_ctx* amuse_audio_create(int device, const float* con_params, const float* emo_params, const float* sty_params,
                                    size_t n_each, const float* mel_banks, const float* window, float norm_mean,
                                    float norm_std, int frame_based_feats) {
    if (!con_params || !emo_params || !sty_params || !mel_banks || !window) {
        failf(AMUSE_EINVAL, "NULL argument%s");
        return nullptr;
    }
    if (n_each != AMUSE_AST_PARAMS) {
        failf(AMUSE_EINVAL, "%sparameter count mismatch: %ld per encoder (want %ld)", "", (long)n_each, (long)AMUSE_AST_PARAMS);
        return nullptr;
    }
    if (!(norm_std > 0.f)) {
        failf(AMUSE_EINVAL, "norm_std must be positive%s");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        failf(AMUSE_EHIP, "hipSetDevice(%s%ld) failed", "", device);
        return nullptr;
    }
    amuse_audio_ctx* c = new amuse_audio_ctx();
    c->device = device;
    c->frame_based = frame_based_feats ? 1 : 0;
    c->norm_mean = norm_mean;
    c->norm_std = norm_std;
    const float* ps[3] = {con_params, emo_params, sty_params};
    // device image of the filter bank: transposed (a bin's 128 weights contiguous) + each filter's support
    std::vector<float> mel_t((size_t)257 * kAstMel);
    std::vector<int> mel_rng(2 * kAstMel);
    for (int m = 0; m < kAstMel; ++m) {
        int k0 = 257, k1 = 0;
        for (int k = 0; k < 257; ++k) {
            const float w = mel_banks[(size_t)m * 257 + k];
            mel_t[(size_t)k * kAstMel + m] = w;
            if (w != 0.f) { if (k < k0) k0 = k; k1 = k + 1; }
        }
        mel_rng[2 * m] = k0 < k1 ? k0 : 0;
        mel_rng[2 * m + 1] = k0 < k1 ? k1 : 0;
    }
    int rc = up_f32(c, &c->melw, mel_t.data(), mel_t.size()) || up_f32(c, &c->window, window, 400);
    if (!rc) {
        float* rng = nullptr;   // (ints travel through the float uploader bit for bit)
        static_assert(sizeof(int) == sizeof(float), "");
        rc = up_f32(c, &rng, reinterpret_cast<const float*>(mel_rng.data()), mel_rng.size());
        c->mel_range = reinterpret_cast<int*>(rng);
    }
    for (int e = 0; e < 3 && !rc; ++e) rc = build_encoder(c, c->enc[e], ps[e]);
    if (rc) {
        amuse_audio_destroy(c);
        return nullptr;
    }
    return c;
}

void amuse_audio_destroy(amuse_audio_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void* p : c->owned) (void)hipFree(p);
    for (Workspace& w : c->ws) free_ws(w);
    if (c->fbank) (void)hipFree(c->fbank);
    for (int i = 0; i < 2; ++i) {
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    delete c;
}

int amuse_audio_fbank(amuse_audio_ctx* c, const float* waves, int n_samples, int B, float* fbank_out, void* stream) {
    if (!c || !waves || !fbank_out) return failf(AMUSE_EINVAL, "NULL argument%s");
    if (B < 1 || n_samples < 1) return failf(AMUSE_EINVAL, "%sB and n_samples must be >= 1 (got %ld, %ld)", "", B, n_samples);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_fbank(waves, n_samples, B, c->window, c->melw, c->mel_range, c->norm_mean, c->norm_std, fbank_out, (hipStream_t)stream));
    return 0;
}

int amuse_audio_encode(amuse_audio_ctx* c, int which, const float* fbank, int B, float* feat_out, float* hidden_out,
                       int tap_block, void* stream) {
    if (!c || !fbank || !feat_out) return failf(AMUSE_EINVAL, "NULL argument%s");
    if (which < 0 || which > 2) return failf(AMUSE_EINVAL, "%sencoder index %ld not in 0..2", "", which);
    if (B < 1) return failf(AMUSE_EINVAL, "%sB must be >= 1, got %ld", "", B);
    if (hidden_out && (tap_block < 0 || tap_block >= kAstLayers)) return failf(AMUSE_EINVAL, "%stap_block %ld not in 0..11", "", tap_block);
    HIP_TRY(hipSetDevice(c->device));
    const int chunk = B < kChunk ? B : kChunk;
    if (int e = ensure_ws(c->ws[0], chunk)) return e;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        if (int e = run_encoder(c, c->ws[0], c->enc[which], fbank + (size_t)b0 * kAstFrames * kAstMel, nb, feat_out + (size_t)b0 * kAstFeat,
                                hidden_out ? hidden_out + (size_t)b0 * kAstTokens * kAstDim : nullptr, tap_block, (hipStream_t)stream))
            return e;
    }
    return 0;
}

int amuse_audio_features(amuse_audio_ctx* c, const float* waves, int n_samples, int B, float* con_out, float* emo_out,
                         float* sty_out, void* stream) {
    if (!c || !waves) return failf(AMUSE_EINVAL, "NULL argument%s");
    if (B < 1 || n_samples < 1) return failf(AMUSE_EINVAL, "%sB and n_samples must be >= 1 (got %ld, %ld)", "", B, n_samples);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int chunk = B < kChunk ? B : kChunk;
    float* outs[3] = {con_out, emo_out, sty_out};
    if (int e = ensure_fbank(c, chunk)) return e;
    if (int e = ensure_side_streams(c)) return e;
    for (int e = 0; e < 3; ++e)
        if (outs[e])
            if (int rc = ensure_ws(c->ws[e], chunk)) return rc;
    // per chunk: fbank on `st`, then a fork-join over two side streams (stream-ordered with `st` through events, so the
    // call stays asynchronous and capturable): encoder e on stream e with workspace e
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        HIP_TRY(launch_fbank(waves + (size_t)b0 * n_samples, n_samples, nb, c->window, c->melw, c->mel_range, c->norm_mean, c->norm_std, c->fbank, st));
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        for (int e = 1; e < 3; ++e) {
            if (!outs[e]) continue;
            HIP_TRY(hipStreamWaitEvent(c->side[e - 1], c->ev_fork, 0));
            if (int rc = run_encoder(c, c->ws[e], c->enc[e], c->fbank, nb, outs[e] + (size_t)b0 * kAstFeat, nullptr, 0, c->side[e - 1])) return rc;
            HIP_TRY(hipEventRecord(c->ev_join[e - 1], c->side[e - 1]));
        }
        if (outs[0])
            if (int rc = run_encoder(c, c->ws[0], c->enc[0], c->fbank, nb, outs[0] + (size_t)b0 * kAstFeat, nullptr, 0, st)) return rc;
        for (int e = 1; e < 3; ++e)
            if (outs[e]) HIP_TRY(hipStreamWaitEvent(st, c->ev_join[e - 1], 0));   // (also: the next chunk's fbank overwrites c->fbank)
    }
    return 0;
}

// GEMM in isolation (tools/gpu_gemm_bench.py, tests): C = A . W^T + bias with epilogue 0 (bf16 out) or 3 (fp32 out);
// A dev bf16 TILE-MAJOR [M padded to 128][K] (amuse_debug_tile), W dev bf16 in the kernel's packed fragment order
// (GemmArgs::W), out tile-major [M padded to 128][N]
int amuse_debug_gemm(const void* A, const void* W, const float* bias, int M, int N, int K, int epi, void* out, void* stream) {
    if (!A || !W || !bias || !out) return failf(AMUSE_EINVAL, "NULL argument%s");
    if (M < 1 || N % kGemmTN || K % 64 || (epi != EPI_BF16 && epi != EPI_F32)) return failf(AMUSE_EINVAL, "%sbad GEMM shape / epilogue (N %ld K %ld)", "", N, K);
    GemmArgs g{};
    g.A = (const unsigned short*)A; g.W = (const unsigned short*)W; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out_bf16 = (unsigned short*)out; g.out_f32 = (float*)out;
    HIP_TRY(launch_gemm(g, epi, (hipStream_t)stream));
    return 0;
}

// what 0: bf16 row-major [M][F] -> tile-major [M padded to 128][F] (pad rows zeroed); 1: bf16 tile-major -> row-major [M][F];
// 2: fp32 tile-major -> row-major [M][F]
int amuse_debug_tile(const void* src, void* dst, int M, int F, int what, void* stream) {
    if (!src || !dst) return failf(AMUSE_EINVAL, "NULL argument%s");
    if (M < 1 || F < 32 || F % 32 || what < 0 || what > 2) return failf(AMUSE_EINVAL, "%sbad shape / direction (F %ld what %ld)", "", F, what);
    hipStream_t st = (hipStream_t)stream;
    if (what == 0) HIP_TRY(launch_tile_bf16((const unsigned short*)src, (unsigned short*)dst, M, F, st));
    else if (what == 1) HIP_TRY(launch_untile_bf16((const unsigned short*)src, (unsigned short*)dst, M, F, st));
    else HIP_TRY(launch_untile_f32((const float*)src, (float*)dst, M, F, M, M, st));
    return 0;
}

}  // extern "C"
