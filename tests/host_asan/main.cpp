// Drives the library's host code (amuse_api.hip, amuse_variants.hip, amuse_audio_api.hip) through the C ABI with the stubbed runtime of hip_stub.cpp
// under AddressSanitizer + UBSan: context construction and weight packing (both precisions, encoder streams), re-packing in
// place, schedules, every entry point's argument checks and workspace growth, the audio context (weight images of three
// encoders, workspaces for several batch sizes incl. chunking), teardown without leaks of "device" memory.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/amuse_hip.h"

long amuse_stub_live_allocations();

#define REQUIRE(cond)                                                                      \
    do {                                                                                   \
        if (!(cond)) { printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, amuse_last_error()); return 1; } \
    } while (0)

static void fill(std::vector<float>& v, uint32_t seed, float scale) {
    uint32_t s = seed;
    for (float& x : v) { s = s * 1664525u + 1013904223u; x = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
}

int main(int argc, char** argv) {
    const bool with_audio = argc < 2 || atoi(argv[1]) != 0;
    std::vector<float> den(AMUSE_DENOISER_PARAMS), pri(AMUSE_PRIOR_PARAMS);
    fill(den, 1, 0.2f);
    fill(pri, 2, 0.2f);
    REQUIRE(amuse_create(0, den.data(), den.size() - 1, pri.data(), pri.size()) == nullptr);   // wrong count is refused
    REQUIRE(amuse_create(0, nullptr, den.size(), pri.data(), pri.size()) == nullptr);
    amuse_ctx* c = amuse_create(0, den.data(), den.size(), pri.data(), pri.size());
    REQUIRE(c != nullptr);
    REQUIRE(amuse_abi_version() == AMUSE_ABI_VERSION);
    // schedules of 1, 50 and 1000 steps (contents are arbitrary: the kernels are stubbed)
    for (int T : {1, 50, AMUSE_MAX_STEPS}) {
        std::vector<int> ts(T);
        std::vector<float> coef((size_t)T * 8, 0.5f), freqs(128, 0.1f);
        for (int i = 0; i < T; ++i) ts[i] = T - 1 - i;
        amuse_schedule s{T, ts.data(), coef.data(), (T == 50) ? freqs.data() : nullptr};
        REQUIRE(amuse_set_schedule(c, &s, nullptr) == 0);
    }
    {
        amuse_schedule bad{0, nullptr, nullptr, nullptr};
        REQUIRE(amuse_set_schedule(c, &bad, nullptr) != 0);
        bad.n_steps = AMUSE_MAX_STEPS + 1;
        REQUIRE(amuse_set_schedule(c, &bad, nullptr) != 0);
    }
    // "device" buffers are host memory here; sizes as the header documents them
    const int BMAX = 700;   // > the VAE chunk of 512 and > every clips-per-tile boundary
    std::vector<float> cond((size_t)BMAX * 256), lat((size_t)BMAX * 128), traj((size_t)AMUSE_MAX_STEPS * BMAX * 128);
    std::vector<float> poses((size_t)BMAX * 300 * 55 * 3), trans((size_t)BMAX * 300 * 3), feats((size_t)BMAX * 300 * 333);
    std::vector<int> lengths(BMAX, 300);
    lengths[3] = 1; lengths[5] = 299;
    for (int B : {1, 2, 3, 5, 64, 128, 129, 256, 513, BMAX})
        for (int prec : {AMUSE_PREC_F32, AMUSE_PREC_BF16, AMUSE_PREC_F32X, AMUSE_PREC_F16}) {
            for (int g = 0; g <= 5; ++g) {
                REQUIRE(amuse_set_clips_per_group(c, g) == 0);
                REQUIRE(amuse_sample(c, cond.data(), cond.data(), g & 1 ? nullptr : cond.data(), B, prec, 7, 11, nullptr, nullptr, lat.data(), nullptr, nullptr) == 0);
            }
            REQUIRE(amuse_set_clips_per_group(c, 0) == 0);
            REQUIRE(amuse_sample(c, cond.data(), nullptr, nullptr, B, prec, 7, 0, lat.data(), traj.data(), lat.data(), traj.data(), nullptr) == 0);
            REQUIRE(amuse_denoise_step(c, lat.data(), 981, cond.data(), cond.data(), cond.data(), B, prec, lat.data(), nullptr, nullptr) == 0);
            std::vector<int> ts(B, 5);
            std::vector<float> sa(B, 0.9f), sb(B, 0.1f);
            REQUIRE(amuse_diffusion_forward(c, lat.data(), lat.data(), ts.data(), sa.data(), sb.data(), cond.data(), cond.data(), cond.data(), B, prec,
                                            lat.data(), lat.data(), nullptr) == 0);
            for (int path : {AMUSE_DECODE_AUTO, AMUSE_DECODE_STAGED, AMUSE_DECODE_FUSED}) {
                REQUIRE(amuse_set_decode_path(c, path) == 0);
                REQUIRE(amuse_vae_decode(c, lat.data(), B & 1 ? lengths.data() : nullptr, B, prec, AMUSE_QUAT_P3D, feats.data(), poses.data(), trans.data(), nullptr) == 0);
            }
            REQUIRE(amuse_vae_encode(c, feats.data(), lengths.data(), B, prec, lat.data(), lat.data(), lat.data(), lat.data(), nullptr) == 0);
            REQUIRE(amuse_smplx_to_feats(c, poses.data(), trans.data(), B, feats.data(), nullptr) == 0);
            REQUIRE(amuse_diffusion_backward(c, cond.data(), cond.data(), cond.data(), B, prec, AMUSE_QUAT_LEGACY, 1, 2, nullptr, nullptr, lat.data(), poses.data(),
                                             trans.data(), nullptr) == 0);
            REQUIRE(amuse_counter_normal(c, 3, 4, B, 0, 1, lat.data(), nullptr) == 0);
        }
    // argument checks
    REQUIRE(amuse_sample(c, nullptr, nullptr, nullptr, 1, AMUSE_PREC_BF16, 0, 0, nullptr, nullptr, lat.data(), nullptr, nullptr) != 0);
    REQUIRE(amuse_sample(c, cond.data(), nullptr, nullptr, 0, AMUSE_PREC_BF16, 0, 0, nullptr, nullptr, lat.data(), nullptr, nullptr) != 0);
    REQUIRE(amuse_sample(c, cond.data(), nullptr, nullptr, 1, 9, 0, 0, nullptr, nullptr, lat.data(), nullptr, nullptr) != 0);
    REQUIRE(amuse_set_clips_per_group(c, 6) != 0);
    REQUIRE(amuse_set_decode_path(c, 4) != 0);
    REQUIRE(amuse_set_decode_path(c, AMUSE_DECODE_CLIP) == 0 && amuse_set_decode_path(c, AMUSE_DECODE_AUTO) == 0);
    REQUIRE(amuse_vae_encode(c, feats.data(), nullptr, 1, AMUSE_PREC_F32, nullptr, nullptr, nullptr, nullptr, nullptr) != 0);
    // re-packing in place: every subset of streams, either array alone
    fill(den, 3, 0.1f);
    for (int what = 1; what <= AMUSE_UPD_ALL; ++what) {
        const int rc = amuse_update_weights(c, den.data(), den.size(), pri.data(), pri.size(), what, nullptr);
        REQUIRE((what & (AMUSE_UPD_F32 | AMUSE_UPD_BF16 | AMUSE_UPD_F32X | AMUSE_UPD_F16)) ? rc == 0 : rc != 0);   // the encoder streams alone are refused
    }
    REQUIRE(amuse_update_weights(c, den.data(), den.size(), nullptr, 0, AMUSE_UPD_BF16, nullptr) == 0);
    REQUIRE(amuse_update_weights(c, nullptr, 0, pri.data(), pri.size(), AMUSE_UPD_ALL, nullptr) == 0);
    REQUIRE(amuse_update_weights(c, den.data(), 5, nullptr, 0, AMUSE_UPD_ALL, nullptr) != 0);
    // the device re-pack: builds the gather maps (three probe runs of both builders through the capture hook) on its first call
    for (int what : {2, 7, 1, 6, 8, 15, 16, 31}) REQUIRE(amuse_update_weights_device(c, den.data(), pri.data(), what, nullptr) == 0);
    REQUIRE(amuse_update_weights_device(c, den.data(), nullptr, 2, nullptr) == 0);
    REQUIRE(amuse_update_weights_device(c, nullptr, nullptr, 2, nullptr) != 0);
    REQUIRE(amuse_update_weights_device(c, den.data(), pri.data(), 4, nullptr) != 0);
    amuse_destroy(c);
    amuse_destroy(nullptr);
    // the Denoiser variants (amuse_variants.hip): construction + packing, schedule, sampling / single steps on both paths of the
    // pose-space step (staged below 64 clips, fused from there), re-packing, the argument checks of the variant-only entry points
    REQUIRE(amuse_denoiser_param_count(AMUSE_ARCH_ENC) == AMUSE_DENOISER_PARAMS && amuse_denoiser_param_count(7) == 0);
    REQUIRE(amuse_create_arch(0, 4, den.data(), den.size(), pri.data(), pri.size()) == nullptr);
    for (int arch : {AMUSE_ARCH_DEC, AMUSE_ARCH_ENC_POSE, AMUSE_ARCH_DEC_POSE}) {
        const bool pose = arch != AMUSE_ARCH_DEC;
        std::vector<float> dv(amuse_denoiser_param_count(arch));
        fill(dv, 10 + arch, 0.2f);
        REQUIRE(amuse_create_arch(0, arch, dv.data(), dv.size() - 1, pri.data(), pri.size()) == nullptr);
        amuse_ctx* v = amuse_create_arch(0, arch, dv.data(), dv.size(), pose ? nullptr : pri.data(), pose ? 0 : pri.size());
        REQUIRE(v != nullptr);
        REQUIRE(amuse_arch(v) == arch);
        const size_t sd = amuse_state_dim(v);
        REQUIRE(sd == (pose ? (size_t)AMUSE_POSE_STATE : 128u));
        const int T = 3, VB = 70;
        std::vector<int> ts{2, 1, 0};
        std::vector<float> coef((size_t)T * 8, 0.5f);
        amuse_schedule s{T, ts.data(), coef.data(), nullptr};
        REQUIRE(amuse_set_schedule(v, &s, nullptr) == 0);
        std::vector<float> x((size_t)VB * sd), out((size_t)VB * sd), vtraj((size_t)T * VB * sd), vnoise((size_t)T * VB * sd);
        for (int B : {1, 17, 64, VB})
            for (int prec : {AMUSE_PREC_F32, AMUSE_PREC_BF16, AMUSE_PREC_F32X, AMUSE_PREC_F16}) {
                REQUIRE(amuse_sample(v, cond.data(), cond.data(), B & 1 ? nullptr : cond.data(), B, prec, 7, 11, nullptr, nullptr, out.data(), nullptr, nullptr) == 0);
                REQUIRE(amuse_sample(v, cond.data(), nullptr, nullptr, B, prec, 7, 0, x.data(), vnoise.data(), out.data(), vtraj.data(), nullptr) == 0);
                REQUIRE(amuse_denoise_step(v, x.data(), 981, cond.data(), cond.data(), cond.data(), B, prec, out.data(), nullptr, nullptr) == 0);
                REQUIRE(amuse_counter_normal(v, 3, 4, B, 0, 1, out.data(), nullptr) == 0);
                {
                    std::vector<int> tsb(B, 2);
                    std::vector<float> sa(B, 0.9f), sb(B, 0.1f);
                    REQUIRE(amuse_diffusion_forward(v, x.data(), vnoise.data(), tsb.data(), sa.data(), sb.data(), cond.data(), cond.data(), cond.data(), B, prec,
                                                    out.data(), vtraj.data(), nullptr) == 0);
                    REQUIRE(amuse_diffusion_backward(v, cond.data(), cond.data(), cond.data(), B, prec, AMUSE_QUAT_P3D, 1, 2, nullptr, nullptr, out.data(), poses.data(),
                                                     trans.data(), nullptr) == 0);
                    REQUIRE(amuse_update_weights_device(v, dv.data(), nullptr, AMUSE_UPD_BF16, nullptr) != 0);   // shipped configuration only
                }
                if (pose) {
                    REQUIRE(amuse_denoise_step_pose(v, x.data(), 5, cond.data(), nullptr, cond.data(), B & 1 ? lengths.data() : nullptr, B, prec, out.data(), nullptr) == 0);
                    REQUIRE(amuse_vae_decode(v, lat.data(), nullptr, B, prec, AMUSE_QUAT_P3D, feats.data(), poses.data(), trans.data(), nullptr) != 0);   // no prior
                } else {
                    REQUIRE(amuse_denoise_step_pose(v, x.data(), 5, cond.data(), nullptr, nullptr, nullptr, B, prec, out.data(), nullptr) != 0);
                    REQUIRE(amuse_vae_decode(v, out.data(), nullptr, B, prec, AMUSE_QUAT_P3D, feats.data(), poses.data(), trans.data(), nullptr) == 0);
                }
            }
        REQUIRE(amuse_feats_to_smplx(v, feats.data(), 5, AMUSE_QUAT_P3D, poses.data(), trans.data(), nullptr) == 0);
        REQUIRE(amuse_feats_to_smplx(v, nullptr, 5, AMUSE_QUAT_P3D, poses.data(), trans.data(), nullptr) != 0);
        REQUIRE(amuse_feats_to_smplx(v, feats.data(), 0, AMUSE_QUAT_P3D, poses.data(), trans.data(), nullptr) != 0);
        for (int mask : {1, 0}) REQUIRE(amuse_debug_set_ablation(v, mask) == 0);
        fill(dv, 20 + arch, 0.1f);
        for (int what : {AMUSE_UPD_F32, AMUSE_UPD_BF16, AMUSE_UPD_F32X, AMUSE_UPD_F16, AMUSE_UPD_ALL})
            REQUIRE(amuse_update_weights(v, dv.data(), dv.size(), nullptr, 0, what, nullptr) == 0);
        REQUIRE(amuse_update_weights(v, dv.data(), dv.size() - 1, nullptr, 0, AMUSE_UPD_ALL, nullptr) != 0);
        amuse_destroy(v);
    }
    if (with_audio) {
        std::vector<float> ast(AMUSE_AST_PARAMS), mel((size_t)128 * 257), win(400, 0.5f);
        fill(ast, 5, 0.05f);
        fill(mel, 6, 1.0f);
        for (size_t i = 0; i < mel.size(); ++i) mel[i] = (i % 257) / 2 == i / 257 ? 1.0f : 0.0f;   // a sparse bank with supports
        REQUIRE(amuse_audio_create(0, ast.data(), ast.data(), ast.data(), AMUSE_AST_PARAMS - 1, mel.data(), win.data(), -4.f, 4.5f, 1) == nullptr);
        REQUIRE(amuse_audio_create(0, ast.data(), ast.data(), ast.data(), AMUSE_AST_PARAMS, mel.data(), win.data(), -4.f, 0.f, 1) == nullptr);
        amuse_audio_ctx* a = amuse_audio_create(0, ast.data(), ast.data(), ast.data(), AMUSE_AST_PARAMS, mel.data(), win.data(), -4.f, 4.5f, 1);
        REQUIRE(a != nullptr);
        const int NB = 35;
        std::vector<float> wav((size_t)NB * 16000), fb((size_t)NB * 1024 * 128), f256((size_t)NB * 256), hid((size_t)NB * 1214 * 768);
        for (int B : {1, 3, 33, NB, 2}) {
            REQUIRE(amuse_audio_fbank(a, wav.data(), 16000, B, fb.data(), nullptr) == 0);
            for (int which = 0; which < 3; ++which) REQUIRE(amuse_audio_encode(a, which, fb.data(), B, f256.data(), which == 1 ? hid.data() : nullptr, 11, nullptr) == 0);
            REQUIRE(amuse_audio_features(a, wav.data(), 16000, B, f256.data(), B & 1 ? nullptr : f256.data(), f256.data(), nullptr) == 0);
        }
        REQUIRE(amuse_audio_encode(a, 3, fb.data(), 1, f256.data(), nullptr, 0, nullptr) != 0);
        REQUIRE(amuse_audio_encode(a, 0, fb.data(), 1, f256.data(), hid.data(), 12, nullptr) != 0);
        REQUIRE(amuse_audio_features(a, wav.data(), 0, 1, f256.data(), nullptr, nullptr, nullptr) != 0);
        REQUIRE(amuse_debug_gemm(fb.data(), fb.data(), f256.data(), 128, 256, 64, 0, hid.data(), nullptr) == 0);
        REQUIRE(amuse_debug_gemm(fb.data(), fb.data(), f256.data(), 128, 128, 64, 0, hid.data(), nullptr) != 0);
        REQUIRE(amuse_debug_tile(fb.data(), hid.data(), 100, 64, 0, nullptr) == 0);
        REQUIRE(amuse_debug_tile(fb.data(), hid.data(), 100, 48, 0, nullptr) != 0);
        amuse_audio_destroy(a);
        amuse_audio_destroy(nullptr);
    }
    REQUIRE(amuse_stub_live_allocations() == 0);
    printf("HOST ASAN OK (audio %d)\n", (int)with_audio);
    return 0;
}
