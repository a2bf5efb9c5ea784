// The launch plan, one source of truth: for EVERY clip count 1..8192, all four precisions and 3 / 4 / 5 tokens per clip, what amuse_sample / amuse_vae_decode /
// amuse_vae_encode / a pose-space Denoiser step ACTUALLY take on AUTO (amuse_debug_last_plan, read after the call) is exactly what amuse_plan returns for that
// clip count - and a pin (amuse_set_clips_per_group / amuse_set_decode_path with the plan of ANOTHER job size, what a shard does) overrides the call's own choice.
// Runs on the library's host code with the stubbed runtime of hip_stub.cpp (launches are no-ops, "device" memory is host memory): no GPU.
// Large output arrays are never touched by host code, so they are passed as one small dummy buffer (AddressSanitizer would flag a host access).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/amuse_hip.h"

#define REQUIRE(cond)                                                                      \
    do {                                                                                   \
        if (!(cond)) { printf("FAILED %s:%d: %s (%s) at B=%d prec=%d\n", __FILE__, __LINE__, #cond, amuse_last_error(), B, prec); return 1; } \
    } while (0)

int main() {
    int B = 0, prec = 0;
    std::vector<float> den(AMUSE_DENOISER_PARAMS, 0.01f), pri(AMUSE_PRIOR_PARAMS, 0.01f);
    amuse_ctx* c = amuse_create(0, den.data(), den.size(), pri.data(), pri.size());
    REQUIRE(c != nullptr);
    int ts[1] = {0};
    float coef[8] = {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f};
    amuse_schedule s{1, ts, coef, nullptr};
    REQUIRE(amuse_set_schedule(c, &s, nullptr) == 0);
    const int BMAX = 8192;
    std::vector<float> cond((size_t)BMAX * 256, 0.f), lat((size_t)BMAX * 128, 0.f), small(4096, 0.f);
    float* big = small.data();   // poses / trans / feats / pose-space states: never read or written on the host
    long checked = 0;
    for (B = 1; B <= BMAX; ++B)
        for (prec = AMUSE_PREC_F32; prec <= AMUSE_PREC_F16; ++prec) {
            int g = -1, dp = -1, ep = -1, sp = -1, lg = -1, ldp = -1, lep = -1;
            for (int tokens = 3; tokens <= 5; ++tokens) {
                REQUIRE(amuse_plan(AMUSE_ARCH_ENC, prec, B, tokens, &g, &dp, &ep, &sp) == 0);
                REQUIRE(amuse_sample(c, cond.data(), tokens >= 4 ? cond.data() : nullptr, tokens == 5 ? cond.data() : nullptr, B, prec, 7, 0, nullptr, nullptr, lat.data(), nullptr, nullptr) == 0);
                REQUIRE(amuse_debug_last_plan(c, &lg, nullptr, nullptr, nullptr) == 0);
                REQUIRE(lg == g && g >= 1 && g <= 16 / tokens && sp == AMUSE_DECODE_STAGED);
                ++checked;
            }
            REQUIRE(amuse_vae_decode(c, lat.data(), nullptr, B, prec, AMUSE_QUAT_P3D, nullptr, big, big, nullptr) == 0);
            REQUIRE(amuse_vae_encode(c, big, nullptr, B, prec, nullptr, lat.data(), nullptr, nullptr, nullptr) == 0);
            REQUIRE(amuse_debug_last_plan(c, nullptr, &ldp, &lep, nullptr) == 0);
            REQUIRE(ldp == dp && lep == ep);
            REQUIRE((prec == AMUSE_PREC_F32) == (dp == AMUSE_DECODE_STAGED && B >= 64) || B < 64);
            checked += 2;
            if (B % 97 == 0) {   // a shard of a 4096-clip job: the job's plan pinned, the shard's own count must not matter
                int jg, jdp;
                REQUIRE(amuse_plan(AMUSE_ARCH_ENC, AMUSE_PREC_F32X, 4096, 5, &jg, &jdp, nullptr, nullptr) == 0);
                REQUIRE(jg == 3 && jdp == AMUSE_DECODE_CLIP);
                REQUIRE(amuse_set_clips_per_group(c, jg) == 0 && amuse_set_decode_path(c, jdp) == 0);
                REQUIRE(amuse_sample(c, cond.data(), cond.data(), cond.data(), B, prec, 7, 0, nullptr, nullptr, lat.data(), nullptr, nullptr) == 0);
                REQUIRE(amuse_vae_decode(c, lat.data(), nullptr, B, prec, AMUSE_QUAT_P3D, nullptr, big, big, nullptr) == 0);
                REQUIRE(amuse_debug_last_plan(c, &lg, &ldp, nullptr, nullptr) == 0);
                const int want = prec == AMUSE_PREC_F32 ? AMUSE_DECODE_STAGED : prec == AMUSE_PREC_F32X ? AMUSE_DECODE_CLIP : AMUSE_DECODE_FUSED;
                REQUIRE(lg == 3 && ldp == want);
                REQUIRE(amuse_set_clips_per_group(c, 0) == 0 && amuse_set_decode_path(c, AMUSE_DECODE_AUTO) == 0);
            }
        }
    amuse_destroy(c);
    // the pose-space trans_enc Denoiser (S = 304 rows per clip and step): the step's kernel family
    std::vector<float> dv(amuse_denoiser_param_count(AMUSE_ARCH_ENC_POSE), 0.01f);
    amuse_ctx* v = amuse_create_arch(0, AMUSE_ARCH_ENC_POSE, dv.data(), dv.size(), nullptr, 0);
    B = 0;
    REQUIRE(v != nullptr);
    REQUIRE(amuse_set_schedule(v, &s, nullptr) == 0);
    for (B = 1; B <= BMAX; B += (B < 1100 ? 1 : 37))
        for (prec = AMUSE_PREC_F32; prec <= AMUSE_PREC_F16; ++prec) {
            int g = -1, sp = -1, lsp = -1;
            REQUIRE(amuse_plan(AMUSE_ARCH_ENC_POSE, prec, B, 5, &g, nullptr, nullptr, &sp) == 0);
            REQUIRE(amuse_denoise_step_pose(v, big, 5, cond.data(), cond.data(), cond.data(), nullptr, B, prec, big, nullptr) == 0);
            REQUIRE(amuse_debug_last_plan(v, nullptr, nullptr, nullptr, &lsp) == 0);
            REQUIRE(g == 1 && lsp == sp);
            ++checked;
        }
    amuse_destroy(v);
    B = 1; prec = 0;
    int x;
    REQUIRE(amuse_plan(9, 0, 1, 5, &x, &x, &x, &x) != 0 && amuse_plan(0, 7, 1, 5, &x, &x, &x, &x) != 0 && amuse_plan(0, 0, 0, 5, &x, &x, &x, &x) != 0 &&
            amuse_plan(0, 0, 1, 2, &x, &x, &x, &x) != 0 && amuse_plan(0, 0, 1, 5, nullptr, nullptr, nullptr, nullptr) == 0);
    printf("PLAN SWEEP OK (%ld calls checked)\n", checked);
    return 0;
}
