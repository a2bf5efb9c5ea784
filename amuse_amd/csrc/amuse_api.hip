// C ABI of libamuse_hip.so (include/amuse_hip.h): context, weight packing into MFMA-fragment
// streams, workspace, and the launch sequences.  Host code only - kernels live in k_*.hip.
#include "amuse_host.hpp"
#include "amuse_variants.hpp"

namespace {
thread_local char g_err[512] = "";
}  // namespace
int amuse_failf(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// the same error slot for the library's other translation units (amuse_audio_api.hip)
__attribute__((visibility("hidden"))) int amuse_fail_msg(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}


namespace {
constexpr int kVaeChunk = 512;
constexpr int kEncRows = kFrames + 2;  // encoder sequence: 2 distribution tokens + 300 frames
constexpr size_t kVaeFloatsPerClip = (size_t)kEncRows * kD * (1 + 3 + 1 + 4) + kLayers * kD;  // x, qkv, o, skip, ca | stats

constexpr int kUpdBit[4] = {AMUSE_UPD_F32, AMUSE_UPD_BF16, AMUSE_UPD_F32X, AMUSE_UPD_F16};   // per PREC_* index

int build_denoiser(amuse_ctx* c, const float* den, int what = AMUSE_UPD_ALL) {
    static const ParamIndex DI = denoiser_index();
    const Params D{DI, den};
    // ---- weight stream of the 4-wave kernel (k_sampler.hip: the fp32 parity mode; every other mode samples on an 8-wave kernel): [wave][per-step units]
    for (const int prec : {PREC_F32}) {
        if (!(what & kUpdBit[prec])) continue;   // amuse_update_weights: only the requested precisions are re-packed
        std::vector<uint4> all;
        size_t per_wave = 0;
        for (int w = 0; w < 4; ++w) {
            std::vector<uint4> s;
            for (int b = 0; b < 9; ++b) {
                const std::string p = blk_name("encoder", b);
                if (b >= 5) {
                    pack_skiplin(s, prec, D, "encoder", b - 5, w);
                    s.insert(s.end(), (size_t)skip_pad_units(prec) * 64, uint4{0, 0, 0, 0});  // ring alignment
                }
                pack_qkv(s, prec, D.get(p + ".self_attn.in_proj_weight"), w, true);
                pack_outproj_ffn_quarters(s, prec, D, p, w);
            }
            if (w == 0) per_wave = s.size();
            else if (s.size() != per_wave) return fail(AMUSE_ESTATE, "internal: uneven denoiser wave streams");
            all.insert(all.end(), s.begin(), s.end());
            all.insert(all.end(), s.begin(), s.begin() + (size_t)kRing * 64);  // ring wrap: tail = head
        }
        c->den_wave_units[prec] = (uint32_t)(per_wave / 64);
        if (upload(&c->den_w[prec], all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    for (const int p16 : {PREC_BF16, PREC_F16}) {   // 8-wave throughput kernel (bf16 / fp16 operands): wave w8 = 4 s + h; A waves (s = 0) carry head h + FFN quarters 0,1, B waves quarters 2,3
        if (!(what & kUpdBit[p16])) continue;
        std::vector<uint4> all;
        for (int w8 = 0; w8 < 8; ++w8) {
            const int h = w8 & 3, sgrp = w8 >> 2;
            std::vector<uint4> s;
            // per block, in issue order (k_sampler8.hip).  A: out_proj | in_proj q,k | v (ahead of an output block: skip
            // linear, skip-input half | q,k | v | out_proj), then FFN quarters 0,1 as F1a F1b F2a F2b.  B: [skip linear,
            // x half], then FFN quarters 2,3.
            for (int b = 0; b < 9; ++b) {
                const std::string p = blk_name("encoder", b);
                auto f1 = [&](int q) { const int h0 = 8 * h + 2 * q; pack_gemm(s, p16, D.get(p + ".linear1.weight"), 512, 128, {h0, h0 + 1}, range(0, 8)); };
                auto f2 = [&](int q) { const int h0 = 8 * h + 2 * q; pack_gemm(s, p16, D.get(p + ".linear2.weight"), 128, 512, range(0, 8), {h0, h0 + 1}); };
                const int qa = 2 * sgrp, qb = 2 * sgrp + 1;
                const float* wskip = b >= 5 ? D.get("encoder.linear_blocks." + std::to_string(b - 5) + ".weight") : nullptr;
                if (sgrp == 0) {
                    // a block's group of 32: 8 leading units, then q, k | v.  The leading 8 are out_proj - or, ahead of an
                    // output block, the skip-input half (k-tiles 8..15 of cat(x, skip)) of the skip linear for output
                    // tiles 2h, 2h+1, with out_proj following as a group of its own
                    const auto outproj = [&] { pack_gemm(s, p16, D.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * h, 2 * h + 1}); };
                    if (b >= 5) pack_gemm(s, p16, wskip, 128, 256, {2 * h, 2 * h + 1}, range(8, 16));
                    else outproj();
                    pack_qkv(s, p16, D.get(p + ".self_attn.in_proj_weight"), h, true);
                    if (b >= 5) outproj();
                } else if (b >= 5) {
                    // the x half (k-tiles 0..7) of the same two output tiles
                    pack_gemm(s, p16, wskip, 128, 256, {2 * h, 2 * h + 1}, range(0, 8));
                }
                f1(qa); f1(qb); f2(qa); f2(qb);
            }
            uint32_t& units = c->den_w8_units[sgrp];
            if (h == 0) units = (uint32_t)(s.size() / 64);
            else if (s.size() / 64 != units) return fail(AMUSE_ESTATE, "internal: uneven 8-wave denoiser streams");
            all.insert(all.end(), s.begin(), s.end());
            if (sgrp == 0) all.insert(all.end(), s.begin(), s.begin() + (size_t)kRing8 * 64);  // ring wrap: tail = head
        }
        if (upload(p16 == PREC_BF16 ? &c->den_w8 : &c->den_w8h, all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    if (what & AMUSE_UPD_F32X) {   // 8-wave fp32x kernel: the same roles, split-fp16 units (two per 16 x 32 weight tile: hi, lo)
        std::vector<uint4> all;
        for (int w8 = 0; w8 < 8; ++w8) {
            const int h = w8 & 3, sgrp = w8 >> 2;
            std::vector<uint4> s;
            // per block, in issue order (k_sampler8x.hip).  A: lead (v - or, ahead of an output block, the skip-input half of the
            // skip linear for output tiles 2h, 2h+1) | q,k for k-pairs 0,1 | [v, output blocks] | q,k for k-pairs 2,3 | out_proj
            // | F1a F1b F2a F2b (FFN quarters 0,1).  B: [skip linear, x half], F1a F1b F2a F2b (quarters 2,3).
            for (int b = 0; b < 9; ++b) {
                const std::string p = blk_name("encoder", b);
                auto f1 = [&](int q) { const int h0 = 8 * h + 2 * q; pack_gemm(s, PREC_F16X2, D.get(p + ".linear1.weight"), 512, 128, {h0, h0 + 1}, range(0, 8)); };
                auto f2 = [&](int q) { const int h0 = 8 * h + 2 * q; pack_gemm(s, PREC_F16X2, D.get(p + ".linear2.weight"), 128, 512, range(0, 8), {h0, h0 + 1}); };
                const int qa = 2 * sgrp, qb = 2 * sgrp + 1;
                const float* wskip = b >= 5 ? D.get("encoder.linear_blocks." + std::to_string(b - 5) + ".weight") : nullptr;
                if (sgrp == 0) {
                    const float* in_w = D.get(p + ".self_attn.in_proj_weight");
                    const std::vector<int> qk_tiles = {2 * h, 2 * h + 1, 8 + 2 * h, 8 + 2 * h + 1};
                    const auto outproj = [&] { pack_gemm(s, PREC_F16X2, D.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * h, 2 * h + 1}); };
                    const auto vproj = [&] { pack_gemm(s, PREC_F16X2, in_w, 384, 128, {16 + 2 * h, 16 + 2 * h + 1}, range(0, 8)); };
                    if (b >= 5) pack_gemm(s, PREC_F16X2, wskip, 128, 256, {2 * h, 2 * h + 1}, range(8, 16));
                    else vproj();
                    pack_gemm(s, PREC_F16X2, in_w, 384, 128, qk_tiles, range(0, 4));
                    if (b >= 5) vproj();
                    pack_gemm(s, PREC_F16X2, in_w, 384, 128, qk_tiles, range(4, 8));
                    outproj();
                } else if (b >= 5) {
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, {2 * h, 2 * h + 1}, range(0, 8));
                }
                f1(qa); f1(qb); f2(qa); f2(qb);
            }
            uint32_t& units = c->den_w8x_units[sgrp];
            if (h == 0) units = (uint32_t)(s.size() / 64);
            else if (s.size() / 64 != units) return fail(AMUSE_ESTATE, "internal: uneven 8-wave fp32x denoiser streams");
            all.insert(all.end(), s.begin(), s.end());
            if (sgrp == 0) all.insert(all.end(), s.begin(), s.begin() + (size_t)kRing8 * 64);  // ring wrap: tail = head
        }
        all.insert(all.end(), (size_t)kRing8 * 64, uint4{0, 0, 0, 0});   // the last B wave's initial ring fill reads past its slice
        if (upload(&c->den_w8x, all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    {
        auto pv = build_pvec(D, "encoder", false);
        if (upload(&c->den_pvec, pv.data(), pv.size() * 4)) return AMUSE_EHIP;
        if (upload(&c->den_pe, D.get("query_pos.pe"), 500 * 128 * 4)) return AMUSE_EHIP;
        float fr[128];
        for (int k = 0; k < 128; ++k) fr[k] = expf(-logf(10000.f) * (float)k / 128.f);
        if (upload(&c->den_freqs, fr, sizeof(fr))) return AMUSE_EHIP;
        auto w1t = transpose(D.get("time_embedding.linear_1.weight"), 128, 256);
        auto w2t = transpose(D.get("time_embedding.linear_2.weight"), 128, 128);
        if (upload(&c->te_w1t, w1t.data(), w1t.size() * 4) || upload(&c->te_w2t, w2t.data(), w2t.size() * 4) ||
            upload(&c->te_b1, D.get("time_embedding.linear_1.bias"), 512) ||
            upload(&c->te_b2, D.get("time_embedding.linear_2.bias"), 512))
            return AMUSE_EHIP;
        const char* names[3] = {"con", "emo", "sty"};
        for (int n = 0; n < 3; ++n) {
            auto wt = transpose(D.get(std::string("emb_proj_") + names[n] + ".1.weight"), 128, 256);
            if (upload(&c->cond_wt[n], wt.data(), wt.size() * 4) ||
                upload(&c->cond_b[n], D.get(std::string("emb_proj_") + names[n] + ".1.bias"), 512))
                return AMUSE_EHIP;
        }
    }
    return 0;
}

int build_prior(amuse_ctx* c, const float* pri, int what = AMUSE_UPD_ALL) {
    static const ParamIndex PI = prior_index();
    if (!g_capture) c->vae_c1_valid[0] = c->vae_c1_valid[1] = c->vae_c1_valid[2] = c->vae_c1_valid[3] = false;   // block 0's hoisted constant belongs to the old decoder weights
    const Params Pp{PI, pri};
    // ---- VAE decoder weight streams: [stage][wave][units]
    for (int prec = 0; prec < 4; ++prec) {
        if (!(what & kUpdBit[prec])) continue;
        std::vector<uint4> all;
        for (int st = 0; st < kVaeStages; ++st) {
            c->vae_stage_base[prec][st] = (uint32_t)(all.size() / 64);
            size_t per_wave = 0;
            for (int w = 0; w < 4; ++w) {
                std::vector<uint4> s;
                if (st >= 1) {
                    const int b = st - 1;
                    pack_outproj_ffn(s, prec, Pp, blk_name("decoder", b), w);
                    if (b >= 4 && b <= 7) pack_skiplin(s, prec, Pp, "decoder", b - 4, w);
                }
                if (st < 9) pack_qkv(s, prec, Pp.get(blk_name("decoder", st) + ".self_attn.in_proj_weight"), w, false);
                else pack_gemm(s, prec, Pp.get("final_layer.weight"), kFeats, 128, range(6 * w, 6 * w + 6), range(0, 8));
                if (w == 0) per_wave = s.size();
                else if (s.size() != per_wave) return fail(AMUSE_ESTATE, "internal: uneven vae wave streams");
                all.insert(all.end(), s.begin(), s.end());
            }
            c->vae_stage_units[prec][st] = (uint32_t)(per_wave / 64);
        }
        all.insert(all.end(), (size_t)kVaeRing * 64, uint4{0, 0, 0, 0});  // the last wave's ring reads past its slice
        if (upload(&c->vae_w[prec], all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    if (what & AMUSE_UPD_F32X) {   // fp32x row stages, eight tiles per workgroup (k_vae_rows8.hip): per stage ONE stream in consumption order, 16-unit
        // (8 hi | lo pairs) LDS stages: every stage is one k-pair x 8 output tiles, or - linear1 - 4 k-pairs x 2 output tiles
        std::vector<uint4> s;
        for (int st = 0; st < kVaeStages; ++st) {
            c->vae_w8x_base[st] = (uint32_t)(s.size() / 64);
            if (st >= 1) {
                const int b = st - 1;
                const std::string p = blk_name("decoder", b);
                pack_gemm(s, PREC_F16X2, Pp.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
                for (int ch = 0; ch < 16; ++ch) {
                    pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8));
                    pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1});
                }
                if (b >= 4 && b <= 7) {   // the skip linear ahead of output block b + 1: the x half, then the popped-skip half
                    const float* wskip = Pp.get("decoder.linear_blocks." + std::to_string(b - 4) + ".weight");
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
                }
            }
            if (st < 9) {
                const float* in_w = Pp.get(blk_name("decoder", st) + ".self_attn.in_proj_weight");
                for (int grp = 0; grp < 3; ++grp) pack_gemm(s, PREC_F16X2, in_w, 384, 128, range(8 * grp, 8 * grp + 8), range(0, 8));
            } else {
                for (int q = 0; q < 4; ++q) pack_gemm(s, PREC_F16X2, Pp.get("final_layer.weight"), kFeats, 128, range(6 * q, 6 * q + 6), range(0, 8));
            }
            if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: rows8 stream is not whole stages");
        }
        s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
        if (upload(&c->vae_w8x, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    if (what & AMUSE_UPD_F32X) {   // fp32x fused decoder (k_vae_fusedx.hip): ONE stream of unit pairs for the clip's eight waves, in consumption order, 16-unit stages
        std::vector<uint4> s;
        for (int b = 0; b < 9; ++b) {
            const std::string p = blk_name("decoder", b);
            if (b >= 5) {   // skip linear ahead of an output block: the x half (k-pairs 0..3), then the popped-skip half
                const float* wskip = Pp.get("decoder.linear_blocks." + std::to_string(b - 5) + ".weight");
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
            }
            const float* in_w = Pp.get(p + ".self_attn.in_proj_weight");
            for (int h = 0; h < 4; ++h) {   // per head: k | v tiles per k-pair (two stages), then q (one stage)
                pack_gemm(s, PREC_F16X2, in_w, 384, 128, {8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
                pack_gemm(s, PREC_F16X2, in_w, 384, 128, {2 * h, 2 * h + 1}, range(0, 8));
            }
            pack_gemm(s, PREC_F16X2, Pp.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
            // FFN in 16 chunks of 32 hidden features, linear1 one chunk ahead: linear1(0), 15 x [linear1(ch + 1), linear2(ch)], linear2(15)
            const auto f1 = [&](int ch) { pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8)); };
            const auto f2 = [&](int ch) { pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1}); };
            f1(0);
            for (int ch = 0; ch < 15; ++ch) { f1(ch + 1); f2(ch); }
            f2(15);
        }
        for (int q = 0; q < 4; ++q) pack_gemm(s, PREC_F16X2, Pp.get("final_layer.weight"), kFeats, 128, range(6 * q, 6 * q + 6), range(0, 8));
        if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: fused fp32x decode stream is not whole stages");
        s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
        if (upload(&c->vae_wfx, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    for (const int p16 : {PREC_BF16, PREC_F16}) {   // fused decode kernel (k_vae_fused.hip; bf16 / fp16 operands): ONE stream for the four waves, in consumption order, cut
        if (!(what & kUpdBit[p16])) continue;
        // into stages of kVaeFusedStageUnits units (every phase below is a whole number of stages)
        std::vector<uint4> s;
        const auto pad = [&](int units) { s.insert(s.end(), (size_t)units * 64, uint4{0, 0, 0, 0}); };
        for (int b = 0; b < 9; ++b) {
            const std::string p = blk_name("decoder", b);
            if (b >= 5) {   // skip linear ahead of an output block: the x half (k-tiles 0..7), then the popped-skip half
                const float* wskip = Pp.get("decoder.linear_blocks." + std::to_string(b - 5) + ".weight");
                pack_gemm(s, p16, wskip, 128, 256, range(0, 8), range(0, 8));
                pack_gemm(s, p16, wskip, 128, 256, range(0, 8), range(8, 16));
            }
            const float* in_w = Pp.get(p + ".self_attn.in_proj_weight");
            for (int h = 0; h < 4; ++h) {   // per head: stage A = k | v tiles per k-pair; stage B = q, out_proj's k-slice
                pack_gemm(s, p16, in_w, 384, 128, {8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
                pack_gemm(s, p16, in_w, 384, 128, {2 * h, 2 * h + 1}, range(0, 8));
                pack_gemm(s, p16, Pp.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * h, 2 * h + 1});
            }
            // FFN in 16 chunks of 32 hidden features, software-pipelined: [linear1(0) | pad], 15 x [linear1(i + 1) | linear2(i)],
            // [linear2(15) | pad]
            const auto f1 = [&](int ch) { pack_gemm(s, p16, Pp.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8)); };
            const auto f2 = [&](int ch) { pack_gemm(s, p16, Pp.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1}); };
            f1(0); pad(8);
            for (int ch = 0; ch < 15; ++ch) { f1(ch + 1); f2(ch); }
            f2(15); pad(8);
        }
        for (int half = 0; half < 2; ++half)   // final_layer ONCE: 24 output tiles in two halves of 48 units (k-pair outer) - the kernel's last stage holds it in LDS whole
            pack_gemm(s, p16, Pp.get("final_layer.weight"), kFeats, 128, range(12 * half, 12 * half + 12), range(0, 8));
        if (s.size() % ((size_t)kVaeFusedStageUnits * 64) != 0) return fail(AMUSE_ESTATE, "internal: fused decode stream is not whole stages");
        pad(2 * kVaeFusedStageUnits);   // the fetch runs two stages ahead
        if (upload(p16 == PREC_BF16 ? &c->vae_wf : &c->vae_wfh, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    {
        auto pv = build_pvec(Pp, "decoder", true);
        if (upload(&c->vae_pvec, pv.data(), pv.size() * 4)) return AMUSE_EHIP;
        std::vector<float> fb(16 * kFeatTiles, 0.f);
        memcpy(fb.data(), Pp.get("final_layer.bias"), kFeats * 4);
        if (upload(&c->vae_final_bias, fb.data(), fb.size() * 4)) return AMUSE_EHIP;
        if (upload(&c->vae_pe, Pp.get("query_pos_decoder.pe"), 500 * 128 * 4)) return AMUSE_EHIP;
        std::vector<float> wv_t(9 * 128 * 128), wo_t(9 * 128 * 128), bv(9 * 128), bo(9 * 128);
        for (int b = 0; b < 9; ++b) {
            const std::string p = blk_name("decoder", b) + ".multihead_attn";
            auto t1 = transpose(Pp.get(p + ".in_proj_weight") + 256 * 128, 128, 128);
            auto t2 = transpose(Pp.get(p + ".out_proj.weight"), 128, 128);
            memcpy(wv_t.data() + (size_t)b * 128 * 128, t1.data(), 128 * 128 * 4);
            memcpy(wo_t.data() + (size_t)b * 128 * 128, t2.data(), 128 * 128 * 4);
            memcpy(bv.data() + b * 128, Pp.get(p + ".in_proj_bias") + 256, 512);
            memcpy(bo.data() + b * 128, Pp.get(p + ".out_proj.bias"), 512);
        }
        if (upload(&c->vae_wv_t, wv_t.data(), wv_t.size() * 4) || upload(&c->vae_wo_t, wo_t.data(), wo_t.size() * 4) ||
            upload(&c->vae_bv, bv.data(), bv.size() * 4) || upload(&c->vae_bo, bo.data(), bo.size() * 4))
            return AMUSE_EHIP;
    }
    // ---- VAE encoder weight streams: stage 0 = skel_embedding (K = 333 padded to 22 k-tiles, 2 output tiles per wave)
    // + in_proj(0); stage i+1 = post-attention of block i (+ skip linear) + in_proj(i+1); stage 9 = post-attention of block 8
    for (int prec = 0; prec < 4; ++prec) {
        if (!(what & kUpdBit[prec]) || !(what & AMUSE_UPD_ENCODER)) continue;
        std::vector<uint4> all;
        for (int st = 0; st < kVaeStages; ++st) {
            c->vaee_stage_base[prec][st] = (uint32_t)(all.size() / 64);
            size_t per_wave = 0;
            for (int w = 0; w < 4; ++w) {
                std::vector<uint4> s;
                if (st == 0) pack_gemm(s, prec, Pp.get("skel_embedding.weight"), 128, kFeats, {2 * w, 2 * w + 1}, range(0, 22));
                if (st >= 1) {
                    const int b = st - 1;
                    pack_outproj_ffn(s, prec, Pp, blk_name("encoder", b), w);
                    if (b >= 4 && b <= 7) pack_skiplin(s, prec, Pp, "encoder", b - 4, w);
                }
                if (st < 9) pack_qkv(s, prec, Pp.get(blk_name("encoder", st) + ".self_attn.in_proj_weight"), w, false);
                if (w == 0) per_wave = s.size();
                else if (s.size() != per_wave) return fail(AMUSE_ESTATE, "internal: uneven vae encoder wave streams");
                all.insert(all.end(), s.begin(), s.end());
            }
            c->vaee_stage_units[prec][st] = (uint32_t)(per_wave / 64);
        }
        all.insert(all.end(), (size_t)kVaeRing * 64, uint4{0, 0, 0, 0});
        if (upload(&c->vaee_w[prec], all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    if ((what & AMUSE_UPD_F32X) && (what & AMUSE_UPD_ENCODER)) {   // encode's stages 1..9 for the fp32x row kernel without split-K (the decoder's stream layout above)
        std::vector<uint4> s;
        for (int st = 0; st < kVaeStages; ++st) {
            c->vaee_w8x_base[st] = (uint32_t)(s.size() / 64);
            if (st == 0) continue;   // (the embedding stage stays with k_vae_rows<f16x2, M_ENC>)
            const int b = st - 1;
            const std::string p = blk_name("encoder", b);
            pack_gemm(s, PREC_F16X2, Pp.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
            for (int ch = 0; ch < 16; ++ch) {
                pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8));
                pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1});
            }
            if (b >= 4 && b <= 7) {
                const float* wskip = Pp.get("encoder.linear_blocks." + std::to_string(b - 4) + ".weight");
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
            }
            if (st < 9) {
                const float* in_w = Pp.get(blk_name("encoder", st) + ".self_attn.in_proj_weight");
                for (int grp = 0; grp < 3; ++grp) pack_gemm(s, PREC_F16X2, in_w, 384, 128, range(8 * grp, 8 * grp + 8), range(0, 8));
            }
            if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: rows8 encoder stream is not whole stages");
        }
        s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
        if (upload(&c->vaee_w8x, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    if ((what & AMUSE_UPD_F32X) && (what & AMUSE_UPD_ENCODER)) {   // encode as one persistent workgroup per clip (k_vae_fusedx.hip k_den_fusedx<encode>): skel_embedding, then
        // the nine encoder blocks in the fused fp32x decoder's order
        std::vector<uint4> s;
        pack_gemm(s, PREC_F16X2, Pp.get("skel_embedding.weight"), 128, kFeats, range(0, 8), range(0, 22));
        for (int b = 0; b < 9; ++b) {
            const std::string p = blk_name("encoder", b);
            if (b >= 5) {
                const float* wskip = Pp.get("encoder.linear_blocks." + std::to_string(b - 5) + ".weight");
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
            }
            const float* in_w = Pp.get(p + ".self_attn.in_proj_weight");
            for (int h = 0; h < 4; ++h) {
                pack_gemm(s, PREC_F16X2, in_w, 384, 128, {8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
                pack_gemm(s, PREC_F16X2, in_w, 384, 128, {2 * h, 2 * h + 1}, range(0, 8));
            }
            pack_gemm(s, PREC_F16X2, Pp.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
            const auto f1 = [&](int ch) { pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8)); };
            const auto f2 = [&](int ch) { pack_gemm(s, PREC_F16X2, Pp.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1}); };
            f1(0);
            for (int ch = 0; ch < 15; ++ch) { f1(ch + 1); f2(ch); }
            f2(15);
        }
        if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: fused fp32x encoder stream is not whole stages");
        s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
        if (upload(&c->vaee_wfx, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
    }
    {
        auto pv = build_pvec(Pp, "encoder", false);
        if (upload(&c->vaee_pvec, pv.data(), pv.size() * 4) ||
            upload(&c->vaee_pe, Pp.get("query_pos_encoder.pe"), 500 * 128 * 4) ||
            upload(&c->vaee_tok, Pp.get("global_motion_token"), 2 * 128 * 4) ||
            upload(&c->vaee_emb_bias, Pp.get("skel_embedding.bias"), 128 * 4))
            return AMUSE_EHIP;
    }
    return 0;
}

int build_ctx(amuse_ctx* c, const float* den, const float* pri) {
    if (int e = c->arch == AMUSE_ARCH_ENC ? build_denoiser(c, den) : variant_build(c, den, AMUSE_UPD_ALL)) return e;
    if (pri)
        if (int e = build_prior(c, pri)) return e;
    HIP_TRY(hipMalloc((void**)&c->d_timesteps, AMUSE_MAX_STEPS * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&c->d_coef, AMUSE_MAX_STEPS * 8 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->d_time_tok, AMUSE_MAX_STEPS * kD * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->d_ts1, sizeof(int)));
    HIP_TRY(hipMalloc((void**)&c->d_tt1, kD * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&c->d_coef1, 8 * sizeof(float)));
    HIP_TRY(hipMemset(c->d_coef1, 0, 8 * sizeof(float)));
    return 0;
}

int cond_tokens(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int* S_out, hipStream_t s) {
    CondArgs ca{};
    int n = 0;
    const float* zs[3] = {con, emo, sty};
    for (int i = 0; i < 3; ++i)
        if (zs[i]) { ca.z[n] = zs[i]; ca.wt[n] = c->cond_wt[i]; ca.bias[n] = c->cond_b[i]; ++n; }
    if (int e = ensure(&c->cond_tok, &c->cond_cap, (size_t)B * 3 * kD)) return e;
    ca.pe = c->den_pe; ca.out = c->cond_tok; ca.B = B; ca.ncond = n; ca.pe_base = 2;
    HIP_TRY(launch_cond_tokens(ca, s));
    *S_out = 2 + n;
    return 0;
}

// clips per workgroup tile: the pin (amuse_set_clips_per_group) or the plan's rule (amuse_host.hpp plan_clips_per_group).  A clip's arithmetic depends on its row offset
// inside the tile only through rounding (the softmax / PV accumulation order), so results are reproduced BITWISE by any launch that uses the same clips per tile and puts
// the clip in the same slot of its tile - i.e. by shards that start at multiples of g (amuse_amd/shard.py takes g from amuse_plan for the job's TOTAL clip count and aligns
// the shards) - and to rounding otherwise.
int pick_group(amuse_ctx* c, int B, int S) {
    const int gmax = 16 / S;
    int g = c->clips_per_group > 0 ? c->clips_per_group : plan_clips_per_group(AMUSE_ARCH_ENC, B, S);
    if (g > gmax) g = gmax;
    c->last_plan[0] = g;
    return g;
}

// kernel choice for one sampling launch: fp32 -> the 4-wave parity kernel (k_sampler.hip); fp32x / bf16 / fp16 -> the 8-wave kernels
bool use_sample8(int precision) { return precision != PREC_F32; }
void set_stream(const amuse_ctx* c, SampleArgs& a, int precision) {
    if (precision < 3) { a.wstream = c->den_w[precision]; a.wave_units = c->den_wave_units[precision]; }
    a.wave_units_a = c->den_w8_units[0]; a.wave_units_b = c->den_w8_units[1];
}
hipError_t dispatch_sample(amuse_ctx* c, SampleArgs& a, int precision, hipStream_t st) {
    if (use_sample8(precision)) {  // with prof_out: stamps come back as [8 waves][96] in the same 768-entry buffer
        if (precision == PREC_F16X2) {
            a.wstream = c->den_w8x; a.wave_units_a = c->den_w8x_units[0]; a.wave_units_b = c->den_w8x_units[1];
            return launch_sample8x(a, st);
        }
        if (precision == PREC_F16) {   // the same streams' layout and unit counts, fp16 weights
            a.wstream = c->den_w8h;
            return launch_sample8h(a, st);
        }
        a.wstream = c->den_w8;
        return launch_sample8(a, st);
    }
    return launch_sample(a, precision, st);
}

// decode / encode kernels of a call: the pin (amuse_set_decode_path) or the plan (amuse_host.hpp plan_decode_path / plan_encode_path), recorded for amuse_debug_last_plan
constexpr int kVaeFusedChunk = 4096;
int decode_path_of(amuse_ctx* c, int precision, int B) {
    int path = resolve_path(c->decode_path, plan_decode_path(precision, B), precision);
    if (path == AMUSE_DECODE_CLIP && !c->vae_wfx) path = AMUSE_DECODE_FUSED;
    return c->last_plan[1] = path;
}
int encode_path_of(amuse_ctx* c, int precision, int B) {
    int path = precision == AMUSE_PREC_F32X ? resolve_path(c->decode_path, plan_encode_path(precision, B), precision) : AMUSE_DECODE_STAGED;
    if (path == AMUSE_DECODE_CLIP && !c->vaee_wfx) path = AMUSE_DECODE_FUSED;
    if (path == AMUSE_DECODE_FUSED && !c->vaee_w8x) path = AMUSE_DECODE_STAGED;
    return c->last_plan[2] = path;
}

int stage_lengths(amuse_ctx* c, const int* lengths, int B, hipStream_t st) {
    if (!lengths) return 0;
    for (int b = 0; b < B; ++b)
        if (lengths[b] < 1 || lengths[b] > kFrames) return fail(AMUSE_EINVAL, "lengths[%d] = %d not in 1..300", b, lengths[b]);
    if (c->len_cap < (size_t)B) {
        if (c->d_lengths) HIP_TRY(hipFree(c->d_lengths));
        c->d_lengths = nullptr; c->len_cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_lengths, (size_t)B * sizeof(int)));
        c->len_cap = B;
    }
    HIP_TRY(hipMemcpyAsync(c->d_lengths, lengths, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int ensure_vae_ws(amuse_ctx* c, int chunk) {
    if (c->vae_cap >= (size_t)chunk) return 0;
    if (c->vae_ws) HIP_TRY(hipFree(c->vae_ws));
    c->vae_ws = nullptr; c->vae_cap = 0;
    HIP_TRY(hipMalloc((void**)&c->vae_ws, ((size_t)chunk * kVaeFloatsPerClip + 256) * sizeof(float)));   // (+ 1 KiB: k_vae_fusedx copies a clip's 4.5 KiB of ca in five 1 KiB pieces)
    c->vae_cap = chunk;
    return 0;
}

int check_common(amuse_ctx* c, const float* con, int B, int precision) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!con) return fail(AMUSE_EINVAL, "con is NULL (the content embedding is mandatory, denoiser.py:153-157)");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    if (precision < AMUSE_PREC_F32 || precision > AMUSE_PREC_F16) return fail(AMUSE_EINVAL, "bad precision %d", precision);
    HIP_TRY(hipSetDevice(c->device));
    return 0;
}
}  // namespace

extern "C" {

int amuse_abi_version(void) { return AMUSE_ABI_VERSION; }
int amuse_debug_f16_split(const float* w, size_t n, uint16_t* hi, uint16_t* lo) {
    if (!w || !hi || !lo) return fail(AMUSE_EINVAL, "NULL argument");
    for (size_t i = 0; i < n; ++i) {
        hi[i] = f2h(w[i]);
        lo[i] = f2h(w[i] - h2f(hi[i]));
    }
    return 0;
}
const char* amuse_last_error(void) { return g_err; }

size_t amuse_denoiser_param_count(int arch) { return arch == AMUSE_ARCH_ENC ? (size_t)AMUSE_DENOISER_PARAMS : variant_param_count(arch); }
int amuse_arch(const amuse_ctx* c) { return c ? c->arch : AMUSE_EINVAL; }
size_t amuse_state_dim(const amuse_ctx* c) { return c ? variant_state_dim(c->arch) : 0; }

amuse_ctx* amuse_create(int device, const float* denoiser_params, size_t n_denoiser, const float* prior_params,
                        size_t n_prior) {
    if (!prior_params) { fail(AMUSE_EINVAL, "NULL parameter array"); return nullptr; }
    return amuse_create_arch(device, AMUSE_ARCH_ENC, denoiser_params, n_denoiser, prior_params, n_prior);
}

amuse_ctx* amuse_create_arch(int device, int arch, const float* denoiser_params, size_t n_denoiser, const float* prior_params,
                             size_t n_prior) {
    if (arch < AMUSE_ARCH_ENC || arch > AMUSE_ARCH_DEC_POSE) { fail(AMUSE_EINVAL, "unknown arch %d", arch); return nullptr; }
    const bool pose = (arch & 2) != 0;
    if (!denoiser_params || (!prior_params && !pose)) { fail(AMUSE_EINVAL, "NULL parameter array"); return nullptr; }
    if (n_denoiser != amuse_denoiser_param_count(arch) || (prior_params ? n_prior != AMUSE_PRIOR_PARAMS : n_prior != 0)) {
        fail(AMUSE_EINVAL, "parameter count mismatch: denoiser %zu (want %zu for arch %d), prior %zu (want %u)", n_denoiser,
             amuse_denoiser_param_count(arch), arch, n_prior, AMUSE_PRIOR_PARAMS);
        return nullptr;
    }
    if (denoiser_index().total != AMUSE_DENOISER_PARAMS || prior_index().total != AMUSE_PRIOR_PARAMS ||
        variant_param_count(AMUSE_ARCH_DEC) != AMUSE_DENOISER_PARAMS_DEC || variant_param_count(AMUSE_ARCH_ENC_POSE) != AMUSE_DENOISER_PARAMS_ENC_POSE ||
        variant_param_count(AMUSE_ARCH_DEC_POSE) != AMUSE_DENOISER_PARAMS_DEC_POSE) {
        fail(AMUSE_ESTATE, "internal: state-dict index does not add up");
        return nullptr;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { fail(AMUSE_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e)); return nullptr; }
    amuse_ctx* c = new amuse_ctx();
    c->device = device;
    c->arch = arch;
    c->has_prior = prior_params != nullptr;
    if (build_ctx(c, denoiser_params, prior_params) != 0) { amuse_destroy(c); return nullptr; }
    return c;
}

int amuse_update_weights(amuse_ctx* c, const float* denoiser_params, size_t n_denoiser, const float* prior_params,
                         size_t n_prior, int what, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!denoiser_params && !prior_params) return fail(AMUSE_EINVAL, "nothing to update");
    if (what < 1 || what > AMUSE_UPD_ALL || !(what & (AMUSE_UPD_F32 | AMUSE_UPD_BF16 | AMUSE_UPD_F32X | AMUSE_UPD_F16))) return fail(AMUSE_EINVAL, "bad `what` mask %d", what);
    if (denoiser_params && n_denoiser != amuse_denoiser_param_count(c->arch))
        return fail(AMUSE_EINVAL, "denoiser parameter count %zu (want %zu)", n_denoiser, amuse_denoiser_param_count(c->arch));
    if (prior_params && n_prior != AMUSE_PRIOR_PARAMS)
        return fail(AMUSE_EINVAL, "prior parameter count %zu (want %u)", n_prior, AMUSE_PRIOR_PARAMS);
    if (prior_params && !c->has_prior) return fail(AMUSE_ESTATE, "this context was created without MotionPrior weights");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // kernels in flight still read the old streams
    if (denoiser_params) {
        if (int e = c->arch == AMUSE_ARCH_ENC ? build_denoiser(c, denoiser_params, what) : variant_build(c, denoiser_params, what)) return e;
        c->T = 0;   // the hoisted time-token table belongs to the old time-embedding weights: set the schedule again
    }
    if (prior_params)
        if (int e = build_prior(c, prior_params, what)) return e;
    return 0;
}

namespace {
// The gather maps.  Every image the builders upload is a permutation of parameters plus zero padding (the one exception,
// the timestep frequencies, does not depend on the parameters and is skipped), so three builder runs on probe parameters - byte k
// of (index + 1), a value bf16 holds exactly - spell out, per image element, which parameter it carries.
int build_repack_maps(amuse_ctx* c) {
    struct Img { std::vector<int> map; bool prior; };
    std::map<void**, Img> imgs;
    std::vector<float> den(AMUSE_DENOISER_PARAMS), pri(AMUSE_PRIOR_PARAMS);
    // element type of an image: 0 = fp32, 1 = bf16, 2 = split-fp16, 3 = fp16 (launch_repack's `kind`)
    auto kind_of = [&](void** slot) {
        if (slot == (void**)&c->den_w[PREC_F16X2] || slot == (void**)&c->den_w8x || slot == (void**)&c->vae_w[PREC_F16X2] ||
            slot == (void**)&c->vaee_w[PREC_F16X2] || slot == (void**)&c->vae_w8x || slot == (void**)&c->vaee_w8x || slot == (void**)&c->vae_wfx || slot == (void**)&c->vaee_wfx) return 2;
        if (slot == (void**)&c->den_w8h || slot == (void**)&c->vae_wfh || slot == (void**)&c->vae_w[PREC_F16] || slot == (void**)&c->vaee_w[PREC_F16]) return 3;
        return (slot == (void**)&c->den_w[PREC_BF16] || slot == (void**)&c->den_w8 || slot == (void**)&c->vae_w[PREC_BF16] ||
                slot == (void**)&c->vae_wf || slot == (void**)&c->vaee_w[PREC_BF16]) ? 1 : 0;
    };
    for (int k = 0; k < 3; ++k) {
        for (size_t i = 0; i < den.size(); ++i) den[i] = (float)(((i + 1) >> (8 * k)) & 255);
        for (size_t i = 0; i < pri.size(); ++i) pri[i] = (float)(((i + 1) >> (8 * k)) & 255);
        for (int which = 0; which < 2; ++which) {
            Capture cap;
            g_capture = &cap;
            g_probe_f16 = true;
            const int rc = which == 0 ? build_denoiser(c, den.data(), AMUSE_UPD_ALL) : build_prior(c, pri.data(), AMUSE_UPD_ALL);
            g_probe_f16 = false;
            g_capture = nullptr;
            if (rc) return rc;
            for (auto& kv : cap.bufs) {
                if (kv.first == (void**)&c->den_freqs) continue;
                const int kind = kind_of(kv.first);
                const size_t n = kv.second.size() / (kind ? 2 : 4);
                Img& im = imgs[kv.first];
                if (k == 0) { im.map.assign(n, 0); im.prior = which == 1; }
                else if (im.map.size() != n) return fail(AMUSE_ESTATE, "internal: packed image changed size between probe runs");
                for (size_t j = 0; j < n; ++j) {
                    float v;
                    if (kind == 1) {
                        uint16_t h;
                        memcpy(&h, kv.second.data() + 2 * j, 2);
                        const uint32_t u = (uint32_t)h << 16;
                        memcpy(&v, &u, 4);
                    } else if (kind == 2 || kind == 3) {
                        uint16_t h;
                        memcpy(&h, kv.second.data() + 2 * j, 2);
                        v = h2f(h);
                    } else {
                        memcpy(&v, kv.second.data() + 4 * j, 4);
                    }
                    if (!(v >= 0.f && v <= 255.f && v == (float)(int)v)) return fail(AMUSE_ESTATE, "internal: a packed image is not a gather of the parameters");
                    im.map[j] |= (int)v << (8 * k);
                }
            }
        }
    }
    for (auto& kv : imgs) {
        void** slot = kv.first;
        const size_t limit = kv.second.prior ? AMUSE_PRIOR_PARAMS : AMUSE_DENOISER_PARAMS;
        for (int m : kv.second.map)
            if (m < 0 || (size_t)m > limit) return fail(AMUSE_ESTATE, "internal: gather index out of range");
        int cls = 0;   // which AMUSE_UPD_* bits the image needs; 0 = small parameters, always replaced
        if (slot == (void**)&c->den_w[PREC_F32] || slot == (void**)&c->vae_w[PREC_F32]) cls = AMUSE_UPD_F32;
        else if (slot == (void**)&c->den_w[PREC_F16X2] || slot == (void**)&c->den_w8x || slot == (void**)&c->vae_w[PREC_F16X2] || slot == (void**)&c->vae_w8x || slot == (void**)&c->vae_wfx) cls = AMUSE_UPD_F32X;
        else if (slot == (void**)&c->vaee_w[PREC_F16X2] || slot == (void**)&c->vaee_w8x || slot == (void**)&c->vaee_wfx) cls = AMUSE_UPD_F32X | AMUSE_UPD_ENCODER;
        else if (slot == (void**)&c->den_w8h || slot == (void**)&c->vae_wfh || slot == (void**)&c->vae_w[PREC_F16]) cls = AMUSE_UPD_F16;
        else if (slot == (void**)&c->vaee_w[PREC_F16]) cls = AMUSE_UPD_F16 | AMUSE_UPD_ENCODER;
        else if (slot == (void**)&c->den_w[PREC_BF16] || slot == (void**)&c->den_w8 || slot == (void**)&c->vae_w[PREC_BF16] || slot == (void**)&c->vae_wf) cls = AMUSE_UPD_BF16;
        else if (slot == (void**)&c->vaee_w[PREC_F32]) cls = AMUSE_UPD_F32 | AMUSE_UPD_ENCODER;
        else if (slot == (void**)&c->vaee_w[PREC_BF16]) cls = AMUSE_UPD_BF16 | AMUSE_UPD_ENCODER;
        int* dmap = nullptr;
        HIP_TRY(hipMalloc((void**)&dmap, kv.second.map.size() * sizeof(int)));
        c->owned.push_back(dmap);
        HIP_TRY(hipMemcpy(dmap, kv.second.map.data(), kv.second.map.size() * sizeof(int), hipMemcpyHostToDevice));
        c->repack.push_back({slot, dmap, kv.second.map.size(), kv.second.prior ? 1 : 0, kind_of(slot), cls});
    }
    return 0;
}
}  // namespace

int amuse_update_weights_device(amuse_ctx* c, const float* denoiser_params_dev, const float* prior_params_dev, int what, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!denoiser_params_dev && !prior_params_dev) return fail(AMUSE_EINVAL, "nothing to update");
    if (what < 1 || what > AMUSE_UPD_ALL || !(what & (AMUSE_UPD_F32 | AMUSE_UPD_BF16 | AMUSE_UPD_F32X | AMUSE_UPD_F16))) return fail(AMUSE_EINVAL, "bad `what` mask %d", what);
    if (c->arch != AMUSE_ARCH_ENC || !c->has_prior) return fail(AMUSE_ESTATE, "the device re-pack exists for the shipped configuration (AMUSE_ARCH_ENC) only; use amuse_update_weights");
    HIP_TRY(hipSetDevice(c->device));
    if (c->repack.empty())
        if (int e = build_repack_maps(c)) return e;
    if (prior_params_dev) c->vae_c1_valid[0] = c->vae_c1_valid[1] = c->vae_c1_valid[2] = c->vae_c1_valid[3] = false;
    for (const auto& r : c->repack) {
        const float* src = r.prior ? prior_params_dev : denoiser_params_dev;
        if (!src || !*r.slot) continue;                 // (an image the context never built, e.g. the 4-wave bf16 stream after an update)
        if ((r.cls & what) != r.cls) continue;          // every bit the image needs must be requested
        HIP_TRY(launch_repack(src, r.map, *r.slot, r.n, r.kind, (hipStream_t)stream));
    }
    // the hoisted time-token table belongs to the old time-embedding weights: rebuilt here, stream-ordered, from the schedule's
    // timesteps (still on the device) - no host round trip, the schedule stays set
    if (denoiser_params_dev && c->T > 0)
        HIP_TRY(launch_time_tokens(c->d_timesteps, c->T, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2, c->den_pe + kD, c->d_time_tok,
                                   (hipStream_t)stream));
    return 0;
}

void amuse_destroy(amuse_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    variant_destroy(c);
    for (void* p : c->owned)
        if (p) (void)hipFree(p);
    void* ptrs[] = {c->den_w[0], c->den_w[1], c->den_w[2], c->den_w8, c->den_w8h, c->den_w8x, c->vae_wfh, c->den_pvec, c->den_pe, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t,
                    c->te_b2, c->cond_wt[0], c->cond_wt[1], c->cond_wt[2], c->cond_b[0], c->cond_b[1], c->cond_b[2],
                    c->vae_w[0], c->vae_w[1], c->vae_w[2], c->vae_w[3], c->vae_w8x, c->vae_wfx, c->vaee_w8x, c->vaee_wfx, c->vaee_w[2], c->vaee_w[3], c->vae_pvec, c->vae_final_bias, c->vae_pe, c->vae_wv_t, c->vae_bv,
                    c->vae_wo_t, c->vae_bo, c->vaee_w[0], c->vaee_w[1], c->vaee_pvec, c->vaee_pe, c->vaee_tok,
                    c->vaee_emb_bias, c->d_timesteps, c->d_coef, c->d_time_tok, c->d_ts1, c->d_tt1, c->d_coef1,
                    c->cond_tok, c->lat_tmp, c->fwd_ws, c->vae_ws, c->d_lengths, c->vae_wf, c->vae_skip, c->vae_ca_ws, c->vae_c1[0], c->vae_c1[1], c->vae_c1[2], c->vae_c1[3]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->vae_c1_ev)
        if (e) (void)hipEventDestroy(e);
    delete c;
}

int amuse_set_clips_per_group(amuse_ctx* c, int g) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (g < 0 || g > 5) return fail(AMUSE_EINVAL, "clips per group must be 0 (auto) .. 5, got %d", g);
    c->clips_per_group = g;
    return 0;
}

int amuse_set_decode_path(amuse_ctx* c, int path) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (path != AMUSE_DECODE_AUTO && path != AMUSE_DECODE_STAGED && path != AMUSE_DECODE_FUSED && path != AMUSE_DECODE_CLIP)
        return fail(AMUSE_EINVAL, "bad decode path %d", path);
    c->decode_path = path;
    return 0;
}

int amuse_plan(int arch, int precision, int clips_total, int tokens, int* clips_per_group, int* decode_path, int* encode_path, int* step_path) {
    if (arch < AMUSE_ARCH_ENC || arch > AMUSE_ARCH_DEC_POSE) return fail(AMUSE_EINVAL, "bad arch %d", arch);
    if (precision < AMUSE_PREC_F32 || precision > AMUSE_PREC_F16) return fail(AMUSE_EINVAL, "bad precision %d", precision);
    if (clips_total < 1) return fail(AMUSE_EINVAL, "clips_total must be >= 1, got %d", clips_total);
    if (tokens < 3 || tokens > 5) return fail(AMUSE_EINVAL, "tokens must be 3..5 (latent + time + content [+ emotion] [+ style]), got %d", tokens);
    if (clips_per_group) *clips_per_group = plan_clips_per_group(arch, clips_total, tokens);
    if (decode_path) *decode_path = plan_decode_path(precision, clips_total);
    if (encode_path) *encode_path = plan_encode_path(precision, clips_total);
    if (step_path) *step_path = plan_step_path(arch, precision, clips_total);
    return 0;
}

int amuse_debug_last_plan(const amuse_ctx* c, int* clips_per_group, int* decode_path, int* encode_path, int* step_path) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (clips_per_group) *clips_per_group = c->last_plan[0];
    if (decode_path) *decode_path = c->last_plan[1];
    if (encode_path) *encode_path = c->last_plan[2];
    if (step_path) *step_path = c->last_plan[3];
    return 0;
}

int amuse_debug_set_decode_tap(amuse_ctx* c, float* tap_out) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    c->decode_tap = tap_out;
    return 0;
}

int amuse_debug_set_ablation(amuse_ctx* c, int mask) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (mask < 0 || mask > 1) return fail(AMUSE_EINVAL, "bad ablation mask %d", mask);
    c->ablate = mask;
    return 0;
}

int amuse_set_schedule(amuse_ctx* c, const amuse_schedule* s, void* stream) {
    if (!c || !s) return fail(AMUSE_EINVAL, "NULL argument");
    if (s->n_steps < 1 || s->n_steps > AMUSE_MAX_STEPS) return fail(AMUSE_EINVAL, "n_steps %d out of range", s->n_steps);
    if (!s->timesteps || !s->coef) return fail(AMUSE_EINVAL, "schedule tables are NULL");
    for (int i = 0; i < s->n_steps; ++i) {
        if (s->timesteps[i] < 0) return fail(AMUSE_EINVAL, "negative timestep at step %d", i);
        if (!(s->coef[i * 8 + 1] > 0.f)) return fail(AMUSE_EINVAL, "sqrt(alpha_bar) must be > 0 at step %d", i);
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(c->d_timesteps, s->timesteps, s->n_steps * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_coef, s->coef, (size_t)s->n_steps * 8 * sizeof(float), hipMemcpyHostToDevice));
    if (s->freqs) HIP_TRY(hipMemcpy(c->den_freqs, s->freqs, 128 * sizeof(float), hipMemcpyHostToDevice));
    c->T = s->n_steps;
    if (c->arch != AMUSE_ARCH_ENC) {
        if (int e = variant_set_schedule(c, st)) { c->T = 0; return e; }
    } else {
        HIP_TRY(launch_time_tokens(c->d_timesteps, s->n_steps, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2,
                                   c->den_pe + kD, c->d_time_tok, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int amuse_sample(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int precision,
                 uint64_t seed, uint64_t clip_index0, const float* x_init, const float* step_noise,
                 float* latents_out, float* traj_out, void* stream) {
    if (int e = check_common(c, con, B, precision)) return e;
    if (c->T < 1) return fail(AMUSE_ESTATE, "amuse_set_schedule has not been called");
    if (!latents_out) return fail(AMUSE_EINVAL, "latents_out is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (c->arch != AMUSE_ARCH_ENC)
        return variant_sample(c, con, emo, sty, B, precision, seed, clip_index0, x_init, step_noise, latents_out, traj_out, st);
    int S = 0;
    if (int e = cond_tokens(c, con, emo, sty, B, &S, st)) return e;
    SampleArgs a{};
    set_stream(c, a, precision);
    a.pvec = c->den_pvec; a.time_tok = c->d_time_tok; a.cond_tok = c->cond_tok; a.pe0 = c->den_pe;
    a.coef = c->d_coef; a.x_init = x_init; a.step_noise = step_noise;
    a.latents_out = latents_out; a.traj_out = traj_out; a.eps_out = nullptr; a.tap_out = nullptr;
    a.seed = seed; a.clip0 = clip_index0;
    a.B = B; a.T = c->T; a.S = S; a.G = pick_group(c, B, S); a.no_update = 0;
    HIP_TRY(dispatch_sample(c, a, precision, st));
    return 0;
}

int amuse_profile_sample(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int precision,
                         int prof_step, unsigned long long* stamps_out, void* stream) {
    if (int e = check_common(c, con, B, precision)) return e;
    if (c->T < 1) return fail(AMUSE_ESTATE, "amuse_set_schedule has not been called");
    if (!stamps_out || prof_step < 0 || prof_step >= c->T) return fail(AMUSE_EINVAL, "bad stamps_out / prof_step");
    if (c->arch != AMUSE_ARCH_ENC) return fail(AMUSE_ESTATE, "phase stamps exist for the AMUSE_ARCH_ENC sampling kernels only");
    hipStream_t st = (hipStream_t)stream;
    int S = 0;
    if (int e = cond_tokens(c, con, emo, sty, B, &S, st)) return e;
    if (int e = ensure(&c->lat_tmp, &c->lat_cap, (size_t)B * kD)) return e;
    HIP_TRY(hipMemsetAsync(stamps_out, 0, 4 * kProfStamps * sizeof(unsigned long long), st));
    SampleArgs a{};
    set_stream(c, a, precision);
    a.pvec = c->den_pvec; a.time_tok = c->d_time_tok; a.cond_tok = c->cond_tok; a.pe0 = c->den_pe;
    a.coef = c->d_coef; a.latents_out = c->lat_tmp;
    a.seed = 1; a.clip0 = 0;
    a.B = B; a.T = c->T; a.S = S; a.G = pick_group(c, B, S); a.no_update = 0;
    a.prof_out = stamps_out; a.prof_step = prof_step;
    HIP_TRY(dispatch_sample(c, a, precision, st));
    return 0;
}

int amuse_denoise_step(amuse_ctx* c, const float* x_t, int timestep, const float* con, const float* emo,
                       const float* sty, int B, int precision, float* eps_out, float* tap_out, void* stream) {
    if (int e = check_common(c, con, B, precision)) return e;
    if (!x_t || !eps_out) return fail(AMUSE_EINVAL, "x_t / eps_out is NULL");
    if (timestep < 0) return fail(AMUSE_EINVAL, "negative timestep");
    hipStream_t st = (hipStream_t)stream;
    if (c->arch != AMUSE_ARCH_ENC) return variant_denoise(c, x_t, &timestep, false, con, emo, sty, nullptr, B, precision, eps_out, tap_out, st);
    HIP_TRY(hipMemcpyAsync(c->d_ts1, &timestep, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `timestep` lives on this call's stack
    HIP_TRY(launch_time_tokens(c->d_ts1, 1, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2, c->den_pe + kD,
                               c->d_tt1, st));
    int S = 0;
    if (int e = cond_tokens(c, con, emo, sty, B, &S, st)) return e;
    SampleArgs a{};
    set_stream(c, a, precision);
    a.pvec = c->den_pvec; a.time_tok = c->d_tt1; a.cond_tok = c->cond_tok; a.pe0 = c->den_pe;
    a.coef = c->d_coef1; a.x_init = x_t; a.step_noise = nullptr;
    a.latents_out = nullptr; a.traj_out = nullptr; a.eps_out = eps_out; a.tap_out = tap_out;
    a.seed = 0; a.clip0 = 0;
    a.B = B; a.T = 1; a.S = S; a.G = pick_group(c, B, S); a.no_update = 1;
    HIP_TRY(dispatch_sample(c, a, precision, st));
    return 0;
}

int amuse_diffusion_forward(amuse_ctx* c, const float* z0, const float* noise, const int* timesteps, const float* sqrt_ab,
                            const float* sqrt_1m_ab, const float* con, const float* emo, const float* sty, int B,
                            int precision, float* noisy_out, float* noise_pred_out, void* stream) {
    if (int e = check_common(c, con, B, precision)) return e;
    if (!z0 || !noise || !timesteps || !sqrt_ab || !sqrt_1m_ab || !noise_pred_out) return fail(AMUSE_EINVAL, "NULL argument");
    for (int b = 0; b < B; ++b)
        if (timesteps[b] < 0) return fail(AMUSE_EINVAL, "timesteps[%d] = %d is negative", b, timesteps[b]);
    hipStream_t st = (hipStream_t)stream;
    if (c->arch != AMUSE_ARCH_ENC) {   // the same call on a variant's state ([B][state_dim]); its denoiser pass takes the per-clip timesteps
        const size_t sd = variant_state_dim(c->arch);
        if (int e = ensure(&c->fwd_ws, &c->fwd_cap, (size_t)B * (sd + 2))) return e;
        float* noisy_v = c->fwd_ws;
        float* sa_v = noisy_v + (size_t)B * sd;
        float* sb_v = sa_v + B;
        HIP_TRY(hipMemcpyAsync(sa_v, sqrt_ab, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(sb_v, sqrt_1m_ab, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // the host arrays belong to the caller
        HIP_TRY(launch_add_noise(z0, noise, sa_v, sb_v, noisy_v, B, st, (int)sd));
        if (int e = variant_denoise(c, noisy_v, timesteps, true, con, emo, sty, nullptr, B, precision, noise_pred_out, nullptr, st)) return e;
        if (noisy_out) HIP_TRY(hipMemcpyAsync(noisy_out, noisy_v, (size_t)B * sd * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    // per-clip scratch: noisy latents, time tokens, the two coefficient vectors and the timesteps
    if (int e = ensure(&c->fwd_ws, &c->fwd_cap, (size_t)B * (2 * kD + 3))) return e;
    float* noisy = c->fwd_ws;
    float* ttok = noisy + (size_t)B * kD;
    float* sa = ttok + (size_t)B * kD;
    float* sb = sa + B;
    int* ts = reinterpret_cast<int*>(sb + B);
    HIP_TRY(hipMemcpyAsync(ts, timesteps, (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sa, sqrt_ab, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sb, sqrt_1m_ab, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the host arrays belong to the caller
    HIP_TRY(launch_add_noise(z0, noise, sa, sb, noisy, B, st));
    HIP_TRY(launch_time_tokens(ts, B, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2, c->den_pe + kD, ttok, st));
    int S = 0;
    if (int e = cond_tokens(c, con, emo, sty, B, &S, st)) return e;
    SampleArgs a{};
    set_stream(c, a, precision);
    a.pvec = c->den_pvec; a.time_tok = ttok; a.time_tok_clip = ttok; a.cond_tok = c->cond_tok; a.pe0 = c->den_pe;
    a.coef = c->d_coef1; a.x_init = noisy; a.eps_out = noise_pred_out;
    a.B = B; a.T = 1; a.S = S; a.G = pick_group(c, B, S); a.no_update = 1;
    HIP_TRY(dispatch_sample(c, a, precision, st));
    if (noisy_out) HIP_TRY(hipMemcpyAsync(noisy_out, noisy, (size_t)B * kD * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

namespace {
// Block 0's hoisted constant (vae_c1) is produced on the stream of the decode that first needed it; the validity flag is host state.  A later decode on
// ANOTHER stream (the trainer's sampler stream beside the caller's) must not read it before those launches finish: an event marks the producer's place,
// and a consumer on a different stream waits on it.  Same stream: nothing to do (stream order).
hipError_t c1_produced(amuse_ctx* c, int i, hipStream_t st) {
    if (!c->vae_c1_ev[i])
        if (hipError_t e = hipEventCreateWithFlags(&c->vae_c1_ev[i], hipEventDisableTiming)) return e;
    c->vae_c1_stream[i] = st;
    return hipEventRecord(c->vae_c1_ev[i], st);
}
hipError_t c1_consumed(amuse_ctx* c, int i, hipStream_t st) {
    if (!c->vae_c1_ev[i] || c->vae_c1_stream[i] == st) return hipSuccess;
    return hipStreamWaitEvent(st, c->vae_c1_ev[i], 0);
}
}  // namespace

int amuse_vae_decode(amuse_ctx* c, const float* z, const int* lengths, int B, int precision, int quat_mode,
                     float* feats_out, float* poses_out, float* trans_out, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!c->has_prior) return fail(AMUSE_ESTATE, "this context was created without MotionPrior weights");
    if (!z) return fail(AMUSE_EINVAL, "z is NULL");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    if (precision < AMUSE_PREC_F32 || precision > AMUSE_PREC_F16) return fail(AMUSE_EINVAL, "bad precision %d", precision);
    if (quat_mode != AMUSE_QUAT_P3D && quat_mode != AMUSE_QUAT_LEGACY) return fail(AMUSE_EINVAL, "bad quat_mode %d", quat_mode);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (int e = stage_lengths(c, lengths, B, st)) return e;
    const int path = decode_path_of(c, precision, B);
    if (is_op16(precision) && path != AMUSE_DECODE_STAGED) {
        // bf16 / fp16 throughput modes from kFusedMinClips clips up: one persistent workgroup per clip (k_vae_fused.hip)
        const int chunk = B < kVaeFusedChunk ? B : kVaeFusedChunk;
        if (c->vae_skip_cap < (size_t)chunk) {
            if (c->vae_skip) HIP_TRY(hipFree(c->vae_skip));
            c->vae_skip = nullptr; c->vae_skip_cap = 0;
            HIP_TRY(hipMalloc((void**)&c->vae_skip, (size_t)chunk * kVaeFusedSkipBytesPerClip));
            c->vae_skip_cap = chunk;
        }
        if (int e = ensure(&c->vae_ca_ws, &c->vae_ca_cap, (size_t)chunk * kLayers * kD + 256)) return e;   // + the DMA's overrun
        // Block 0's self-attention half does not depend on the latent (k_vae_fused.hip / amuse_fused.hpp decoder_block, c1): computed once
        // per weight set by the kernel's own tapped instantiation on one clip, stream-ordered in front of the first decode that uses it
        // (the same bits as recomputing block 0 per clip: profiles/r04_decode_hoist_ab.txt).
        const int pi = precision == PREC_F16 ? 1 : 0;
        if (!c->vae_c1_valid[pi] && !c->decode_tap && !(c->ablate & 1)) {
            constexpr size_t kTapFloats = (size_t)11 * kFrames * kD;
            if (!c->vae_c1[pi]) HIP_TRY(hipMalloc((void**)&c->vae_c1[pi], ((size_t)kFrames * kD + kTapFloats) * sizeof(float)));
            float* tap = c->vae_c1[pi] + (size_t)kFrames * kD;
            HIP_TRY(launch_vae_ca(z, c->vae_wv_t, c->vae_bv, c->vae_wo_t, c->vae_bo, c->vae_ca_ws, 1, st));
            VaeFusedArgs fa{};
            fa.wstream = precision == PREC_F16 ? c->vae_wfh : c->vae_wf; fa.pvec = c->vae_pvec; fa.final_bias = c->vae_final_bias; fa.pe = c->vae_pe;
            fa.ca = c->vae_ca_ws; fa.skip = c->vae_skip; fa.B = 1; fa.quat_mode = quat_mode; fa.tap_out = tap;
            HIP_TRY(precision == PREC_F16 ? launch_vae_fusedh(fa, st) : launch_vae_fused(fa, st));
            HIP_TRY(hipMemcpyAsync(c->vae_c1[pi], tap + (size_t)10 * kFrames * kD, (size_t)kFrames * kD * sizeof(float), hipMemcpyDeviceToDevice, st));
            HIP_TRY(c1_produced(c, pi, st));
            c->vae_c1_valid[pi] = true;
        }
        if (c->vae_c1_valid[pi]) HIP_TRY(c1_consumed(c, pi, st));
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = (B - b0) < chunk ? (B - b0) : chunk;
            HIP_TRY(launch_vae_ca(z + (size_t)b0 * kD, c->vae_wv_t, c->vae_bv, c->vae_wo_t, c->vae_bo, c->vae_ca_ws, nb, st));
            VaeFusedArgs fa{};
            fa.wstream = precision == PREC_F16 ? c->vae_wfh : c->vae_wf; fa.pvec = c->vae_pvec; fa.final_bias = c->vae_final_bias; fa.pe = c->vae_pe;
            fa.ca = c->vae_ca_ws; fa.lengths = lengths ? c->d_lengths + b0 : nullptr; fa.skip = c->vae_skip;
            fa.feats_out = feats_out ? feats_out + (size_t)b0 * kFrames * kFeats : nullptr;
            fa.poses_out = poses_out ? poses_out + (size_t)b0 * kFrames * kJoints * 3 : nullptr;
            fa.trans_out = trans_out ? trans_out + (size_t)b0 * kFrames * 3 : nullptr;
            fa.B = nb; fa.quat_mode = quat_mode;
            fa.tap_out = b0 == 0 ? c->decode_tap : nullptr;   // (amuse_debug_set_decode_tap: tests)
            fa.c1 = (c->vae_c1_valid[pi] && !fa.tap_out) ? c->vae_c1[pi] : nullptr;
            fa.ablate_attention = c->ablate & 1;
            HIP_TRY(precision == PREC_F16 ? launch_vae_fusedh(fa, st) : launch_vae_fused(fa, st));
        }
        return 0;
    }
    const int chunk = B < kVaeChunk ? B : kVaeChunk;
    if (int e = ensure_vae_ws(c, chunk)) return e;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        const size_t rows = (size_t)nb * kFrames;
        float* ws = c->vae_ws;
        VaeRowsArgs ra{};
        ra.wstream = c->vae_w[precision];
        memcpy(ra.stage_base, c->vae_stage_base[precision], sizeof(ra.stage_base));
        memcpy(ra.stage_units, c->vae_stage_units[precision], sizeof(ra.stage_units));
        ra.pvec = c->vae_pvec; ra.final_bias = c->vae_final_bias; ra.pe = c->vae_pe;
        ra.x = ws; ws += rows * kD;
        ra.q = ws; ws += rows * kD;
        ra.k = ws; ws += rows * kD;
        ra.v = ws; ws += rows * kD;
        float* attn_o = ws; ws += rows * kD;
        ra.attn_o = attn_o;
        ra.skip = ws; ws += 4 * rows * kD;
        float* ca = ws;
        ra.ca = ca;
        ra.lengths = lengths ? c->d_lengths + b0 : nullptr;
        ra.feats_out = feats_out ? feats_out + (size_t)b0 * kFrames * kFeats : nullptr;
        ra.poses_out = poses_out ? poses_out + (size_t)b0 * kFrames * kJoints * 3 : nullptr;
        ra.trans_out = trans_out ? trans_out + (size_t)b0 * kFrames * 3 : nullptr;
        ra.B = nb; ra.quat_mode = quat_mode; ra.tiles = 19;
        HIP_TRY(launch_vae_ca(z + (size_t)b0 * kD, c->vae_wv_t, c->vae_bv, c->vae_wo_t, c->vae_bo, ca, nb, st));
        VaeAttnArgs aa{};
        aa.q = ra.q; aa.k = ra.k; aa.v = ra.v; aa.lengths = ra.lengths; aa.o = attn_o; aa.B = nb; aa.q_tiles = 19;
        // fp32x: the row stages without split-K (k_vae_rows8.hip; FUSED) or the split-K row kernel k_vae_rows<f16x2> (STAGED), chosen from the clips of the CALL (not of the
        // chunk: a job's last chunk must not change kernels) or as amuse_set_decode_path pins it, so that a job-level choice (amuse_plan) keeps fp32x shards bitwise too
        const bool rows8 = precision == PREC_F16X2 && path != AMUSE_DECODE_STAGED;
        // the fp32x decode as ONE persistent workgroup per clip (k_vae_fusedx.hip; CLIP) where the call's clips fill rounds of the chip; its scratch arrays are this path's
        // attn_o and skip
        if (precision == PREC_F16X2 && path == AMUSE_DECODE_CLIP) {
            // block 0's self-attention half is one [300][128] constant per weight set for full-length clips (the decoder's queries are the positional table): computed
            // once by THIS kernel on one clip (c1_out: the same instruction stream, the same bits), then every full-length clip starts behind norm1 and the kernel's
            // weight stream behind block 0's sixteen attention stages.  Explicit lengths take the full path.
            const bool hoistx = !lengths && !c->decode_tap;
            if (hoistx && !c->vae_c1_valid[3]) {
                if (!c->vae_c1[3]) HIP_TRY(hipMalloc((void**)&c->vae_c1[3], (size_t)kFrames * kD * sizeof(float)));
                VaeFusedXArgs px{};
                px.wstream = c->vae_wfx; px.pvec = c->vae_pvec; px.final_bias = c->vae_final_bias; px.pe = c->vae_pe; px.ca = ca;
                px.skip = ra.skip; px.obuf = attn_o; px.B = 1; px.quat_mode = quat_mode; px.c1_out = c->vae_c1[3];
                HIP_TRY(launch_vae_fusedx(px, st));
                HIP_TRY(c1_produced(c, 3, st));
                c->vae_c1_valid[3] = true;
            }
            if (hoistx) HIP_TRY(c1_consumed(c, 3, st));
            VaeFusedXArgs fx{};
            fx.c1 = hoistx ? c->vae_c1[3] : nullptr;
            fx.wstream = c->vae_wfx; fx.pvec = c->vae_pvec; fx.final_bias = c->vae_final_bias; fx.pe = c->vae_pe; fx.ca = ca; fx.lengths = ra.lengths;
            fx.skip = ra.skip; fx.obuf = attn_o; fx.feats_out = ra.feats_out; fx.poses_out = ra.poses_out; fx.trans_out = ra.trans_out;
            fx.tap_out = b0 == 0 ? c->decode_tap : nullptr;
            fx.B = nb; fx.quat_mode = quat_mode;
            HIP_TRY(launch_vae_fusedx(fx, st));
            continue;
        }
        VaeRowsArgs r8 = ra;
        if (rows8) {
            r8.wstream = c->vae_w8x;
            memcpy(r8.stage_base, c->vae_w8x_base, sizeof(r8.stage_base));
        }
        // fp32x row stages, all clips full length: block 0's self-attention half is one [300][128] constant per weight set (see the fused
        // path above) - computed once by these kernels themselves on one clip (a tile's arithmetic does not depend on its launch: same bits),
        // then every decode starts at stage 1 behind norm1.  Explicit lengths (even all 300) take the full path.
        const bool hoist8 = rows8 && !lengths;
        if (hoist8 && !c->vae_c1_valid[2]) {
            if (!c->vae_c1[2]) HIP_TRY(hipMalloc((void**)&c->vae_c1[2], (size_t)kFrames * kD * sizeof(float)));
            VaeRowsArgs p8 = r8;
            p8.B = 1; p8.lengths = nullptr; p8.feats_out = p8.poses_out = p8.trans_out = nullptr;
            VaeAttnArgs pa = aa;
            pa.B = 1; pa.lengths = nullptr;
            p8.stage = 0;
            HIP_TRY(launch_vae_rows8x(p8, st));
            HIP_TRY(launch_vae_attn(pa, precision, VAE_MODE_DEC, st));
            p8.stage = 1; p8.c1_out = c->vae_c1[2];
            HIP_TRY(launch_vae_rows8x(p8, st));
            HIP_TRY(c1_produced(c, 2, st));
            c->vae_c1_valid[2] = true;
        }
        if (hoist8) { HIP_TRY(c1_consumed(c, 2, st)); r8.c1 = c->vae_c1[2]; }
        for (int stage = hoist8 ? 1 : 0; stage < kVaeStages; ++stage) {
            ra.stage = stage;
            r8.stage = stage;
            if (rows8) HIP_TRY(launch_vae_rows8x(r8, st));
            else HIP_TRY(launch_vae_rows(ra, precision, VAE_MODE_DEC, st));
            if (stage < kLayers) HIP_TRY(launch_vae_attn(aa, precision, VAE_MODE_DEC, st));
        }
    }
    return 0;
}

int amuse_vae_encode(amuse_ctx* c, const float* feats, const int* lengths, int B, int precision, const float* eps,
                     float* mu_out, float* std_out, float* latent_out, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!c->has_prior) return fail(AMUSE_ESTATE, "this context was created without MotionPrior weights");
    if (!feats) return fail(AMUSE_EINVAL, "feats is NULL");
    if (!mu_out && !std_out && !latent_out) return fail(AMUSE_EINVAL, "no output requested");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    if (precision < AMUSE_PREC_F32 || precision > AMUSE_PREC_F16) return fail(AMUSE_EINVAL, "bad precision %d", precision);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    if (int e = stage_lengths(c, lengths, B, st)) return e;
    const int chunk = B < kVaeChunk ? B : kVaeChunk;
    if (int e = ensure_vae_ws(c, chunk)) return e;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        const size_t rows = (size_t)nb * kEncRows;
        float* ws = c->vae_ws;
        VaeRowsArgs ra{};
        ra.wstream = c->vaee_w[precision];
        memcpy(ra.stage_base, c->vaee_stage_base[precision], sizeof(ra.stage_base));
        memcpy(ra.stage_units, c->vaee_stage_units[precision], sizeof(ra.stage_units));
        ra.pvec = c->vaee_pvec; ra.pe = c->vaee_pe; ra.tok = c->vaee_tok; ra.emb_bias = c->vaee_emb_bias;
        ra.x = ws; ws += rows * kD;
        ra.q = ws; ws += rows * kD;
        ra.k = ws; ws += rows * kD;
        ra.v = ws; ws += rows * kD;
        float* attn_o = ws; ws += rows * kD;
        ra.attn_o = attn_o;
        ra.skip = ws; ws += 4 * rows * kD;
        ra.stats_out = ws;  // [nb][2][128] <= the decoder's [nb][9][128] cross-attention slot
        ra.lengths = lengths ? c->d_lengths + b0 : nullptr;
        ra.enc_feats = feats + (size_t)b0 * kFrames * kFeats;
        ra.B = nb;
        VaeAttnArgs aa{};
        aa.q = ra.q; aa.k = ra.k; aa.v = ra.v; aa.lengths = ra.lengths; aa.o = attn_o; aa.B = nb;
        // fp32x from kFusedMinClips clips of the call (or as amuse_set_decode_path pins it - the decode's rule): stages 1..9 on the row kernel without split-K
        const int epath = encode_path_of(c, precision, B);
        const bool rows8 = epath != AMUSE_DECODE_STAGED;
        if (epath == AMUSE_DECODE_CLIP) {   // (the decode's rule and pins) the whole encoder as one persistent workgroup per clip
            DenFusedXArgs fx{};
            fx.wstream = c->vaee_wfx; fx.pvec = c->vaee_pvec; fx.emb_bias = c->vaee_emb_bias; fx.pe = c->vaee_pe; fx.ttok = c->vaee_tok;
            fx.x_in = ra.enc_feats; fx.eps_out = ra.stats_out; fx.lengths = ra.lengths; fx.obuf = attn_o; fx.skip = ra.skip;
            fx.B = nb; fx.S = kEncRows; fx.npre = 2; fx.encode = 1;
            HIP_TRY(launch_den_fusedx(fx, st));
            const size_t o = (size_t)b0 * kD;
            HIP_TRY(launch_vae_latent(ra.stats_out, eps ? eps + o : nullptr, mu_out ? mu_out + o : nullptr, std_out ? std_out + o : nullptr, latent_out ? latent_out + o : nullptr, nb, st));
            continue;
        }
        VaeRowsArgs r8 = ra;
        if (rows8) {
            r8.wstream = c->vaee_w8x;
            memcpy(r8.stage_base, c->vaee_w8x_base, sizeof(r8.stage_base));
        }
        for (int stage = 0; stage < kVaeStages; ++stage) {
            ra.stage = stage;
            ra.tiles = stage == kVaeStages - 1 ? 1 : 19;       // only the distribution rows leave the last block
            aa.q_tiles = stage == kLayers - 1 ? 1 : 19;
            r8.stage = stage;
            r8.tiles = ra.tiles;
            if (rows8 && stage >= 1) HIP_TRY(launch_vae_rows8x(r8, st, VAE_MODE_ENC));
            else HIP_TRY(launch_vae_rows(ra, precision, VAE_MODE_ENC, st));
            if (stage < kLayers) HIP_TRY(launch_vae_attn(aa, precision, VAE_MODE_ENC, st));
        }
        const size_t o = (size_t)b0 * kD;
        HIP_TRY(launch_vae_latent(ra.stats_out, eps ? eps + o : nullptr, mu_out ? mu_out + o : nullptr,
                                  std_out ? std_out + o : nullptr, latent_out ? latent_out + o : nullptr, nb, st));
    }
    return 0;
}

int amuse_smplx_to_feats(amuse_ctx* c, const float* poses, const float* trans, int B, float* feats_out, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!poses || !trans || !feats_out) return fail(AMUSE_EINVAL, "NULL argument");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_smplx_to_feats(poses, trans, (size_t)B * kFrames, feats_out, (hipStream_t)stream));
    return 0;
}

int amuse_diffusion_backward(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int precision,
                             int quat_mode, uint64_t seed, uint64_t clip_index0, const float* x_init,
                             const float* step_noise, float* latents_out, float* poses_out, float* trans_out,
                             void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!poses_out || !trans_out) return fail(AMUSE_EINVAL, "poses_out / trans_out is NULL");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    float* lat = latents_out;
    if (!lat) {
        HIP_TRY(hipSetDevice(c->device));
        if (int e = ensure(&c->lat_tmp, &c->lat_cap, (size_t)B * variant_state_dim(c->arch))) return e;
        lat = c->lat_tmp;
    }
    if (int e = amuse_sample(c, con, emo, sty, B, precision, seed, clip_index0, x_init, step_noise, lat, nullptr, stream))
        return e;
    // diffusion_only: the sampled state IS the feature sequence - no decode (infer_ldm.py:165), only the conversion of :168-173
    if (c->arch & 2) return amuse_feats_to_smplx(c, lat, B, quat_mode, poses_out, trans_out, stream);
    return amuse_vae_decode(c, lat, nullptr, B, precision, quat_mode, nullptr, poses_out, trans_out, stream);
}

int amuse_counter_normal(amuse_ctx* c, uint64_t seed, uint64_t clip_index0, int B, int step, int rng_stream,
                         float* out, void* stream) {
    if (!c || !out) return fail(AMUSE_EINVAL, "NULL argument");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_counter_normal(seed, clip_index0, B, step, rng_stream, out, (hipStream_t)stream, (int)variant_state_dim(c->arch)));
    return 0;
}

int amuse_denoise_step_pose(amuse_ctx* c, const float* x_t, int timestep, const float* con, const float* emo, const float* sty,
                            const int* lengths, int B, int precision, float* eps_out, void* stream) {
    if (int e = check_common(c, con, B, precision)) return e;
    if (!(c->arch & 2)) return fail(AMUSE_ESTATE, "amuse_denoise_step_pose needs a pose-space variant (AMUSE_ARCH_ENC_POSE / _DEC_POSE)");
    if (!x_t || !eps_out) return fail(AMUSE_EINVAL, "x_t / eps_out is NULL");
    if (timestep < 0) return fail(AMUSE_EINVAL, "negative timestep");
    return variant_denoise(c, x_t, &timestep, false, con, emo, sty, lengths, B, precision, eps_out, nullptr, (hipStream_t)stream);
}

int amuse_feats_to_smplx(amuse_ctx* c, const float* feats, int B, int quat_mode, float* poses_out, float* trans_out, void* stream) {
    if (!c) return fail(AMUSE_EINVAL, "ctx is NULL");
    if (!feats || (!poses_out && !trans_out)) return fail(AMUSE_EINVAL, "NULL argument");
    if (B < 1) return fail(AMUSE_EINVAL, "B must be >= 1, got %d", B);
    if (quat_mode != AMUSE_QUAT_P3D && quat_mode != AMUSE_QUAT_LEGACY) return fail(AMUSE_EINVAL, "bad quat_mode %d", quat_mode);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_feats_to_smplx(feats, (size_t)B * kFrames, quat_mode, poses_out, trans_out, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
