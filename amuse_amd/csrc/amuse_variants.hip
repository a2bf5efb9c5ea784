// Denoiser variants (reference models/latent_diffusion/denoiser.py:64-66,92-131,174-204): host side - state-dict index, weight
// streams, hoisted tables, launch sequences.  Kernels: k_sampler_dec.hip (arch "trans_dec", latent sample: persistent T-step kernel),
// k_vae.hip M_DEN_E / M_DEN_D (diffusion_only: one step = the staged rows / attention kernels at S = 304 / 300), k_misc.hip prologues.
#include "amuse_variants.hpp"

struct amuse_variant {
    // arch DEC: [4 waves][units + kRing][64] per precision (PREC_* index)
    uint4* dec_w[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t dec_units[4] = {0, 0, 0, 0};
    // pose-space archs: staged streams [stage][wave][units] per precision
    uint4* rows_w[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t stage_base[4][kVaeStages];
    uint32_t stage_units[4][kVaeStages];
    uint4* rows8_w = nullptr;                   // fp32x, diffusion_only + trans_enc: stages 1..8 for the row kernel without split-K (k_vae_rows8.hip, ENC form)
    uint32_t rows8_base[kVaeStages];
    uint4* fusedx_w = nullptr;                  // fp32x, diffusion_only + trans_enc: the whole step as ONE stream for the per-clip kernel (k_vae_fusedx.hip k_den_fusedx)
    float* pvec = nullptr;           // PV_* (ENC_POSE) / PVX_* (trans_dec archs) layout
    float* m_pe = nullptr;           // mem_pos.pe [500][128]
    float *wkv_t = nullptr, *bkv = nullptr;       // trans_dec: cross-attention k / v projections [9][2][128 in][128 out], [9][2][128]
    float *emb_bias = nullptr, *final_bias = nullptr;   // pose_embd.bias [128], pose_proj.bias padded to [384]
    float* tkv_sched = nullptr;      // [AMUSE_MAX_STEPS][9][2][128]: K / V of the time memory token per step of the schedule
    // workspaces
    float* ckv = nullptr; size_t ckv_cap = 0;     // [B][ncond][9][2][128]
    float* tkv1 = nullptr; size_t tkv1_cap = 0;   // teacher-forced steps: [1 or B][9][2][128]
    float* ws = nullptr; size_t ws_cap = 0;       // pose stages: x, q, k, v, o, 4 skip levels of [B][304][128]
    uint4* fused_w[2] = {nullptr, nullptr};       // ENC_POSE: the fused step kernel's stream (k_den_fused.hip), bf16 | fp16
    uint4* skip = nullptr; size_t skip_cap = 0;   // its skip stack, clips
    float* tt = nullptr; size_t tt_cap = 0;       // teacher-forced steps: time tokens [1 or B][128] | device copy of the timesteps
};

namespace {
constexpr int kTkv = kLayers * 2 * kD;        // floats of one token's K / V over the nine layers
constexpr int kPoseRowsMax = kFrames + 4;
constexpr size_t kPoseWsPerClip = (size_t)kPoseRowsMax * kD * (1 + 3 + 1 + 4);
constexpr int kPoseChunk = 256;
constexpr int kUpdBitV[4] = {AMUSE_UPD_F32, AMUSE_UPD_BF16, AMUSE_UPD_F32X, AMUSE_UPD_F16};

bool arch_dec(int arch) { return arch & 1; }
bool arch_pose(int arch) { return arch & 2; }

ParamIndex variant_index(int arch) {
    ParamIndex P;
    if (arch_pose(arch)) {
        P.add("pose_embd.weight", 128 * kFeats); P.add("pose_embd.bias", 128);
        P.add("pose_proj.weight", kFeats * 128); P.add("pose_proj.bias", kFeats);
    }
    P.add("time_embedding.linear_1.weight", 128 * 256); P.add("time_embedding.linear_1.bias", 128);
    P.add("time_embedding.linear_2.weight", 128 * 128); P.add("time_embedding.linear_2.bias", 128);
    for (const char* n : {"con", "emo", "sty"}) {
        P.add(std::string("emb_proj_") + n + ".1.weight", 128 * 256);
        P.add(std::string("emb_proj_") + n + ".1.bias", 128);
    }
    P.add("query_pos.pe", 500 * 128); P.add("mem_pos.pe", 500 * 128);
    if (arch_dec(arch)) {   // TransformerDecoder: layers, then norm (cross_attention.py:195-203)
        for (int i = 0; i < kLayers; ++i) dec_layer(P, "decoder.layers." + std::to_string(i));
        P.add("decoder.norm.weight", 128); P.add("decoder.norm.bias", 128);
    } else {
        skip_stack(P, "encoder", false);
    }
    return P;
}
const ParamIndex& index_of(int arch) {
    static const ParamIndex idx[4] = {variant_index(0), variant_index(1), variant_index(2), variant_index(3)};
    return idx[arch & 3];
}
std::string dec_name(int l) { return "decoder.layers." + std::to_string(l); }

// PVX_* parameter vector of the nine TransformerDecoderLayers + decoder.norm
std::vector<float> build_pvec_dec(const Params& P) {
    std::vector<float> pv(PVX_TOTAL, 0.f);
    for (int l = 0; l < kLayers; ++l) {
        float* b = pv.data() + l * PVX_BLOCK;
        const std::string p = dec_name(l);
        fill_block_pvec(b, P, p, true);
        memcpy(b + PVX_CQ_B, P.get(p + ".multihead_attn.in_proj_bias"), 128 * 4);
        memcpy(b + PVX_CO_B, P.get(p + ".multihead_attn.out_proj.bias"), 128 * 4);
    }
    memcpy(pv.data() + PVX_FINAL_W, P.get("decoder.norm.weight"), 128 * 4);
    memcpy(pv.data() + PVX_FINAL_B, P.get("decoder.norm.bias"), 128 * 4);
    return pv;
}

// wave w's units of one decoder layer behind its self-attention, in consumption order (k_sampler_dec.hip decoder_layer;
// k_vae.hip M_DEN_D stage): [self v (latent archs) |] self out_proj slice, cross q, cross out_proj slice, linear1, linear2
void pack_dec_layer(std::vector<uint4>& s, int prec, const Params& P, int l, int w, bool self_value_path) {
    const std::string p = dec_name(l);
    if (self_value_path) pack_gemm(s, prec, P.get(p + ".self_attn.in_proj_weight"), 384, 128, {16 + 2 * w, 16 + 2 * w + 1}, range(0, 8));
    pack_gemm(s, prec, P.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * w, 2 * w + 1});
    pack_gemm(s, prec, P.get(p + ".multihead_attn.in_proj_weight"), 384, 128, {2 * w, 2 * w + 1}, range(0, 8));
    pack_gemm(s, prec, P.get(p + ".multihead_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * w, 2 * w + 1});
    pack_gemm(s, prec, P.get(p + ".linear1.weight"), 512, 128, range(8 * w, 8 * w + 8), range(0, 8));
    pack_gemm(s, prec, P.get(p + ".linear2.weight"), 128, 512, range(0, 8), range(8 * w, 8 * w + 8));
}

int stage_lengths_v(amuse_ctx* c, const int* lengths, int B, hipStream_t st) {
    if (!lengths) return 0;
    bool full = false;
    for (int b = 0; b < B; ++b) {
        if (lengths[b] < 1 || lengths[b] > kFrames) return fail(AMUSE_EINVAL, "lengths[%d] = %d not in 1..300", b, lengths[b]);
        full |= lengths[b] == kFrames;
    }
    // lengths_to_mask sizes the mask by max(lengths) and `sample[~mask.T] = 0` needs it to be 300 (denoiser.py:145,187)
    if (!full) return fail(AMUSE_EINVAL, "max(lengths) must be 300 (the reference's mask indexing fails otherwise)");
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

// condition tokens (+ positions) into c->cond_tok [B][ncond][128]; trans_dec archs: their K / V into v->ckv
int variant_cond(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int* ncond_out, hipStream_t st) {
    amuse_variant* v = c->var;
    CondArgs ca{};
    int n = 0;
    const float* zs[3] = {con, emo, sty};
    for (int i = 0; i < 3; ++i)
        if (zs[i]) { ca.z[n] = zs[i]; ca.wt[n] = c->cond_wt[i]; ca.bias[n] = c->cond_b[i]; ++n; }
    if (int e = ensure(&c->cond_tok, &c->cond_cap, (size_t)B * 3 * kD)) return e;
    // emb_latent = cat(time, con, emo, sty): positions 1.. of query_pos (diffusion_only + trans_enc, denoiser.py:180-181) or of
    // mem_pos (trans_dec, denoiser.py:194)
    ca.pe = arch_dec(c->arch) ? v->m_pe : c->den_pe;
    ca.pe_base = 1;
    ca.out = c->cond_tok; ca.B = B; ca.ncond = n;
    HIP_TRY(launch_cond_tokens(ca, st));
    if (arch_dec(c->arch)) {
        if (int e = ensure(&v->ckv, &v->ckv_cap, (size_t)B * 3 * kTkv)) return e;
        HIP_TRY(launch_mem_kv(c->cond_tok, B * n, v->wkv_t, v->bkv, v->ckv, st));
    }
    *ncond_out = n;
    return 0;
}
const float* time_pe_row(const amuse_ctx* c) { return arch_dec(c->arch) ? c->var->m_pe : c->den_pe; }   // position 0 of the token's table

// one denoising step of a pose-space variant on clips [0, nb) of the given arrays: ten row stages, nine attention launches
struct PoseStep {
    const float* x_in; float* x_out; float* eps_out; const float* coef; const float* step_noise;
    const float* ttok; size_t ttok_stride; const float* tkv; size_t tkv_clip_stride;
    const float* cond_tok; const float* ckv; const int* lengths_dev;
    int ncond, step; uint64_t seed, clip0;
    bool rows8;   // fp32x staged step: stages 1..8 on k_vae_rows8x (decided from the CALL's clip count)
    bool fusedx;  // ... or, where the call's clips fill rounds of the chip, the whole step as ONE per-clip kernel (k_den_fusedx)
};
// kernels of one pose-space Denoiser step: the pin (amuse_set_decode_path) or the plan (amuse_host.hpp plan_step_path), keyed by the CALL's clip count: the fused
// per-clip step kernel of the 16-bit modes (k_den_fused.hip), fp32x's row kernel without split-K for the staged step's stages 1..8 (FUSED) and its per-clip kernel
// (k_den_fusedx, CLIP) where the clips fill rounds of the chip - AMUSE_ARCH_ENC_POSE only; everything else is staged
int step_path_of(amuse_ctx* c, int precision, int B) {
    int path = resolve_path(c->decode_path, plan_step_path(c->arch, precision, B), precision);
    if (c->arch != AMUSE_ARCH_ENC_POSE) path = AMUSE_DECODE_STAGED;
    if (path == AMUSE_DECODE_CLIP && !c->var->fusedx_w) path = AMUSE_DECODE_FUSED;
    if (path == AMUSE_DECODE_FUSED && precision == PREC_F16X2 && !c->var->rows8_w) path = AMUSE_DECODE_STAGED;
    return c->last_plan[3] = path;
}

int pose_step(amuse_ctx* c, const PoseStep& p, int nb, int precision, bool fused, hipStream_t st) {
    amuse_variant* v = c->var;
    const bool dec = arch_dec(c->arch);
    if (fused) {
        DenFusedArgs fa{};
        fa.wstream = v->fused_w[precision == PREC_F16]; fa.pvec = v->pvec; fa.final_bias = v->final_bias; fa.emb_bias = v->emb_bias;
        fa.pe = c->den_pe; fa.ttok = p.ttok; fa.ttok_stride = p.ttok_stride; fa.ctok = p.cond_tok; fa.skip = v->skip;
        fa.eps_out = p.eps_out; fa.coef = p.coef; fa.step_noise = p.step_noise; fa.lengths = p.lengths_dev;
        fa.seed = p.seed; fa.clip0 = p.clip0; fa.step = p.step; fa.B = nb; fa.npre = 1 + p.ncond;
        fa.ablate_attention = c->ablate & 1;
        // the kernel updates its state array in place; a teacher-forced step (no coefficients) only reads it
        if (p.x_out && p.x_out != p.x_in) HIP_TRY(hipMemcpyAsync(p.x_out, p.x_in, (size_t)nb * AMUSE_POSE_STATE * sizeof(float), hipMemcpyDeviceToDevice, st));
        fa.x = p.x_out ? p.x_out : const_cast<float*>(p.x_in);
        HIP_TRY(precision == PREC_F16 ? launch_den_fusedh(fa, st) : launch_den_fused(fa, st));
        return 0;
    }
    const int npre = dec ? 0 : 1 + p.ncond, S = kFrames + npre;
    const size_t rows = (size_t)nb * S;
    float* ws = v->ws;
    VaeRowsArgs ra{};
    ra.wstream = v->rows_w[precision];
    memcpy(ra.stage_base, v->stage_base[precision], sizeof(ra.stage_base));
    memcpy(ra.stage_units, v->stage_units[precision], sizeof(ra.stage_units));
    ra.pvec = v->pvec; ra.final_bias = v->final_bias; ra.pe = c->den_pe; ra.emb_bias = v->emb_bias;
    ra.x = ws; ws += rows * kD;
    ra.q = ws; ws += rows * kD;
    ra.k = ws; ws += rows * kD;
    ra.v = ws; ws += rows * kD;
    float* attn_o = ws; ws += rows * kD;
    ra.attn_o = attn_o;
    ra.skip = ws;
    ra.lengths = p.lengths_dev;
    ra.enc_feats = p.x_in; ra.feats_out = p.eps_out; ra.x_out = p.x_out; ra.coef = p.coef; ra.step_noise = p.step_noise;
    ra.seed = p.seed; ra.clip0 = p.clip0; ra.step = p.step;
    ra.B = nb; ra.tiles = 19; ra.S = S; ra.npre = npre;
    ra.pre_tok_t = p.ttok; ra.pre_tok_t_stride = p.ttok_stride; ra.pre_tok_c = p.cond_tok;
    ra.mem = MemKV{p.tkv, p.tkv_clip_stride, p.ckv, p.ncond};
    VaeAttnArgs aa{};
    aa.q = ra.q; aa.k = ra.k; aa.v = ra.v; aa.lengths = nullptr; aa.o = attn_o; aa.B = nb; aa.q_tiles = 19; aa.S = S;
    const int mode = dec ? VAE_MODE_DEN_D : VAE_MODE_DEN_E;
    VaeRowsArgs r8 = ra;
    if (p.rows8) {   // (fp32x, trans_enc: stages 1..8 are plain encoder-layer stages - the row kernel without split-K, as MotionPrior.encode's)
        r8.wstream = v->rows8_w;
        memcpy(r8.stage_base, v->rows8_base, sizeof(r8.stage_base));
    }
    if (p.rows8 && p.fusedx && v->fusedx_w) {   // the whole step as one persistent workgroup per clip (its scratch: this path's attn_o and skip arrays)
        DenFusedXArgs fx{};
        fx.wstream = v->fusedx_w; fx.pvec = v->pvec; fx.emb_bias = v->emb_bias; fx.final_bias = v->final_bias; fx.pe = c->den_pe;
        fx.ttok = p.ttok; fx.ttok_stride = p.ttok_stride; fx.ctok = p.cond_tok;
        fx.x_in = p.x_in; fx.x_out = p.x_out; fx.eps_out = p.eps_out; fx.coef = p.coef; fx.step_noise = p.step_noise; fx.lengths = p.lengths_dev;
        fx.obuf = attn_o; fx.skip = ra.skip; fx.seed = p.seed; fx.clip0 = p.clip0; fx.step = p.step; fx.B = nb; fx.S = S; fx.npre = npre;
        HIP_TRY(launch_den_fusedx(fx, st));
        return 0;
    }
    for (int stage = 0; stage < kVaeStages; ++stage) {
        ra.stage = stage;
        r8.stage = stage;
        if (p.rows8 && stage >= 1 && stage <= 8) HIP_TRY(launch_vae_rows8x(r8, st, VAE_MODE_ENC));
        else HIP_TRY(launch_vae_rows(ra, precision, mode, st));
        if (stage < kLayers) HIP_TRY(launch_vae_attn(aa, precision, mode, st));
    }
    return 0;
}
int ensure_pose_ws(amuse_variant* v, int chunk, bool fused) {
    if (fused) {
        if (v->skip_cap >= (size_t)chunk) return 0;
        if (v->skip) HIP_TRY(hipFree(v->skip));
        v->skip = nullptr; v->skip_cap = 0;
        HIP_TRY(hipMalloc((void**)&v->skip, (size_t)chunk * kVaeFusedSkipBytesPerClip));
        v->skip_cap = chunk;
        return 0;
    }
    return ensure(&v->ws, &v->ws_cap, (size_t)chunk * kPoseWsPerClip);
}
}  // namespace

size_t variant_param_count(int arch) {
    if (arch < 0 || arch > 3) return 0;
    return index_of(arch).total;
}

int variant_build(amuse_ctx* c, const float* den, int what) {
    if (!c->var) c->var = new amuse_variant();
    amuse_variant* v = c->var;
    const int arch = c->arch;
    const Params D{index_of(arch), den};
    const bool dec = arch_dec(arch), pose = arch_pose(arch);
    if (!pose) {   // arch DEC: the persistent kernel's per-wave streams, one pass over the nine layers per step
        for (int prec = 0; prec < 4; ++prec) {
            if (!(what & kUpdBitV[prec])) continue;
            std::vector<uint4> all;
            size_t per_wave = 0;
            for (int w = 0; w < 4; ++w) {
                std::vector<uint4> s;
                for (int l = 0; l < kLayers; ++l) pack_dec_layer(s, prec, D, l, w, true);
                if (w == 0) per_wave = s.size();
                else if (s.size() != per_wave) return fail(AMUSE_ESTATE, "internal: uneven trans_dec wave streams");
                all.insert(all.end(), s.begin(), s.end());
                all.insert(all.end(), s.begin(), s.begin() + (size_t)kRing * 64);   // ring wrap: tail = head
            }
            v->dec_units[prec] = (uint32_t)(per_wave / 64);
            if (v->dec_units[prec] % kRing) return fail(AMUSE_ESTATE, "internal: a trans_dec step is not whole ring revolutions");
            if (upload(&v->dec_w[prec], all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
        }
    } else {
        // staged streams (k_vae.hip): stage 0 = pose_embd (K = 333 padded to 22 k-tiles, 2 output tiles per wave) + in_proj(0);
        // stage i + 1 = what follows block i's self-attention (+ skip linear, trans_enc) + in_proj(i + 1) | pose_proj (24 tiles, 6 per wave)
        for (int prec = 0; prec < 4; ++prec) {
            if (!(what & kUpdBitV[prec])) continue;
            std::vector<uint4> all;
            for (int st = 0; st < kVaeStages; ++st) {
                v->stage_base[prec][st] = (uint32_t)(all.size() / 64);
                size_t per_wave = 0;
                for (int w = 0; w < 4; ++w) {
                    std::vector<uint4> s;
                    if (st == 0) pack_gemm(s, prec, D.get("pose_embd.weight"), 128, kFeats, {2 * w, 2 * w + 1}, range(0, 22));
                    if (st >= 1) {
                        const int b = st - 1;
                        if (dec) {
                            pack_dec_layer(s, prec, D, b, w, false);
                        } else {
                            pack_outproj_ffn(s, prec, D, blk_name("encoder", b), w);
                            if (b >= 4 && b <= 7) pack_skiplin(s, prec, D, "encoder", b - 4, w);
                        }
                    }
                    if (st < 9) pack_qkv(s, prec, D.get((dec ? dec_name(st) : blk_name("encoder", st)) + ".self_attn.in_proj_weight"), w, false);
                    else pack_gemm(s, prec, D.get("pose_proj.weight"), kFeats, 128, range(6 * w, 6 * w + 6), range(0, 8));
                    if (w == 0) per_wave = s.size();
                    else if (s.size() != per_wave) return fail(AMUSE_ESTATE, "internal: uneven pose-denoiser wave streams");
                    all.insert(all.end(), s.begin(), s.end());
                }
                v->stage_units[prec][st] = (uint32_t)(per_wave / 64);
            }
            all.insert(all.end(), (size_t)kVaeRing * 64, uint4{0, 0, 0, 0});   // the last wave's ring reads past its slice
            if (upload(&v->rows_w[prec], all.data(), all.size() * sizeof(uint4))) return AMUSE_EHIP;
        }
        if (!dec && (what & AMUSE_UPD_F32X)) {   // fp32x: stages 1..8 once more as ONE stream per stage in consumption order (k_vae_rows8.hip's layout, amuse_api.hip)
            std::vector<uint4> s;
            for (int st = 0; st < kVaeStages; ++st) {
                v->rows8_base[st] = (uint32_t)(s.size() / 64);
                if (st == 0 || st == 9) continue;   // (pose_embd / pose_proj + update stay with k_vae_rows)
                const int b = st - 1;
                const std::string p = blk_name("encoder", b);
                pack_gemm(s, PREC_F16X2, D.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
                for (int ch = 0; ch < 16; ++ch) {
                    pack_gemm(s, PREC_F16X2, D.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8));
                    pack_gemm(s, PREC_F16X2, D.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1});
                }
                if (b >= 4 && b <= 7) {
                    const float* wskip = D.get("encoder.linear_blocks." + std::to_string(b - 4) + ".weight");
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
                }
                const float* in_w = D.get(blk_name("encoder", st) + ".self_attn.in_proj_weight");
                for (int grp = 0; grp < 3; ++grp) pack_gemm(s, PREC_F16X2, in_w, 384, 128, range(8 * grp, 8 * grp + 8), range(0, 8));
                if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: rows8 pose-denoiser stream is not whole stages");
            }
            s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
            if (upload(&v->rows8_w, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
        }
        if (!dec && (what & AMUSE_UPD_F32X)) {   // fp32x: the whole step as ONE stream of unit pairs for the per-clip kernel (k_vae_fusedx.hip k_den_fusedx): pose_embd (11 k-pairs x 8
            // output tiles), nine encoder blocks in the fused fp32x decoder's order (amuse_api.hip: skip linear, per head k | v then q, out_proj, the FFN with linear1 one chunk
            // ahead), pose_proj in four quarters of six output tiles
            std::vector<uint4> s;
            pack_gemm(s, PREC_F16X2, D.get("pose_embd.weight"), 128, kFeats, range(0, 8), range(0, 22));
            for (int b = 0; b < 9; ++b) {
                const std::string p = blk_name("encoder", b);
                if (b >= 5) {
                    const float* wskip = D.get("encoder.linear_blocks." + std::to_string(b - 5) + ".weight");
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(0, 8));
                    pack_gemm(s, PREC_F16X2, wskip, 128, 256, range(0, 8), range(8, 16));
                }
                const float* in_w = D.get(p + ".self_attn.in_proj_weight");
                for (int h = 0; h < 4; ++h) {
                    pack_gemm(s, PREC_F16X2, in_w, 384, 128, {8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
                    pack_gemm(s, PREC_F16X2, in_w, 384, 128, {2 * h, 2 * h + 1}, range(0, 8));
                }
                pack_gemm(s, PREC_F16X2, D.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), range(0, 8));
                const auto f1 = [&](int ch) { pack_gemm(s, PREC_F16X2, D.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8)); };
                const auto f2 = [&](int ch) { pack_gemm(s, PREC_F16X2, D.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1}); };
                f1(0);
                for (int ch = 0; ch < 15; ++ch) { f1(ch + 1); f2(ch); }
                f2(15);
            }
            for (int q = 0; q < 4; ++q) pack_gemm(s, PREC_F16X2, D.get("pose_proj.weight"), kFeats, 128, range(6 * q, 6 * q + 6), range(0, 8));
            if (s.size() % ((size_t)16 * 64) != 0) return fail(AMUSE_ESTATE, "internal: fused fp32x pose-denoiser stream is not whole stages");
            s.insert(s.end(), (size_t)2 * 16 * 64, uint4{0, 0, 0, 0});   // the fetch runs two stages ahead
            if (upload(&v->fusedx_w, s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
        }
        if (!dec) {
            // fused step kernel (k_den_fused.hip; bf16 / fp16 operands): ONE stream for the eight waves, in consumption order, cut into
            // stages of kVaeFusedStageUnits units - the fused decoder's layout (amuse_api.hip) with pose_embd in front and encoder blocks
            for (const int p16 : {PREC_BF16, PREC_F16}) {
                if (!(what & kUpdBitV[p16])) continue;
                std::vector<uint4> s;
                const auto pad = [&](int units) { s.insert(s.end(), (size_t)units * 64, uint4{0, 0, 0, 0}); };
                pack_gemm(s, p16, D.get("pose_embd.weight"), 128, kFeats, range(0, 8), range(0, 22));   // 11 k-pairs x 8 output tiles
                pad(8);
                for (int b = 0; b < 9; ++b) {
                    const std::string p = blk_name("encoder", b);
                    if (b >= 5) {   // skip linear ahead of an output block: the x half (k-tiles 0..7), then the popped-skip half
                        const float* wskip = D.get("encoder.linear_blocks." + std::to_string(b - 5) + ".weight");
                        pack_gemm(s, p16, wskip, 128, 256, range(0, 8), range(0, 8));
                        pack_gemm(s, p16, wskip, 128, 256, range(0, 8), range(8, 16));
                    }
                    const float* in_w = D.get(p + ".self_attn.in_proj_weight");
                    for (int h = 0; h < 4; ++h) {   // per head: stage A = k | v tiles per k-pair; stage B = q, out_proj's k-slice
                        pack_gemm(s, p16, in_w, 384, 128, {8 + 2 * h, 8 + 2 * h + 1, 16 + 2 * h, 16 + 2 * h + 1}, range(0, 8));
                        pack_gemm(s, p16, in_w, 384, 128, {2 * h, 2 * h + 1}, range(0, 8));
                        pack_gemm(s, p16, D.get(p + ".self_attn.out_proj.weight"), 128, 128, range(0, 8), {2 * h, 2 * h + 1});
                    }
                    const auto f1 = [&](int ch) { pack_gemm(s, p16, D.get(p + ".linear1.weight"), 512, 128, {2 * ch, 2 * ch + 1}, range(0, 8)); };
                    const auto f2 = [&](int ch) { pack_gemm(s, p16, D.get(p + ".linear2.weight"), 128, 512, range(0, 8), {2 * ch, 2 * ch + 1}); };
                    f1(0); pad(8);
                    for (int ch = 0; ch < 15; ++ch) { f1(ch + 1); f2(ch); }
                    f2(15); pad(8);
                }
                for (int half = 0; half < 2; ++half)   // pose_proj ONCE (24 output tiles in two halves of 48 units): the kernel's last stage holds it in LDS whole
                    pack_gemm(s, p16, D.get("pose_proj.weight"), kFeats, 128, range(12 * half, 12 * half + 12), range(0, 8));
                if (s.size() % ((size_t)kVaeFusedStageUnits * 64) != 0) return fail(AMUSE_ESTATE, "internal: fused denoiser stream is not whole stages");
                pad(2 * kVaeFusedStageUnits);   // the fetch runs two stages ahead
                if (upload(&v->fused_w[p16 == PREC_F16], s.data(), s.size() * sizeof(uint4))) return AMUSE_EHIP;
            }
        }
        std::vector<float> fb(16 * kFeatTiles, 0.f);
        memcpy(fb.data(), D.get("pose_proj.bias"), kFeats * 4);
        if (upload(&v->final_bias, fb.data(), fb.size() * 4) || upload(&v->emb_bias, D.get("pose_embd.bias"), 128 * 4)) return AMUSE_EHIP;
    }
    {
        const std::vector<float> pv = dec ? build_pvec_dec(D) : build_pvec(D, "encoder", false);
        if (upload(&v->pvec, pv.data(), pv.size() * 4)) return AMUSE_EHIP;
    }
    if (dec) {
        std::vector<float> wkv((size_t)kLayers * 2 * kD * kD), bkv((size_t)kLayers * 2 * kD);
        for (int l = 0; l < kLayers; ++l)
            for (int kvi = 0; kvi < 2; ++kvi) {
                const std::string p = dec_name(l) + ".multihead_attn";
                const auto t = transpose(D.get(p + ".in_proj_weight") + (size_t)(1 + kvi) * kD * kD, kD, kD);
                memcpy(wkv.data() + ((size_t)l * 2 + kvi) * kD * kD, t.data(), (size_t)kD * kD * 4);
                memcpy(bkv.data() + ((size_t)l * 2 + kvi) * kD, D.get(p + ".in_proj_bias") + (1 + kvi) * kD, kD * 4);
            }
        if (upload(&v->wkv_t, wkv.data(), wkv.size() * 4) || upload(&v->bkv, bkv.data(), bkv.size() * 4)) return AMUSE_EHIP;
        if (!v->tkv_sched) HIP_TRY(hipMalloc((void**)&v->tkv_sched, (size_t)AMUSE_MAX_STEPS * kTkv * sizeof(float)));
    }
    // what every arch shares with the shipped configuration: positions, timestep frequencies, time-embedding MLP, condition projections
    if (upload(&c->den_pe, D.get("query_pos.pe"), 500 * 128 * 4) || upload(&v->m_pe, D.get("mem_pos.pe"), 500 * 128 * 4)) return AMUSE_EHIP;
    float fr[128];
    for (int k = 0; k < 128; ++k) fr[k] = expf(-logf(10000.f) * (float)k / 128.f);
    if (!c->den_freqs && upload(&c->den_freqs, fr, sizeof(fr))) return AMUSE_EHIP;   // (amuse_set_schedule may have installed the caller's values)
    const auto w1t = transpose(D.get("time_embedding.linear_1.weight"), 128, 256);
    const auto w2t = transpose(D.get("time_embedding.linear_2.weight"), 128, 128);
    if (upload(&c->te_w1t, w1t.data(), w1t.size() * 4) || upload(&c->te_w2t, w2t.data(), w2t.size() * 4) ||
        upload(&c->te_b1, D.get("time_embedding.linear_1.bias"), 512) || upload(&c->te_b2, D.get("time_embedding.linear_2.bias"), 512))
        return AMUSE_EHIP;
    const char* names[3] = {"con", "emo", "sty"};
    for (int n = 0; n < 3; ++n) {
        const auto wt = transpose(D.get(std::string("emb_proj_") + names[n] + ".1.weight"), 128, 256);
        if (upload(&c->cond_wt[n], wt.data(), wt.size() * 4) || upload(&c->cond_b[n], D.get(std::string("emb_proj_") + names[n] + ".1.bias"), 512))
            return AMUSE_EHIP;
    }
    return 0;
}

void variant_destroy(amuse_ctx* c) {
    amuse_variant* v = c->var;
    if (!v) return;
    void* ptrs[] = {v->dec_w[0], v->dec_w[1], v->dec_w[2], v->dec_w[3], v->rows_w[0], v->rows_w[1], v->rows_w[2], v->rows_w[3], v->rows8_w, v->fusedx_w, v->pvec, v->m_pe,
                    v->wkv_t, v->bkv, v->emb_bias, v->final_bias, v->tkv_sched, v->ckv, v->tkv1, v->ws, v->tt, v->fused_w[0], v->fused_w[1], v->skip};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete v;
    c->var = nullptr;
}

int variant_set_schedule(amuse_ctx* c, hipStream_t st) {
    // time token of every step + its position: query_pos.pe[0] in front of the frames, mem_pos.pe[0] as the first memory token
    HIP_TRY(launch_time_tokens(c->d_timesteps, c->T, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2, time_pe_row(c), c->d_time_tok, st));
    if (arch_dec(c->arch)) HIP_TRY(launch_mem_kv(c->d_time_tok, c->T, c->var->wkv_t, c->var->bkv, c->var->tkv_sched, st));
    return 0;
}

int variant_sample(amuse_ctx* c, const float* con, const float* emo, const float* sty, int B, int precision, uint64_t seed,
                   uint64_t clip0, const float* x_init, const float* step_noise, float* out, float* traj_out, hipStream_t st) {
    amuse_variant* v = c->var;
    int ncond = 0;
    if (int e = variant_cond(c, con, emo, sty, B, &ncond, st)) return e;
    if (!arch_pose(c->arch)) {
        SampleDecArgs a{};
        a.wstream = v->dec_w[precision]; a.wave_units = v->dec_units[precision];
        a.pvec = v->pvec; a.pe0 = c->den_pe;
        a.mem = MemKV{v->tkv_sched, 0, v->ckv, ncond};
        a.tkv_step_stride = kTkv;
        a.coef = c->d_coef; a.x_init = x_init; a.step_noise = step_noise;
        a.latents_out = out; a.traj_out = traj_out;
        a.seed = seed; a.clip0 = clip0; a.B = B; a.T = c->T; a.no_update = 0;
        HIP_TRY(launch_sample_dec(a, precision, st));
        return 0;
    }
    // pose-space archs: the state is the [300][333] feature sequence, updated in place in `out` step by step
    const size_t sd = AMUSE_POSE_STATE;
    if (x_init) HIP_TRY(hipMemcpyAsync(out, x_init, (size_t)B * sd * sizeof(float), hipMemcpyDeviceToDevice, st));
    else HIP_TRY(launch_counter_normal(seed, clip0, B, 0, 0, out, st, (int)sd));
    const int spath = step_path_of(c, precision, B);
    const bool fused = is_op16(precision) && spath != AMUSE_DECODE_STAGED;        // k_den_fused
    const bool rows8 = precision == PREC_F16X2 && spath != AMUSE_DECODE_STAGED;   // k_vae_rows8x<ENC> for stages 1..8
    const bool fusedx = precision == PREC_F16X2 && spath == AMUSE_DECODE_CLIP;    // k_den_fusedx
    const int chunk = fused ? B : (B < kPoseChunk ? B : kPoseChunk);
    if (int e = ensure_pose_ws(v, chunk, fused)) return e;
    for (int step = 0; step < c->T; ++step) {
        for (int b0 = 0; b0 < B; b0 += chunk) {
            const int nb = (B - b0) < chunk ? (B - b0) : chunk;
            PoseStep p{};
            p.x_in = out + (size_t)b0 * sd; p.x_out = out + (size_t)b0 * sd; p.eps_out = nullptr;
            p.coef = c->d_coef + (size_t)step * 8;
            p.step_noise = step_noise ? step_noise + ((size_t)step * B + b0) * sd : nullptr;
            p.ttok = c->d_time_tok + (size_t)step * kD; p.ttok_stride = 0;
            p.tkv = arch_dec(c->arch) ? v->tkv_sched + (size_t)step * kTkv : nullptr; p.tkv_clip_stride = 0;
            p.cond_tok = c->cond_tok + (size_t)b0 * ncond * kD;
            p.ckv = v->ckv ? v->ckv + (size_t)b0 * ncond * kTkv : nullptr;
            p.lengths_dev = nullptr;   // the sampling loop passes full lengths (infer_ldm.py:135)
            p.ncond = ncond; p.step = step; p.seed = seed; p.clip0 = clip0 + (uint64_t)b0; p.rows8 = rows8; p.fusedx = fusedx;
            if (int e = pose_step(c, p, nb, precision, fused, st)) return e;
        }
        if (traj_out) HIP_TRY(hipMemcpyAsync(traj_out + (size_t)step * B * sd, out, (size_t)B * sd * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

int variant_denoise(amuse_ctx* c, const float* x_t, const int* timesteps, bool per_clip, const float* con, const float* emo,
                    const float* sty, const int* lengths, int B, int precision, float* eps_out, float* tap_out, hipStream_t st) {
    amuse_variant* v = c->var;
    const int nt = per_clip ? B : 1;
    // per-call scratch: time tokens [nt][128] (+ their K / V) and the device copy of the timesteps
    if (int e = ensure(&v->tt, &v->tt_cap, (size_t)B * (kD + 1))) return e;
    float* ttok = v->tt;                                         // [nt][128]
    int* ts = reinterpret_cast<int*>(ttok + (size_t)B * kD);     // [nt]
    HIP_TRY(hipMemcpyAsync(ts, timesteps, (size_t)nt * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // the host array belongs to the caller
    HIP_TRY(launch_time_tokens(ts, nt, c->den_freqs, c->te_w1t, c->te_b1, c->te_w2t, c->te_b2, time_pe_row(c), ttok, st));
    if (arch_dec(c->arch)) {
        if (int e = ensure(&v->tkv1, &v->tkv1_cap, (size_t)nt * kTkv)) return e;
        HIP_TRY(launch_mem_kv(ttok, nt, v->wkv_t, v->bkv, v->tkv1, st));
    }
    int ncond = 0;
    if (int e = variant_cond(c, con, emo, sty, B, &ncond, st)) return e;
    if (!arch_pose(c->arch)) {
        SampleDecArgs a{};
        a.wstream = v->dec_w[precision]; a.wave_units = v->dec_units[precision];
        a.pvec = v->pvec; a.pe0 = c->den_pe;
        a.mem = MemKV{v->tkv1, per_clip ? (size_t)kTkv : 0, v->ckv, ncond};
        a.tkv_step_stride = 0;
        a.coef = c->d_coef1; a.x_init = x_t; a.eps_out = eps_out; a.tap_out = tap_out;
        a.B = B; a.T = 1; a.no_update = 1;
        HIP_TRY(launch_sample_dec(a, precision, st));
        return 0;
    }
    if (tap_out) return fail(AMUSE_EINVAL, "taps exist for the latent variants only");
    if (int e = stage_lengths_v(c, lengths, B, st)) return e;
    const size_t sd = AMUSE_POSE_STATE;
    const int spath = step_path_of(c, precision, B);
    const bool fused = is_op16(precision) && spath != AMUSE_DECODE_STAGED;        // k_den_fused
    const bool rows8 = precision == PREC_F16X2 && spath != AMUSE_DECODE_STAGED;   // k_vae_rows8x<ENC> for stages 1..8
    const bool fusedx = precision == PREC_F16X2 && spath == AMUSE_DECODE_CLIP;    // k_den_fusedx
    const int chunk = fused ? B : (B < kPoseChunk ? B : kPoseChunk);
    if (int e = ensure_pose_ws(v, chunk, fused)) return e;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0) < chunk ? (B - b0) : chunk;
        PoseStep p{};
        p.x_in = x_t + (size_t)b0 * sd; p.x_out = nullptr; p.eps_out = eps_out + (size_t)b0 * sd; p.coef = nullptr;
        p.ttok = ttok + (per_clip ? (size_t)b0 * kD : 0); p.ttok_stride = per_clip ? kD : 0;
        p.tkv = arch_dec(c->arch) ? v->tkv1 + (per_clip ? (size_t)b0 * kTkv : 0) : nullptr; p.tkv_clip_stride = per_clip ? kTkv : 0;
        p.cond_tok = c->cond_tok + (size_t)b0 * ncond * kD;
        p.ckv = v->ckv ? v->ckv + (size_t)b0 * ncond * kTkv : nullptr;
        p.lengths_dev = lengths ? c->d_lengths + b0 : nullptr;
        p.ncond = ncond; p.rows8 = rows8; p.fusedx = fusedx;
        if (int e = pose_step(c, p, nb, precision, fused, st)) return e;
    }
    return 0;
}
