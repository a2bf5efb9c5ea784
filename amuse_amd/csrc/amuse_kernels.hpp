// Kernel argument blocks + host-callable launchers (defined in k_*.hip, used by amuse_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace amuse {

// Has a launcher's one-off set-up (hipFuncSetAttribute: the dynamic LDS size is a per-DEVICE function attribute) run on the
// current device?  One bit per device, lock-free: the C ABI allows contexts on several GPUs in one process, and two threads racing
// through a first launch both do the idempotent set-up.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool done(int* dev) {
        int d = 0;
        (void)hipGetDevice(&d);
        *dev = d;
        return (mask.load(std::memory_order_acquire) >> (d & 63)) & 1ull;
    }
    void set(int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

// ---------------------------------------------------------------- sampling loop (k_sampler.hip)
struct SampleArgs {
    const uint4* wstream;     // packed denoiser weights: [4 waves][wave_units + kRing][64] x 16 B (head replicated at the tail)
    uint32_t wave_units;      // 1 KiB units per wave for one pass over the network
    uint32_t wave_units_a, wave_units_b;  // k_sample8: per-step units of a group-A / group-B wave (streams laid out
                              // [4 A waves][units_a + kRing8] then [4 B waves][units_b])
    const float* pvec;        // small fp32 parameters (amuse_dev.hpp PV_* layout)
    const float* time_tok;    // [T][128]  TimestepEmbedding(t_i) + pe[1]
    const float* time_tok_clip;  // [B][128] or null: per-clip time token of a teacher-forced step (diffusion_forward)
    const float* cond_tok;    // [B][S-2][128]  emb_proj_*(cond) + pe[2+n]
    const float* pe0;         // [128] pe[0]
    const float* coef;        // [T][8] scheduler coefficients (amuse_hip.h)
    const float* x_init;      // [B][128] or null
    const float* step_noise;  // [T][B][128] or null
    float* latents_out;       // [B][128] or null
    float* traj_out;          // [T][B][128] or null
    float* eps_out;           // [B][128] or null: eps_hat of the last step
    float* tap_out;           // [11][16][128] or null (tile 0, step 0)
    uint64_t seed;
    uint64_t clip0;
    int B, T, S, G;
    int no_update;            // 1: teacher-forced (no scheduler update)
    unsigned long long* prof_out;  // [4 waves][kProfStamps] s_memtime stamps of step prof_step, or null
    int prof_step;
};
constexpr int kProfStamps = 192;
hipError_t launch_sample(const SampleArgs& a, int precision, hipStream_t stream);
// bf16 throughput kernel with 8 waves per workgroup (k_sampler8.hip); no phase-timeline instrumentation
hipError_t launch_sample8(const SampleArgs& a, hipStream_t stream);
hipError_t launch_sample8h(const SampleArgs& a, hipStream_t stream);   // the same kernel on fp16 operands (AMUSE_PREC_F16, k_sampler8h.hip)
// fp32x 8-wave kernel (k_sampler8x.hip); streams laid out like k_sample8's, in split-fp16 units
hipError_t launch_sample8x(const SampleArgs& a, hipStream_t stream);
constexpr int kRing8 = 32;
constexpr int kRing = 32;                 // weight-stream ring depth (1 KiB units in flight per wave)
// fp32: the skip linear (32 units per wave) is a whole ring revolution.  bf16 (16 units) needs no padding either:
// the decoupled-issue scheme simply leaves half the ring empty for that GEMM (k_sampler.hip).
constexpr int skip_pad_units(int prec) { return prec != 1 ? (kRing - 32 % kRing) % kRing : 0; }
constexpr int kEncPv = 1664;              // encoder-block small params kept in LDS (PV_* up to LN2)
constexpr int kSkipBytes = 32 * 1024;     // U-Net skip stack [4][8][64] f32x4
constexpr int kSampleCombBytes = 4 * 8 * 64 * 16 + 4 * 16 * 8 + 8 * 64 * 16;  // = kCombBytes (amuse_dev.hpp), 41,472 B
constexpr int kSample8LdsBytes = 8 * 8 * 64 * 16 + 8 * 16 * 8 + 4 * 4 * 64 * 16 + (9 * kEncPv + 4 * 128 + 2 * 128) * 4 +
                                 2 * 8 * 64 * 16 + 2 * 128 * 4;  // 163,328 B (layout in k_sampler8.hip)
// k_sample8x: combine matrix | statistics | 2 parameter slots of 7 KiB + tail | skip stack (split operands) | token rows | latent | time token
constexpr int kSample8xLdsBytes = 8 * 8 * 64 * 16 + 8 * 16 * 8 + (2 * 7 * 256 + 4 * 128 + 2 * 128) * 4 + 4 * 8 * 64 * 16 +
                                  2 * 8 * 64 * 16 + 2 * 128 * 4;   // 134,144 B
constexpr int kSampleLdsBytes = kSampleCombBytes + kSkipBytes + (9 * kEncPv + 4 * 128 + 2 * 128) * 4;  // 137,216 B

// ---------------------------------------------------------------- Denoiser(arch = "trans_dec"), latent sample (k_sampler_dec.hip)
// per-layer small parameters of a TransformerDecoderLayer with a multi-token memory: the PV_* layout of amuse_dev.hpp up to
// PV_BLOCK, then the cross-attention's query bias and out_proj bias (its key / value biases live in the hoisted K / V tables)
constexpr int PVX_CQ_B = 1920, PVX_CO_B = 2048, PVX_BLOCK = 2176;
constexpr int PVX_FINAL_W = 9 * PVX_BLOCK, PVX_FINAL_B = PVX_FINAL_W + 128, PVX_TOTAL = PVX_FINAL_B + 128;
// K / V of the memory tokens [time, con, (emo), (sty)] + mem_pos for every layer, hoisted out of the step loop:
//   key 0 (time): tkv[step or clip][9][2][128];  keys 1..ncond: ckv[clip][ncond][9][2][128]
struct MemKV {
    const float* tkv;
    size_t tkv_clip_stride;   // floats between clips (per-clip timesteps: diffusion_forward), 0 = one entry for all clips
    const float* ckv;
    int ncond;                // 1..3
};
struct SampleDecArgs {
    const uint4* wstream;     // [4 waves][wave_units + kRing][64] x 16 B, head replicated at the tail
    uint32_t wave_units;
    const float* pvec;        // PVX_* layout
    const float* pe0;         // query_pos.pe[0] [128]
    MemKV mem;                // mem.tkv = the table of step 0; step s reads mem.tkv + s * tkv_step_stride
    size_t tkv_step_stride;
    const float* coef; const float* x_init; const float* step_noise;
    float* latents_out; float* traj_out; float* eps_out; float* tap_out;   // tap_out [11][16][128]: tile 0, step 0
    uint64_t seed, clip0;
    int B, T, no_update;
};
constexpr int kSampleDecLdsBytes = kSampleCombBytes + PVX_TOTAL * 4;   // combine buffers | all small parameters
hipError_t launch_sample_dec(const SampleDecArgs& a, int precision, hipStream_t stream);

// ---------------------------------------------------------------- one-off prologue kernels (k_misc.hip)
// kv[n][l][0 | 1][:] = W{k,v}_l tok[n] + b{k,v}_l for the 9 layers' cross-attention in_proj (cross_attention.py:331-336)
hipError_t launch_mem_kv(const float* tok, int N, const float* wkv_t /*[9][2][128 in][128 out]*/, const float* bkv /*[9][2][128]*/,
                         float* kv /*[N][9][2][128]*/, hipStream_t stream);
// poses / trans of a [B][300][333] feature sequence (infer_ldm.py:168-173)
hipError_t launch_feats_to_smplx(const float* feats, size_t nrows, int quat_mode, float* poses, float* trans, hipStream_t stream);
// time_tok[i][:] = Linear2(SiLU(Linear1([cos|sin](t_i * freqs)))) + pe1     (embeddings.py:245-322)
hipError_t launch_time_tokens(const int* timesteps_dev, int T, const float* freqs, const float* w1t,
                              const float* b1, const float* w2t, const float* b2, const float* pe1,
                              float* out, hipStream_t stream);
// cond_tok[b][n][:] = Linear(ReLU(z_n[b])) + pe[2+n]                         (denoiser.py:74-79,153-181)
struct CondArgs {
    const float* z[3];     // dev [B][256] for the present conditions, in token order
    const float* wt[3];    // transposed weights [256][128] matching z[n]
    const float* bias[3];  // [128]
    const float* pe;       // query_pos.pe [500][128] (mem_pos.pe for the trans_dec variants' memory tokens)
    float* out;            // [B][ncond][128]
    int B, ncond;
    int pe_base;           // position of the first condition token: 2 behind [latent, time], 1 behind [time] (variants)
};
hipError_t launch_cond_tokens(const CondArgs& a, hipStream_t stream);
// noisy[b] = sa[b] * z0[b] + sb[b] * noise[b]   (DDPMScheduler.add_noise; call site ldm.py:84)
// kind: 0 = fp32 image, 1 = bf16 image, 2 = split-fp16 image (1 KiB units alternate hi = rn16(w), lo = rn16(w - hi)), 3 = fp16 image
hipError_t launch_repack(const float* params, const int* map, void* dst, size_t n, int kind, hipStream_t stream);
hipError_t launch_add_noise(const float* z0, const float* noise, const float* sa, const float* sb, float* out, int B,
                            hipStream_t stream, int nfeat = 128);
hipError_t launch_counter_normal(uint64_t seed, uint64_t clip0, int B, int step, int rng_stream, float* out,
                                 hipStream_t stream, int nfeat = 128);

// ---------------------------------------------------------------- VAE decode (k_vae.hip)
constexpr int kVaeRing = 16;    // k_vae_rows weight ring; the packed decoder stream is padded by this many units
constexpr int kVaeStages = 10;  // stage 0: PE + QKV(0); stage i+1: post-attention of block i (+ QKV(i+1) | final)
struct VaeRowsArgs {
    const uint4* wstream;             // packed decoder weights (per precision)
    uint32_t stage_base[kVaeStages];  // first unit of each stage
    uint32_t stage_units[kVaeStages]; // units per wave in each stage
    const float* pvec;                // decoder small params, PV_* layout
    const float* final_bias;          // [384] final_layer.bias zero-padded
    const float* pe;                  // query_pos_decoder.pe [500][128]
    const float* ca;                  // [B][9][128] cross-attention constant per clip/block
    const int* lengths;               // dev [B] or null
    float* x;                         // [B*300][128] residual stream (in place)
    float* q; float* k; float* v;     // [B][4][300][32]
    const float* attn_o;              // [B*300][128] attention output (heads concatenated)
    float* skip;                      // [4][B*300][128]
    float* feats_out;                 // [B][300][333] or null
    float* poses_out;                 // [B][300][55][3] or null
    float* trans_out;                 // [B][300][3] or null
    int B;
    int stage;                        // 0..9
    int quat_mode;
    int tiles;                        // 16-row tiles launched per clip: 19, or 1 for the encoder's last stage
    // k_vae_rows8x only: block 0's self-attention half is the same for every full-length clip (the decoder's queries are the positional
    // table; the latent enters through the cross-attention): stage 1 with c1_out writes LN1(pe + SA(pe)) [300][128] and stops, stage 1
    // with c1 starts from it (no stage 0, no attention 0, no out_proj / norm1)
    const float* c1;
    float* c1_out;
    // encoder mode only (MotionPrior.encode): rows are [2 distribution tokens | 300 frames], S = 302
    const float* enc_feats;           // [B][300][333] input motion features
    const float* tok;                 // global_motion_token [2][128]
    const float* emb_bias;            // skel_embedding.bias [128]
    float* stats_out;                 // [B][2][128]: encoder.norm of the distribution rows (mu | logvar)
    // Denoiser diffusion_only modes (VAE_MODE_DEN_E / _D): one denoising step.  enc_feats = x_t [B][300][333], emb_bias = pose_embd.bias,
    // final_bias = pose_proj.bias (padded), pe = query_pos.pe, feats_out = eps_hat (nullable)
    int S;                            // DEN_E: rows per clip = npre + 300
    int npre;                         // DEN_E: condition tokens in front of the frames (2..4)
    const float* pre_tok_t;           // DEN_E: time token + pe[0], [128] (pre_tok_t_stride 0) or per clip [B][128]
    size_t pre_tok_t_stride;
    const float* pre_tok_c;           // DEN_E: [B][npre - 1][128] condition tokens + pe[1..]
    MemKV mem;                        // DEN_D: hoisted K / V of the memory tokens (mem.tkv = this step's table)
    const float* coef;                // dev [8]: this step's scheduler row, or null (teacher-forced step: no update)
    float* x_out;                     // [B][300][333] or null: x_{t-1} (may alias enc_feats)
    const float* step_noise;          // [B][300][333] this step's explicit noise, or null -> counter-based
    uint64_t seed, clip0;
    int step;
};
// mode: which network the stages run (k_vae.hip M_*)
constexpr int VAE_MODE_DEC = 0, VAE_MODE_ENC = 1, VAE_MODE_DEN_E = 2, VAE_MODE_DEN_D = 3;
hipError_t launch_vae_rows(const VaeRowsArgs& a, int precision, int mode, hipStream_t stream);
// fp32x row stages without split-K (k_vae_rows8.hip): decode (every stage) and encode (stages 1..9; mode = VAE_MODE_DEC / _ENC below)
hipError_t launch_vae_rows8x(const VaeRowsArgs& a, hipStream_t stream, int mode = 0);

struct VaeAttnArgs {
    const float* q; const float* k; const float* v;  // [B][4][300][32]; q pre-scaled by 1/sqrt(32)
    const int* lengths;                               // dev [B] or null
    float* o;                                         // [B*300][128]
    int B;
    int q_tiles;                                      // query tiles to produce: 19, or 1 (last encoder block)
    int S;                                            // VAE_MODE_DEN_E: rows per clip (302..304)
};
hipError_t launch_vae_attn(const VaeAttnArgs& a, int precision, int mode, hipStream_t stream);
// ---------------------------------------------------------------- fused decode (k_vae_fused.hip): one workgroup per clip
constexpr int kVaeFusedStageUnits = 16;   // the weight stream is consumed in 16 KiB stages (LDS-DMA ring of three)
constexpr int kVaeFusedLdsBytes = 40960 + 3 * 16384 + 2 * 8192 + 5120;   // K/V images of one head | weight ring | block params | ca
struct VaeFusedArgs {
    const uint4* wstream;      // bf16 stream in consumption order, whole stages (amuse_api.hip), shared by the 4 waves
    const float* pvec;         // decoder small params, PV_* layout
    const float* final_bias;   // [384]
    const float* pe;           // query_pos_decoder.pe [500][128]
    const float* ca;           // [B][9][128] cross-attention constant (k_vae_ca); readable 512 B past the end
    const int* lengths;        // dev [B] or null
    uint4* skip;               // [B][4 levels][20 tiles][4 k-pairs][64 lanes] packed bf16 operands of the skip stack
    float* feats_out;          // [B][300][333] or null
    float* poses_out;          // [B][300][55][3] or null
    float* trans_out;          // [B][300][3] or null
    float* tap_out;            // [11][300][128] or null: clip 0's fp32 residual stream after blocks 0..8, after decoder.norm (tests) and - slot 10 -
                               // behind block 0's norm1 (what the library keeps as c1)
    const float* c1;           // [300][128] or null: norm1(PE + SA(PE)) of block 0 for this weight set (full-length clips start from it)
    int B, quat_mode;
    int ablate_attention;      // 1: the instantiation without softmax(Q K^T) V (amuse_debug_set_ablation: timing only, wrong outputs)
};
constexpr size_t kVaeFusedSkipBytesPerClip = 4 * 20 * 4 * 64 * 16;
hipError_t launch_vae_fused(const VaeFusedArgs& a, hipStream_t stream);
hipError_t launch_vae_fusedh(const VaeFusedArgs& a, hipStream_t stream);   // fp16 operands (AMUSE_PREC_F16, k_vae_fusedh.hip)
// fp32x form of the fused decoder (k_vae_fusedx.hip): split-fp16 operands, the staged fp32x path's arithmetic bit for bit; its scratch arrays are the staged path's
struct VaeFusedXArgs {
    const uint4* wstream;      // unit pairs (hi | lo) in consumption order, whole 16 KiB stages (amuse_api.hip)
    const float* pvec;         // decoder small params, PV_* layout
    const float* final_bias;   // [384]
    const float* pe;           // query_pos_decoder.pe [500][128]
    const float* ca;           // [B][9][128] cross-attention constant (k_vae_ca); readable 512 B past the end
    const int* lengths;        // dev [B] or null
    float* skip;               // [4][B * 300][128] fp32 skip stack (the staged path's array)
    float* obuf;               // [B * 300][128] attention outputs of the current block (the staged path's attn_o)
    float* feats_out;          // [B][300][333] or null
    float* poses_out;          // [B][300][55][3] or null
    float* trans_out;          // [B][300][3] or null
    float* tap_out;            // [10][300][128] or null: clip 0's fp32 residual stream after blocks 0..8 and after decoder.norm (tests)
    const float* c1;           // [300][128] or null: norm1(PE + SA(PE)) of block 0 for this weight set (full-length clips start from it)
    float* c1_out;             // [300][128] or null: clip 0 writes that array (behind block 0's norm1) - how the library obtains c1
    int B, quat_mode;
};
hipError_t launch_vae_fusedx(const VaeFusedXArgs& a, hipStream_t stream);
// one fp32x step of the pose-space Denoiser (diffusion_only + trans_enc; denoiser.py:177-187) as ONE persistent workgroup per clip (k_vae_fusedx.hip k_den_fusedx): pose_embd
// of x_t, the condition tokens in front, nine encoder blocks (the decode kernel's two block halves with an encoder layer), encoder.norm, pose_proj, the mask, eps_hat and -
// with coefficients - the scheduler update of x_t in the parity modes' arithmetic (k_vae.hip's last stage).  Scratch: the staged step's attn_o and skip arrays.
struct DenFusedXArgs {
    const uint4* wstream;      // unit pairs in consumption order, whole 16 KiB stages (amuse_variants.hip)
    const float* pvec;         // the variant's small parameters, PV_* layout
    const float* emb_bias;     // pose_embd.bias [128]
    const float* final_bias;   // pose_proj.bias padded to [384]
    const float* pe;           // query_pos.pe [500][128]
    const float* ttok;         // time token + pe[0]: [128] (ttok_stride 0) or per clip [B][128]
    size_t ttok_stride;
    const float* ctok;         // [B][npre - 1][128] condition tokens + pe[1..]
    const float* x_in;         // [B][300][333] x_t
    float* x_out;              // [B][300][333] or null: x_{t-1} (may alias x_in)
    float* eps_out;            // [B][300][333] or null
    const float* coef;         // dev [8]: this step's scheduler row, or null (teacher-forced step: no update)
    const float* step_noise;   // [B][300][333] this step's explicit noise, or null -> counter-based
    const int* lengths;        // dev [B] or null: eps rows of frames >= lengths[b] are zeroed (denoiser.py:187)
    float* obuf;               // [B * S][128] attention outputs of the current block
    float* skip;               // [4][B * S][128] fp32 skip stack
    unsigned long long seed, clip0;
    int step;
    int B, S, npre;            // clips; rows per clip = npre + 300; condition-token rows in front of the frames (2..4)
    // MotionPrior.encode on the same kernel (vae.py:154-214; encode = 1): x_in = motion features, emb_bias = skel_embedding.bias, pe = query_pos_encoder.pe, ttok = the two
    // distribution tokens [2][128] (positions added here), npre = 2, S = 302, lengths = key mask over the frames, eps_out = [B][2][128] encoder.norm of the token rows; no pose_proj
    int encode;
};
hipError_t launch_den_fusedx(const DenFusedXArgs& a, hipStream_t stream);
// ---------------------------------------------------------------- fused pose-space denoiser step (k_den_fused.hip): one workgroup per clip
struct DenFusedArgs {
    const uint4* wstream;      // 16-bit stream in consumption order, whole stages (amuse_variants.hip)
    const float* pvec;         // encoder small params, PV_* layout
    const float* final_bias;   // pose_proj.bias padded to [384]
    const float* emb_bias;     // pose_embd.bias [128]
    const float* pe;           // query_pos.pe [500][128]
    const float* ttok;         // time token + pe[0]: [128] (ttok_stride 0) or one per clip
    size_t ttok_stride;
    const float* ctok;         // [B][npre - 1][128] condition tokens + pe[1..]
    uint4* skip;               // [B][4 levels][20 tiles][4 k-pairs][64 lanes] packed operands of the skip stack
    float* x;                  // [B][300][333]: x_t; overwritten with x_{t-1} when coef is set
    float* eps_out;            // [B][300][333] or null
    const float* coef;         // dev [8]: the step's scheduler row, or null (teacher-forced step)
    const float* step_noise;   // [B][300][333] or null -> counter-based
    const int* lengths;        // dev [B] or null
    uint64_t seed, clip0;
    int step, B, npre;
    int ablate_attention;      // as VaeFusedArgs
};
hipError_t launch_den_fused(const DenFusedArgs& a, hipStream_t stream);
hipError_t launch_den_fusedh(const DenFusedArgs& a, hipStream_t stream);   // fp16 operands (k_den_fusedh.hip)
// feats[row][0:330] = first two rows of R(axis-angle) per joint, feats[row][330:333] = trans   (infer_ldm.py:459-464)
// training step (k_train.hip): the per-device dropout epoch word every mask-drawing kernel reads (0 unless a captured step advances it)
uint32_t* train_epoch_ptr();
// training step (k_train_gemm.hip): out[M][N] = (bias | accumulate: out) + a[M][K] . (tb ? b[N][K]^T : b[K][N]) for the tall fp32 projections it takes
bool train_gemm_tall_takes(long M, long N, long K, bool tb, bool bias);
// ... and for every other shape / transpose of the step (the 333-wide layers, 32- and 160-row projections, weight gradients): the same file's generic kernel
hipError_t launch_train_gemm_any(const float* a, const float* b, const float* bias, float* out, long M, long N, long K, bool ta, bool tb, bool accumulate, hipStream_t stream);
hipError_t launch_train_gemm_tall(const float* a, const float* b, const float* bias, float* out, long M, long N, long K, bool tb, bool accumulate, hipStream_t stream);
hipError_t launch_smplx_to_feats(const float* poses, const float* trans, size_t nrows, float* feats, hipStream_t stream);
// mu = stats[b][0], std = exp(stats[b][1]) ** 0.5, latent = mu + std * eps   (vae.py:209-213)
hipError_t launch_vae_latent(const float* stats, const float* eps, float* mu, float* std, float* latent, int B,
                             hipStream_t stream);

// ca[b][blk][:] = out_proj(v_proj(z[b]))   (cross_attention.py:331-336 with a 1-token memory)
hipError_t launch_vae_ca(const float* z, const float* wv_t /*[9][128][128]*/, const float* bv /*[9][128]*/,
                         const float* wo_t, const float* bo, float* ca, int B, hipStream_t stream);

}  // namespace amuse
