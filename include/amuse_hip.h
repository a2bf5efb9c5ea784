/*
 * amuse_hip.h - C ABI of libamuse_hip.so: the MI355X (gfx950) implementation of AMUSE's
 * latent-diffusion gesture-sampling hot path.
 *
 * The reference (kiranchhatre/amuse) has NO native/FFI seam on this path: the boundary is the Python
 * object protocol of models/latent_diffusion/infer_ldm.py (PretrainedLPDM_v1).  Each entry point
 * below names the reference interface it replaces (paths relative to the reference root); the
 * ctypes binding a maintainer adds on the reference side is shown in INTEGRATION.md and shipped in
 * amuse_amd/_lib.py.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch/HIP types in signatures (hipStream_t travels as void*)
 *   - every `dev` pointer is DEVICE memory on the ctx's GPU, fp32, row-major, caller-owned;
 *     every `host` pointer is host memory; inputs are never written, outputs are fully overwritten
 *   - return 0 on success, a negative AMUSE_E* code on failure; amuse_last_error() gives the text
 *     (thread-local).  No exceptions cross the ABI.
 *   - one ctx per (GPU, weight set); calls on one ctx must not overlap in time (they share workspace)
 *   - kernels are enqueued on `stream` (NULL = the legacy default stream) and NOT synchronised
 *     before returning, except where stated
 */
#ifndef AMUSE_HIP_H
#define AMUSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Switches.  Kernel choices are API, not environment: amuse_set_decode_path / amuse_set_clips_per_group pin them per context, amuse_plan states the library's own rule.
 * The library reads NO environment variable.  What is left in the whole package (INTEGRATION.md has the same table):
 *   build macro  AMUSE_OP_F16          the declared second compilation of the three 16-bit kernels for fp16 operands (csrc/k_sampler8h.hip, k_vae_fusedh.hip, k_den_fusedh.hip)
 *   build macro  AMUSE_FPROF=1         variant builds only (tools/build_variant.sh): s_memtime phase stamps of the fused per-clip kernels
 *   environment  AMUSE_HIP_LIB         amuse_amd/_lib.py: load another build of this ABI (kernel A/B measurements)
 *   environment  AMUSE_SHARE_GPU=1     amuse_amd/main.py: every --gpus rank stays on --device (two-process tests on a one-GPU box)
 *   environment  AMUSE_BENCH_SHARE_GPU=1   bench.py: --gpus N ranks on one GPU over gloo (the two-rank bench tests on a one-GPU box)
 *   environment  AMUSE_RUN_STAMP, AMUSE_MANIFEST_DIR   launcher -> rank hand-over inside amuse_amd/main.py (not set by users)
 *   environment  AMUSE_TRAIN_FUSED=0, AMUSE_TRAIN_VALIDATE=1, AMUSE_TRAIN_INNER=train   train_gesture: eager layers / torch's distribution checks / the reference's
 *                                      train-mode inner sampler (amuse_amd/train_ops.py, train_gesture.py)
 * The ~30 environment switches and 51 -DAMUSE_* macros of rounds 1-5 (A/B residue) were retired in round 6: tools/probes/retired_switches/. */
#define AMUSE_ABI_VERSION 5   /* 5: amuse_plan / amuse_debug_last_plan (the launch plan); the process-wide environment overrides of kernel choices are gone;
                                 4: amuse_train_* (training-step glue kernels);
                                 3: Denoiser variants (amuse_create_arch, AMUSE_ARCH_*, amuse_denoise_step_pose, amuse_feats_to_smplx);
                                 2: AMUSE_PREC_F32X / _F16, AMUSE_UPD_F32X / _F16; tile-major amuse_debug_gemm; clips per group 1..5 */

/* architecture the kernels are specialised for (configs/diff_latent_v2.json:23-47,
 * configs/prior_emotional_fing.json:6-20, configs/base_new.json "train_pose_framelen") */
#define AMUSE_D_MODEL 128
#define AMUSE_N_HEADS 4
#define AMUSE_FF 512
#define AMUSE_N_LAYERS 9
#define AMUSE_COND_DIM 256
#define AMUSE_N_FRAMES 300
#define AMUSE_N_JOINTS 55
#define AMUSE_N_FEATS 333
#define AMUSE_DENOISER_PARAMS 2192384u /* 130 tensors, state-dict order of Denoiser (denoiser.py:16-133) */
#define AMUSE_PRIOR_PARAMS 4643277u    /* 297 tensors, state-dict order of MotionPrior (vae.py:24-146) */
#define AMUSE_MAX_STEPS 1000

/* Denoiser variants (models/latent_diffusion/denoiser.py:64-66,92-131,174-204): `arch` and `diffusion_only` of
 * configs/diff_latent_v2.json "arch_denoiser".
 *   ENC       arch "trans_enc", latent sample: the shipped configuration - 5 tokens [latent, time, con, emo, sty], skip encoder
 *   DEC       arch "trans_dec", latent sample: tgt = the latent as ONE token (+ query_pos), memory = [time, con, emo, sty]
 *             (+ mem_pos); 9 TransformerDecoderLayer.forward_post (cross_attention.py:323-345: self-attention, cross-attention
 *             onto the 2..4 memory tokens, FFN) + decoder.norm
 *   ENC_POSE  diffusion_only + "trans_enc": the sample is the 300 x 333 pose sequence; [time, con, emo, sty | pose_embd(frames)]
 *             = S = 304 tokens through the skip encoder (NO key mask, denoiser.py:182), pose_proj of the frame rows
 *   DEC_POSE  diffusion_only + "trans_dec": tgt = pose_embd(frames) (S = 300), memory as DEC, pose_proj
 * The per-clip state of the sampling loop is AMUSE_D_MODEL floats for the latent variants and AMUSE_POSE_STATE = 300 * 333 for
 * the pose-space ones (amuse_state_dim); every `x` / `eps` / noise array below is [B][state_dim]. */
enum { AMUSE_ARCH_ENC = 0, AMUSE_ARCH_DEC = 1, AMUSE_ARCH_ENC_POSE = 2, AMUSE_ARCH_DEC_POSE = 3 };
#define AMUSE_POSE_STATE (AMUSE_N_FRAMES * AMUSE_N_FEATS)
#define AMUSE_DENOISER_PARAMS_DEC 2657536u        /* 176 tensors */
#define AMUSE_DENOISER_PARAMS_ENC_POSE 2278093u   /* 134 tensors: pose_embd, pose_proj first (denoiser.py:64-66) */
#define AMUSE_DENOISER_PARAMS_DEC_POSE 2743245u   /* 180 tensors */

enum { AMUSE_OK = 0, AMUSE_EINVAL = -1, AMUSE_EHIP = -2, AMUSE_ENOMEM = -3, AMUSE_ESTATE = -4 };

/* arithmetic of the MFMA GEMMs.  F32: fp32 weights/operands (v_mfma_f32_16x16x4_f32, exact fp32
 * FMA chains) - the parity mode.  BF16: bf16 weights + bf16-rounded operands, fp32 accumulate,
 * fp32 residual stream / LayerNorm / softmax / scheduler state - the throughput mode.
 * F32X: the fast parity mode of the sampling loop - every GEMM operand (weights on the host, activations in registers)
 * is split into two fp16 pieces, x = hi + lo (22 significand bits), and a product is three v_mfma_f32_16x16x32_f16
 * (Wh.xh + Wh.xl + Wl.xh) accumulated in fp32; softmax, LayerNorm, erf GELU and the scheduler update are the F32 code.
 * Holds the F32 mode's parity bars (eps_hat <= 1e-5 against the reference's modules) at a fraction of its step time.
 * Range: GEMM operands pass through fp16, so activations and weights must stay below 65504 in magnitude (beyond that the hi
 * piece is infinite) - three orders of magnitude above what this network carries (LayerNorm'd rows, |weights| < 1, latents up to
 * ~150 under DDPM); small values lose nothing: lo pieces below 2^-14 are fp16 subnormals, which the MI355X MFMA keeps.
 * amuse_vae_decode / amuse_vae_encode run the same split arithmetic in their staged kernels (k_vae.hip PREC_F16X2).
 * F16: the throughput mode on fp16 instead of bf16 operands - the BF16 mode's kernels (8-wave sampler, fused decoder) built for
 * v_mfma_f32_16x16x32_f16 / v_cvt_pk_f16_f32: same speed, same bytes, 11 significand bits instead of 8 - about an eighth of the BF16
 * mode's drift against F32.  Range as for F32X.  amuse_vae_decode / amuse_vae_encode: the BF16 mode's kernels
 * (fused per-clip decoder, staged rows / attention kernels) in their fp16 instantiations, chosen by the same rule. */
enum { AMUSE_PREC_F32 = 0, AMUSE_PREC_BF16 = 1, AMUSE_PREC_F32X = 2, AMUSE_PREC_F16 = 3 };

/* matrix -> quaternion convention of the axis-angle epilogue (infer_ldm.py:172):
 * P3D   = pytorch3d >= 0.5 candidate selection, no sign standardisation (what the reference's
 *         committed outputs show: |axis-angle| up to 4.69 > pi);
 * LEGACY = the snapshot vendored at models/diffusion/utils/rotation_conversions.py:97-119 (q_w >= 0). */
enum { AMUSE_QUAT_P3D = 0, AMUSE_QUAT_LEGACY = 1 };

typedef struct amuse_ctx amuse_ctx;

/* One denoising schedule = what diffusers' scheduler.set_timesteps + scheduler.step need
 * (infer_ldm.py:116-125,142-147,160-161).  Per step i the update applied in-kernel is
 *   x0 = (x - sb*eps) / sa ; if (clip > 0) x0 = clamp(x0, -clip, clip)
 *   x' = c0*x0 (+ cx*x) (+ ce*eps) (+ sigma*z)
 * coef[i] = { sb, sa, c0, cx, ce, sigma, clip, 0 }.  DDIM: cx = 0; DDPM: ce = 0.  Terms whose
 * coefficient is exactly 0 are skipped.  amuse_amd/scheduler.py builds these tables. */
typedef struct {
    int n_steps;            /* T, 1..AMUSE_MAX_STEPS */
    const int* timesteps;   /* host [T]: the integer timestep fed to the time embedding at step i */
    const float* coef;      /* host [T][8] */
    const float* freqs;     /* host [128] or NULL: exp(-ln(1e4) k/128), k = 0..127 (embeddings.py:262-267).
                               The reference evaluates this with torch.exp; a caller that wants bit-equal
                               time embeddings passes torch's values, NULL = libm expf. */
} amuse_schedule;

/* Replaces PretrainedLPDM_v1.setup()'s model construction + weight load (infer_ldm.py:66-109) and
 * PretrainedVAE.load_model (infer_pretrained_vae.py:13-49).  `denoiser_params` / `prior_params`
 * are HOST fp32 arrays holding every state-dict tensor, concatenated in state-dict order (sizes
 * must equal AMUSE_DENOISER_PARAMS / AMUSE_PRIOR_PARAMS).  Packs both precisions' MFMA-fragment
 * weight streams and uploads them.  Returns NULL on failure. */
amuse_ctx* amuse_create(int device, const float* denoiser_params, size_t n_denoiser,
                        const float* prior_params, size_t n_prior);
/* The same for a Denoiser variant: `denoiser_params` holds the state dict of Denoiser(arch, diffusion_only) in its own
 * registration order (amuse_denoiser_param_count(arch) floats; amuse_amd/weights.py denoiser_param_spec lists the keys).
 * prior_params may be NULL (n_prior 0) for the pose-space variants, which never decode (infer_ldm.py:165,177);
 * amuse_vae_* then fail with AMUSE_ESTATE.  arch AMUSE_ARCH_ENC == amuse_create. */
amuse_ctx* amuse_create_arch(int device, int arch, const float* denoiser_params, size_t n_denoiser,
                             const float* prior_params, size_t n_prior);
size_t amuse_denoiser_param_count(int arch);   /* 0 for an unknown arch */
int amuse_arch(const amuse_ctx* ctx);
size_t amuse_state_dim(const amuse_ctx* ctx);  /* floats per clip of the sampled state: 128 or AMUSE_POSE_STATE */
/* New weights into an existing context (same architecture): re-packs and overwrites the weight streams in place.
 * Replaces what the reference gets for free from sharing nn.Module parameters between its training step and the
 * in-loop sampler (scripts/trainer.py:411-415: ldm.diffusion_backward + prior.decode on the weights just stepped).
 * Either array may be NULL (left as is).  `what` limits the host-side packing to what the caller will run:
 * AMUSE_UPD_F32 | AMUSE_UPD_BF16 | AMUSE_UPD_F32X | AMUSE_UPD_F16 = the weight streams of that precision, AMUSE_UPD_ENCODER =
 * MotionPrior.encode's streams too.  Small parameters (biases, LayerNorm, embeddings) are always replaced, so after a partial update only the
 * re-packed precision is valid - running the other one mixes old matrices with new vectors.  Synchronises `stream` first; after a denoiser update the
 * schedule must be set again (the time-token table is a function of the time-embedding weights). */
enum { AMUSE_UPD_F32 = 1, AMUSE_UPD_BF16 = 2, AMUSE_UPD_ENCODER = 4, AMUSE_UPD_F32X = 8, AMUSE_UPD_F16 = 16, AMUSE_UPD_ALL = 31 };
int amuse_update_weights(amuse_ctx* ctx, const float* denoiser_params, size_t n_denoiser, const float* prior_params,
                         size_t n_prior, int what, void* stream);
/* The same from DEVICE arrays (fp32, state-dict order, as above), stream-ordered on `stream` with no host round trip: every packed
 * image is a gather of the parameters, so a kernel re-packs (and rounds to bf16) in place.  The gather maps are built on the first
 * call.  This is what train_gesture's in-loop sampler uses every iteration (amuse_amd/train_gesture.py): the host path costs
 * 24-26 ms per call.  Results are bitwise those of amuse_update_weights on the same values; the schedule stays set (its time-token
 * table is rebuilt on the stream). */
int amuse_update_weights_device(amuse_ctx* ctx, const float* denoiser_params_dev, const float* prior_params_dev, int what,
                                void* stream);
void amuse_destroy(amuse_ctx* ctx);
const char* amuse_last_error(void);
int amuse_abi_version(void);

/* Replaces diffusers.DDIMScheduler(...).set_timesteps(N) (infer_ldm.py:116-123,143-144) and hoists
 * Timesteps + TimestepEmbedding (embeddings.py:245-322; denoiser.py:146-149) out of the loop: builds
 * the [T,128] time-token table on the GPU.  Synchronises `stream`. */
int amuse_set_schedule(amuse_ctx* ctx, const amuse_schedule* sched, void* stream);

/* Replaces the timestep loop of PretrainedLPDM_v1.diffusion_backward (infer_ldm.py:137-161):
 * initial noise, T x { Denoiser.forward, scheduler.step }.
 *   con/emo/sty  dev [B][256]; emo and/or sty may be NULL (token dropped, denoiser.py:159-171)
 *   x_init       dev [B][128] or NULL -> counter-based N(0,1) from (seed, clip_index0 + b)
 *   step_noise   dev [T][B][128] or NULL -> counter-based; only read at steps with sigma != 0
 *   latents_out  dev [B][128]  final latents
 *   traj_out     dev [T][B][128] or NULL: latent after every step (tests)
 * Uses the schedule set by amuse_set_schedule.  Denoiser variants (amuse_create_arch): 128 reads amuse_state_dim(ctx) in every
 * shape above - the pose-space variants sample the [300][333] feature sequence itself, lengths all 300 as the reference's
 * loop passes them (infer_ldm.py:135). */
int amuse_sample(amuse_ctx* ctx, const float* con, const float* emo, const float* sty, int B,
                 int precision, uint64_t seed, uint64_t clip_index0, const float* x_init,
                 const float* step_noise, float* latents_out, float* traj_out, void* stream);

/* Replaces one Denoiser.forward call (denoiser.py:135-204) - teacher-forced single step for tests:
 * eps_out[B][128] = eps_hat(x_t, timestep, con, emo, sty).  Does not need a schedule.
 * tap_out (dev, nullable, [11][16][128]): token rows of the first clip tile after token assembly
 * (slot 0), after each of the 9 blocks (slots 1..9) and after the final LayerNorm (slot 10). */
int amuse_denoise_step(amuse_ctx* ctx, const float* x_t, int timestep, const float* con,
                       const float* emo, const float* sty, int B, int precision, float* eps_out,
                       float* tap_out, void* stream);

/* The pose-space variants' teacher-forced step with the `lengths` argument of Denoiser.forward (denoiser.py:135-145,187,199):
 * eps rows of frames >= lengths[b] are zeroed (`sample[~mask.T] = 0`); the padded frames are still attended (the reference
 * passes no key mask).  lengths host [B] or NULL (= all 300).  x_t, eps_out dev [B][300][333]. */
int amuse_denoise_step_pose(amuse_ctx* ctx, const float* x_t, int timestep, const float* con, const float* emo,
                            const float* sty, const int* lengths, int B, int precision, float* eps_out, void* stream);

/* Replaces the arithmetic of LatentDiffusionModel.diffusion_forward (models/latent_diffusion/ldm.py:71-97), the
 * training-time twin of the sampling step, in eval semantics (no dropout): noisy = sqrt_ab * z0 + sqrt_1m_ab * noise
 * (DDPMScheduler.add_noise) and noise_pred = Denoiser(noisy, timesteps, cond) with ONE TIMESTEP PER CLIP.
 *   z0, noise              dev [B][128]
 *   timesteps              host [B]   (torch.randint(0, num_train_timesteps, (bsz,)) in the reference)
 *   sqrt_ab, sqrt_1m_ab    host [B]   sqrt(alphas_cumprod[t_b]), sqrt(1 - alphas_cumprod[t_b])
 *   noisy_out (nullable), noise_pred_out   dev [B][128] */
int amuse_diffusion_forward(amuse_ctx* ctx, const float* z0, const float* noise, const int* timesteps,
                            const float* sqrt_ab, const float* sqrt_1m_ab, const float* con,
                            const float* emo, const float* sty, int B, int precision, float* noisy_out,
                            float* noise_pred_out, void* stream);

/* Replaces PretrainedVAE.get_motion -> MotionPrior.decode (infer_pretrained_vae.py:58-62,
 * vae.py:216-278) plus the 6D -> matrix -> axis-angle conversion (infer_ldm.py:168-173).
 *   z          dev [B][128]
 *   lengths    host [B] or NULL (= all 300): frames >= length are masked as keys and zeroed
 *   feats_out  dev [B][300][333] or NULL (the raw decoder features)
 *   poses_out  dev [B][300][55][3], trans_out dev [B][300][3] (either may be NULL) */
int amuse_vae_decode(amuse_ctx* ctx, const float* z, const int* lengths, int B, int precision,
                     int quat_mode, float* feats_out, float* poses_out, float* trans_out,
                     void* stream);

/* Replaces PretrainedVAE.get_latent -> MotionPrior.encode (infer_pretrained_vae.py:51-56,
 * vae.py:154-214; call site infer_ldm.py:465): features -> skel_embedding, two distribution tokens
 * prepended, learned PE, 9-block skip-transformer encoder with key-padding mask; mu / logvar are the
 * two distribution rows, std = exp(logvar) ** 0.5, latent = mu + std * eps (Normal.rsample).
 *   feats      dev [B][300][333] motion features (6D rotations | translation)
 *   lengths    host [B] or NULL (= all 300): frames >= length are masked as attention keys
 *   eps        dev [B][128] standard-normal draw for rsample, or NULL (latent_out = mu)
 *   mu_out, std_out, latent_out   dev [B][128], each nullable (at least one required) */
int amuse_vae_encode(amuse_ctx* ctx, const float* feats, const int* lengths, int B, int precision,
                     const float* eps, float* mu_out, float* std_out, float* latent_out,
                     void* stream);

/* Replaces the motion preparation of PretrainedLPDM_v1._loader_helper_v1 (infer_ldm.py:459-464):
 * SMPL-X axis-angle -> rotation matrix -> 6D (first two rows), concatenated with the translation.
 *   poses dev [B][300][55][3], trans dev [B][300][3]  ->  feats_out dev [B][300][333] */
int amuse_smplx_to_feats(amuse_ctx* ctx, const float* poses, const float* trans, int B,
                         float* feats_out, void* stream);

/* Replaces PretrainedLPDM_v1.diffusion_backward end to end (infer_ldm.py:130-178):
 * amuse_sample followed by amuse_vae_decode on the final latents. */
int amuse_diffusion_backward(amuse_ctx* ctx, const float* con, const float* emo, const float* sty,
                             int B, int precision, int quat_mode, uint64_t seed,
                             uint64_t clip_index0, const float* x_init, const float* step_noise,
                             float* latents_out, float* poses_out, float* trans_out, void* stream);

/* The output conversion of PretrainedLPDM_v1.diffusion_backward alone (infer_ldm.py:168-173): features (6D rotations |
 * translation) -> SMPL-X axis-angle.  feats dev [B][300][333] -> poses_out dev [B][300][55][3], trans_out dev [B][300][3].
 * amuse_diffusion_backward of a pose-space variant = amuse_sample + this (the sampled state IS the feature sequence; the
 * reference itself raises at infer_ldm.py:177 where its decode-free branch would be). */
int amuse_feats_to_smplx(amuse_ctx* ctx, const float* feats, int B, int quat_mode, float* poses_out, float* trans_out,
                         void* stream);

/* The build's counter-based normal generator, exposed for tests: out[B][state_dim] for clips
 * clip_index0..+B, `step`, stream 0 (initial latent) or 1 (ancestral noise).  Element e of a clip's state is draw e % 4 of
 * Philox counter (clip, step, e / 4, stream). */
int amuse_counter_normal(amuse_ctx* ctx, uint64_t seed, uint64_t clip_index0, int B, int step,
                         int rng_stream, float* out, void* stream);

/* Diagnostics: runs amuse_sample's kernel with s_memtime stamps taken by the waves of workgroup 0
 * during step `prof_step`; stamps_out dev 768 x uint64 (unused entries 0).
 *   fp32 (4-wave kernel): [4 waves][192] - step start, then per
 *     block: block start, in_proj, attention, out_proj, combine 1, LN1, four FFN quarters, combine 2,
 *     LN2; finally scheduler update.
 *   bf16 (8-wave kernel): [8 waves][96] - step start, then per block: block start (after the skip linear),
 *     attention phase, out_proj combine, FFN, linear2 combine; finally scheduler update.
 * Used by tools/gpu_phase_profile*.py to build profiles/. */
int amuse_profile_sample(amuse_ctx* ctx, const float* con, const float* emo, const float* sty, int B,
                         int precision, int prof_step, unsigned long long* stamps_out, void* stream);

/* Clips per workgroup tile in the sampling kernels: 0 = auto (amuse_plan's clips_per_group for the CALL's clip count), else 1..5; the value used is
 * clamped to 16 / S (S = 5, 4 or 3 tokens per clip: at most 3, 4 or 5 clips share the 16 rows of a tile).
 * Results are bitwise reproducible across launches / shards that use the same value and start at multiples of it
 * (a clip's slot inside its tile decides the rounding of its attention sums); amuse_amd/shard.py applies that rule. */
int amuse_set_clips_per_group(amuse_ctx* ctx, int g);

/* Which kernels amuse_vae_decode (and amuse_diffusion_backward), amuse_vae_encode and the pose-space Denoiser's step use.  Every mode but fp32 has more
 * than one kernel family for MotionPrior.decode (vae.py:216-278); they compute the same function and differ in summation order only (fp32x: 1.5e-6 on
 * features of magnitude 3):
 *   STAGED  the row / attention launches of csrc/k_vae.hip (one 16-row tile per workgroup, split-K); the only family of the fp32 mode.
 *   FUSED   bf16 / fp16: the fused per-clip kernel (csrc/k_vae_fused.hip: one persistent workgroup per clip, residual stream in registers, K/V of the
 *           current head in LDS); fp32x: the row kernel without split-K (csrc/k_vae_rows8.hip: a tile per wave, weights through LDS once per workgroup).
 *   CLIP    fp32x: one persistent workgroup per clip in the parity arithmetic (csrc/k_vae_fusedx.hip: q / k / v never leave the CU); FUSED elsewhere.
 *   AUTO    what amuse_plan returns for the CALL's clip count: FUSED from 64 clips, in fp32x CLIP where the clips fill rounds of the chip's 256 CUs.
 * A sharded job pins the WHOLE job's choice (amuse_plan on the job's total) on every shard, so shards reproduce the single-GPU bits. */
enum { AMUSE_DECODE_AUTO = 0, AMUSE_DECODE_STAGED = 1, AMUSE_DECODE_FUSED = 2, AMUSE_DECODE_CLIP = 3 };
int amuse_set_decode_path(amuse_ctx* ctx, int path);

/* The launch plan of a JOB - the single statement of the library's kernel-choice rules (csrc/amuse_host.hpp plan_*); needs no GPU and no context.
 * For a job of clips_total clips of `tokens` tokens each (5 = latent + time + content + emotion + style; 4 / 3 with emotion / style dropped), Denoiser
 * variant `arch`, mode `precision`:
 *   *clips_per_group  clips per 16-row tile of the latent trans_enc sampler (1 for the other variants): what amuse_sample takes on AUTO for a call of
 *                     that many clips, and the alignment of shard starts (amuse_set_clips_per_group on every shard)
 *   *decode_path / *encode_path / *step_path   AMUSE_DECODE_STAGED / _FUSED / _CLIP: what amuse_vae_decode / amuse_vae_encode / one pose-space Denoiser
 *                     step take on AUTO for a call of that many clips (amuse_set_decode_path with the job's decode_path on every shard)
 * Any output pointer may be NULL.  amuse_debug_last_plan reports what the last amuse_sample / amuse_vae_decode / amuse_vae_encode / pose-space step of a
 * context ACTUALLY took (0 = none yet); tests/test_plan_cpu.py sweeps 1..8192 clips and holds the two equal. */
int amuse_plan(int arch, int precision, int clips_total, int tokens, int* clips_per_group, int* decode_path, int* encode_path, int* step_path);
int amuse_debug_last_plan(const amuse_ctx* ctx, int* clips_per_group, int* decode_path, int* encode_path, int* step_path);

/* ------------------------------------------------------------------------------------------------
 * Audio front-end: replaces PretrainedLPDM_v1.process_single_seq
 * (models/latent_diffusion/infer_ldm.py:180-193) = torchaudio.compliance.kaldi.fbank -> zero-pad / crop to 1024
 * frames -> (x - mean) / (2 std) -> Pretrained_AST_EVP.get_features (models/audio/infer_pretrained_ast_evp.py:42-46)
 * -> AST_EVP.eval_func (AST_EVP.py:84-90): three ASTModel encoders (audio_main_new.py:174-204), 'feature' of each.
 * bf16 GEMM / attention operands, fp32 accumulation and residual stream.
 *
 * Parameters: one flat fp32 array per encoder holding the forward-pass tensors of ASTModel in this order (timm 0.4.5
 * DistilledVisionTransformer names): v.cls_token, v.dist_token, v.pos_embed [1214][768], v.patch_embed.proj.weight
 * [768][1][16][16], .bias, then for blocks 0..11: norm1.weight, norm1.bias, attn.qkv.weight [2304][768], attn.qkv.bias,
 * attn.proj.weight, attn.proj.bias, norm2.weight, norm2.bias, mlp.fc1.weight [3072][768], mlp.fc1.bias,
 * mlp.fc2.weight [768][3072], mlp.fc2.bias; v.norm.weight, v.norm.bias, feature_head.0.weight, .bias,
 * feature_head.1.weight [256][768], .bias   (AMUSE_AST_PARAMS floats). */
typedef struct amuse_audio_ctx amuse_audio_ctx;
#define AMUSE_AST_PARAMS 86385664u
#define AMUSE_AUDIO_CON 0
#define AMUSE_AUDIO_EMO 1
#define AMUSE_AUDIO_STY 2
/* mel_banks host [128][257], window host [400] (kaldi fbank tables: amuse_amd/audio.py builds them);
 * norm_mean / norm_std = configs/base_new.json wav_dtw_mfcc.dataset_mean / dataset_std;
 * frame_based_feats = wav_dtw_mfcc.frame_based_feats (mean over patch tokens) or 0 ((cls + dist) / 2). */
amuse_audio_ctx* amuse_audio_create(int device, const float* con_params, const float* emo_params,
                                    const float* sty_params, size_t n_each, const float* mel_banks,
                                    const float* window, float norm_mean, float norm_std,
                                    int frame_based_feats);
void amuse_audio_destroy(amuse_audio_ctx* ctx);
/* waves dev [B][n_samples] fp32 (16 kHz mono)  ->  fbank_out dev [B][1024][128], normalised and padded */
int amuse_audio_fbank(amuse_audio_ctx* ctx, const float* waves, int n_samples, int B, float* fbank_out,
                      void* stream);
/* One encoder on prepared fbanks: feat_out dev [B][256]; hidden_out (nullable) dev [B][1214][768] receives the
 * fp32 residual stream after block `tap_block` (0..11) for tests. */
int amuse_audio_encode(amuse_audio_ctx* ctx, int which, const float* fbank, int B, float* feat_out,
                       float* hidden_out, int tap_block, void* stream);
/* process_single_seq for B waveforms: con_out / emo_out / sty_out dev [B][256] (each nullable).  Stream-ordered on
 * `stream`; the three encoders run concurrently (per chunk of 32 clips) on two context-owned side streams, forked from
 * and joined back into `stream` with events (results identical to amuse_audio_encode one encoder at a time). */
int amuse_audio_features(amuse_audio_ctx* ctx, const float* waves, int n_samples, int B, float* con_out,
                         float* emo_out, float* sty_out, void* stream);

/* The front-end's GEMM kernel in isolation, for tests and tools/gpu_gemm_bench.py: out = A . W^T + bias.
 * The front-end keeps every GEMM operand TILE-MAJOR in HBM (16-row x 32-feature tiles of 64 lanes x 8 elements, a bf16
 * tile being one MFMA fragment; amuse_amd/csrc/amuse_audio.hpp): A dev bf16 tile-major [M rounded up to 128][K], W dev
 * bf16 in the kernel's packed fragment order (amuse_audio_api.hip pack_w), N % 256 == 0, K % 64 == 0;
 * epi 0: out dev bf16 tile-major [M rounded up to 128][N], epi 3: the same in fp32. */
int amuse_debug_gemm(const void* A, const void* W, const float* bias, int M, int N, int K, int epi,
                     void* out, void* stream);
/* Row-major <-> tile-major copies for the above (F % 32 == 0).  what 0: bf16 row-major [M][F] -> tile-major [M rounded up
 * to 128][F], pad rows zeroed; 1: bf16 tile-major -> row-major [M][F]; 2: fp32 tile-major -> row-major [M][F]. */
int amuse_debug_tile(const void* src, void* dst, int M, int F, int what, void* stream);

/* Debugging taps of the FUSED bf16 decode kernel (k_vae_fused.hip), for tests: while tap_out (dev [11][300][128] fp32) is set,
 * every fused amuse_vae_decode launch writes clip 0's residual stream after decoder blocks 0..8 (slots 0..8), after
 * decoder.norm (slot 9) and behind block 0's norm1 (slot 10: norm1(PE + SA(PE)), the per-weight-set constant full-length clips start
 * from) there, through a separate instantiation of the kernel; NULL switches the taps off again. */
int amuse_debug_set_decode_tap(amuse_ctx* ctx, float* tap_out);

/* Timing ablation of the fused per-clip kernels (k_vae_fused.hip, k_den_fused.hip), for bench.py's attention-only roofline figure:
 * mask 1 = launch the instantiation WITHOUT softmax(Q K^T) V (projections, K / V images, out_proj and every barrier stay) - outputs
 * are wrong by construction; full - ablated launch time = the S ~ 300 self-attention's time.  mask 0 restores the product kernel. */
int amuse_debug_set_ablation(amuse_ctx* ctx, int mask);

/* The host packer's fp32 -> (hi, lo) fp16 split of AMUSE_PREC_F32X, for tests (host memory, no GPU call):
 * hi[i] = rn16(w[i]), lo[i] = rn16(w[i] - hi[i]), round-to-nearest-even with gradual underflow - bit for bit what
 * v_cvt_pk_f16_f32 produces on the device for the activations (and amuse_update_weights_device for the weights). */
int amuse_debug_f16_split(const float* w, size_t n, uint16_t* hi, uint16_t* lo);

/* ------------------------------------------------------------------ training-step glue (csrc/k_train.hip; amuse_amd/train_ops.py)
 * The row-wise arithmetic of a transformer layer's forward and backward pass in the train_gesture iteration (reference
 * scripts/trainer.py:335-498; layers utils/cross_attention.py:259-272,323-345; the reference gets these from torch's eager kernels:
 * nn.Dropout, the residual add, nn.LayerNorm, F.gelu and the bias gradients' reductions).  fp32, row-major [rows][C]; no context - plain
 * device pointers and a stream.  Dropout masks are counter-based: element e is draw e % 4 of Philox4x32-10(key seed, counter (e / 4, offset));
 * the backward calls regenerate the mask from the same (p, seed, offset).  p = 0 is eval mode.
 * `ws` (amuse_train_ws_floats() floats) holds the partial column sums of a call (deterministic: added up in a fixed order by a second launch) and
 * must not be shared by calls that may overlap on different streams. */
size_t amuse_train_ws_floats(void);
/* The library's OWN scratch (split-k partials of the weight gradients and of the generic GEMM) exists once per device and LANE (0 or 1): calls that may overlap on two
 * streams of a device - the trainer issues the Denoiser's forward / backward pass beside the prior's - run on different lanes.  Sets the lane of the CALLING THREAD for
 * its following amuse_train_* calls (initially 0). */
int amuse_train_set_lane(int lane);
/* out = LayerNorm_128(x + dropout(y + bias)) (x, bias nullable); zhat [rows][128] = the normalised rows and rstd [rows] for the backward
 * pass (both nullable) */
int amuse_train_ln_fwd(const float* x, const float* y, const float* bias, const float* gamma, const float* beta, float p, uint64_t seed,
                       uint64_t offset, long rows, float* out, float* zhat, float* rstd, void* stream);
/* dz = LayerNorm backward of dout + dout2 (dout2 nullable: the second branch's gradient of a residual stream); dx = dz (nullable),
 * dy = dz . mask / (1 - p); dgamma, dbeta, dbias = sum_rows dy (each nullable) */
int amuse_train_ln_bwd(const float* dout, const float* dout2, const float* zhat, const float* rstd, const float* gamma, float p, uint64_t seed, uint64_t offset,
                       long rows, float* dx, float* dy, float* dgamma, float* dbeta, float* dbias, float* ws, void* stream);
/* out = dropout(gelu(h + b)), exact erf; h, out [rows][F], F a multiple of 4 up to 1024 */
int amuse_train_bias_gelu_drop_fwd(const float* h, const float* b, float p, uint64_t seed, uint64_t offset, long rows, int F, float* out,
                                   void* stream);
/* dh = da . mask / (1 - p) . gelu'(h + b); db [F] = sum_rows dh */
int amuse_train_bias_gelu_drop_bwd(const float* da, const float* h, const float* b, float p, uint64_t seed, uint64_t offset, long rows,
                                   int F, float* dh, float* db, float* ws, void* stream);
/* out [C] = sum_rows x[r][:] (a bias gradient), C a multiple of 4 up to 1024 */
int amuse_train_colsum(const float* x, long rows, int C, float* out, float* ws, void* stream);

/* Self-attention core of a layer (nn.MultiheadAttention between its in- and out-projection, 4 heads of 32, S <= 304 tokens) in fp32 on the matrix cores
 * (csrc/k_train_attn.hip): o [B S][128] = dropout(softmax(q k^T / sqrt(32))) v per (clip, head) from the packed projections qkv [B S][384] (q | k | v,
 * heads contiguous 32-column slices), lse [B][4][S] for the backward pass; the backward call returns d(qkv).  The dropout mask is a hash of
 * (seed, offset, clip, head, query, key): the backward call regenerates it from the same (p, seed, offset).  mask_debug (nullable): [B][4][S][S] keep / (1 - p). */
int amuse_train_attn_fwd(const float* qkv, int B, int S, float p, uint64_t seed, uint64_t offset, float* o, float* lse, float* mask_debug, void* stream);
int amuse_train_attn_bwd(const float* qkv, const float* o, const float* lse, const float* dout, int B, int S, float p, uint64_t seed, uint64_t offset,
                         float* dqkv, void* stream);
/* torch.optim.AdamW (amsgrad off) over a contiguous range of flat fp32 buffers, `step` = this update's 1-based count (trainer.py:181-184: the reference's
 * optimizer over prior + ldm parameters; amuse_amd/train_gesture.py keeps parameters, gradients and both moments in one buffer each) */
int amuse_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1, double beta2, double eps,
                      double weight_decay, long step, void* stream);
/* The same update with the step count on the DEVICE - for a training step captured as a HIP graph, whose host-side count is frozen at capture:
 * advance != 0 first adds 1 to *step_dev and leaves the two bias-correction scalars (computed in double, as the host path does) in scal_dev[0..1]; the update
 * then reads them.  One advancing call per optimizer step, advance = 0 for the step's further ranges.  step_dev, scal_dev: device memory (8 + 8 bytes). */
int amuse_train_adamw_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1, double beta2, double eps,
                          double weight_decay, long* step_dev, float* scal_dev, int advance, void* stream);
/* The dropout epoch: one device word per GPU that every mask-drawing kernel of the training step (LayerNorm / FFN / value-path dropouts, the attention's hash)
 * mixes into its counter.  It is 0 unless advanced - eager training hands every call a fresh (seed, offset) from the host.  A step captured as a HIP graph
 * replays with the offsets it was captured with, so the graph ends with amuse_train_epoch_advance(1): every replay then draws fresh masks, and the forward and
 * backward halves of one step still see the same value.  amuse_train_epoch_set pins it (tests, resuming). */
int amuse_train_epoch_advance(unsigned add, void* stream);
int amuse_train_epoch_set(unsigned value, void* stream);
/* Layer-level entry points: a whole TransformerEncoderLayer / TransformerDecoderLayer (forward_post, memory of one token) but its self-attention
 * core, forward and backward, in one call each - the kernels above plus the layer's plain GEMMs on the library's own fp32-MFMA kernels
 * (csrc/k_train_gemm.hip; no vendor BLAS).  The caller computes q | k | v = amuse_train_linear_fwd(x, in_proj), runs its attention on them (o2 = the heads' outputs
 * concatenated), calls amuse_train_layer_fwd; on the way back amuse_train_layer_bwd returns d(o2) for the attention's backward pass, whose d(q | k | v)
 * goes through amuse_train_linear_bwd(..., dx = L.dx, accumulate_dx = 1).  Linear weights in PyTorch's [out][in] layout, everything fp32 and dense. */
typedef struct amuse_train_layer {
    long rows; int B, S, H, ff;                 /* rows = B x S tokens of 128 features; H heads; ff = linear1's width (multiple of 4, <= 1024) */
    float p, p_attn;                            /* dropout probabilities: the layer's nn.Dropout modules / the attention's (0 = eval mode) */
    uint64_t seed, off[5];                      /* mask offsets: [0] behind the self-attention, [1] behind the cross-attention, [2] inside the FFN,
                                                   [3] behind the FFN, [4] the cross-attention's probabilities */
    const float *Wo, *bo, *g1, *be1;            /* self_attn.out_proj, norm1 */
    const float *Wv, *bv, *Wc, *bc, *g2, *be2;  /* decoder layer: multihead_attn's value rows of in_proj, its out_proj, norm2 (NULL for an encoder layer) */
    const float *W1, *b1, *W2, *b2, *g3, *be3;  /* linear1, linear2, the last norm (an encoder layer's norm2) */
    const float *x, *o2, *mem;                  /* in: residual stream [rows][128], attention output [rows][128], memory [B][128] (NULL: encoder layer) */
    float *x1, *zh1, *r1;                       /* kept for the backward pass: behind norm1 ([rows][128], normalised rows, [rows]) */
    float *c, *vk, *xm, *zh2, *r2;              /* decoder: value projection [B][128], its masked copy [rows][128], behind norm2 */
    float *h, *a;                               /* [rows][ff]: linear1's output, the FFN activation */
    float *out, *zh3, *r3;                      /* the layer's output and the last norm's state */
    float *tmp;                                 /* [rows][128] scratch */
    const float* dout;                          /* backward in: d(out) */
    float *dx, *do2, *dmem;                     /* backward out: d(x) WITHOUT the attention's share, d(o2), d(mem) [B][128] */
    float *dWo, *dbo, *dg1, *dbe1, *dWv, *dbv, *dWc, *dbc, *dg2, *dbe2, *dW1, *db1, *dW2, *db2, *dg3, *dbe3;
    float *s128a, *s128b, *s512a, *s512b, *sdc; /* backward scratch: 2 x [rows][128], 2 x [rows][ff], [B][128] */
    float* ws;                                  /* amuse_train_ws_floats() floats */
    /* optional: the layer's self-attention inside the same calls (Win != NULL; 4 heads, S <= 304).  Forward: qkv = x Win^T + bin, o2 = attention(qkv)
     * (o2 then is an OUTPUT buffer), lse kept; backward: d(qkv) from d(o2), dWin / dbin, and dx receives the attention's share too. */
    const float *Win, *bin;                     /* self_attn.in_proj_weight [384][128], in_proj_bias [384] */
    float *qkv, *lse;                           /* [rows][384], [B][4][S]: kept for the backward pass */
    float *dqkv, *dWin, *dbin;                  /* backward: [rows][384] scratch, the in-projection's gradients */
    uint64_t off_self;                          /* mask offset of the self-attention's dropout */
} amuse_train_layer;
int amuse_train_layer_fwd(const amuse_train_layer* layer, void* stream);
/* (amuse_train_layer_bwd of a decoder layer with >= 1,024 rows issues the memory token's branch - d(c), dWv, dbv, d(mem) - on a stream of the library's own, forked from
 * and joined back into `stream` inside the call: everything is complete in `stream` order when the call returns, also under stream capture) */
int amuse_train_layer_bwd(const amuse_train_layer* layer, void* stream);
/* out [rows][N] = x [rows][K] W^T + b (b nullable);  dW [N][K] = dy^T x, db [N] = sum_rows dy, dx [rows][K] (+)= dy W (each output nullable) */
int amuse_train_linear_fwd(const float* x, const float* W, const float* b, long rows, int K, int N, float* out, void* stream);
int amuse_train_linear_bwd(const float* dy, const float* x, const float* W, long rows, int K, int N, float* dW, float* db, float* dx,
                           int accumulate_dx, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMUSE_HIP_H */
