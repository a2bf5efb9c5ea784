/* A plain C99 client of include/amuse_hip.h - no torch, no Python, no HIP headers: what a reference-side binding in any language
 * with a C FFI sees of libamuse_hip.so.  Test infrastructure (tests/test_c_client.py builds and runs it).
 *
 *   client abi                      version + error convention, touches no GPU
 *   client plan CLIPS PREC TOKENS   amuse_plan of a job (the library's kernel-choice rule; no GPU): "PLAN g decode encode step"
 *   client run DIR B SEED           DIR/den.f32, prior.f32 (host fp32 parameter images, state-dict order), sched.bin
 *                                   (int32 T, int32 timesteps[T], float coef[T][8], float freqs[128]), cond.f32 ([3][B][256]) ->
 *                                   amuse_create, amuse_set_schedule, the job's amuse_plan pinned as a sharding caller does,
 *                                   amuse_diffusion_backward (bf16, counter-based noise) ->
 *                                   DIR/out_latents.f32 [B][128], out_poses.f32 [B][300][55][3], out_trans.f32 [B][300][3]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amuse_hip.h"

/* the four runtime entry points a caller needs to own device buffers (libamdhip64; enum values of hipMemcpyKind) */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind);
extern int hipDeviceSynchronize(void);
enum { H2D = 1, D2H = 2 };

static void* slurp(const char* dir, const char* name, size_t bytes) {
    char path[1024];
    void* buf = malloc(bytes);
    FILE* f;
    snprintf(path, sizeof path, "%s/%s", dir, name);
    f = fopen(path, "rb");
    if (!f || !buf || fread(buf, 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot read %lu bytes of %s\n", (unsigned long)bytes, path);
        exit(2);
    }
    fclose(f);
    return buf;
}

static void dump(const char* dir, const char* name, const void* dev, size_t bytes) {
    char path[1024];
    void* buf = malloc(bytes);
    FILE* f;
    if (!buf || hipMemcpy(buf, dev, bytes, D2H) != 0) {
        fprintf(stderr, "copy back of %s failed\n", name);
        exit(2);
    }
    snprintf(path, sizeof path, "%s/%s", dir, name);
    f = fopen(path, "wb");
    if (!f || fwrite(buf, 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot write %s\n", path);
        exit(2);
    }
    fclose(f);
    free(buf);
}

static float* to_device(const float* host, size_t n) {
    void* d = NULL;
    if (hipMalloc(&d, n * sizeof(float)) != 0 || hipMemcpy(d, host, n * sizeof(float), H2D) != 0) {
        fprintf(stderr, "device upload failed\n");
        exit(2);
    }
    return (float*)d;
}

static int check(int rc, const char* what) {
    if (rc != AMUSE_OK) {
        fprintf(stderr, "%s: %d (%s)\n", what, rc, amuse_last_error());
        exit(1);
    }
    return rc;
}

int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "abi") == 0) {
        float one = 1.0f;
        if (amuse_abi_version() != AMUSE_ABI_VERSION) return 1;
        if (amuse_create(0, &one, 1, &one, 1) != NULL) return 1;            /* wrong sizes: refused before any GPU call */
        if (strlen(amuse_last_error()) == 0) return 1;
        if (amuse_set_schedule(NULL, NULL, NULL) != AMUSE_EINVAL) return 1;
        if (amuse_sample(NULL, NULL, NULL, NULL, 1, AMUSE_PREC_BF16, 0, 0, NULL, NULL, NULL, NULL, NULL) != AMUSE_EINVAL) return 1;
        amuse_destroy(NULL);
        printf("ABI_OK %d\n", amuse_abi_version());
        return 0;
    }
    if (argc == 5 && strcmp(argv[1], "plan") == 0) {
        int g = -1, dp = -1, ep = -1, sp = -1;
        if (amuse_plan(AMUSE_ARCH_ENC, atoi(argv[3]), atoi(argv[2]), atoi(argv[4]), &g, &dp, &ep, &sp) != 0) {
            fprintf(stderr, "amuse_plan: %s\n", amuse_last_error());
            return 1;
        }
        printf("PLAN %d %d %d %d\n", g, dp, ep, sp);
        return 0;
    }
    if (argc == 5 && strcmp(argv[1], "run") == 0) {
        const char* dir = argv[2];
        const int B = atoi(argv[3]);
        const uint64_t seed = (uint64_t)strtoull(argv[4], NULL, 10);
        float* den = (float*)slurp(dir, "den.f32", (size_t)AMUSE_DENOISER_PARAMS * sizeof(float));
        float* prior = (float*)slurp(dir, "prior.f32", (size_t)AMUSE_PRIOR_PARAMS * sizeof(float));
        int* head = (int*)slurp(dir, "sched.bin", sizeof(int));
        const int T = head[0];
        char* sched_raw;
        float* cond;
        float *d_cond, *d_lat, *d_poses, *d_trans;
        int g = 0, dpath = 0;
        amuse_schedule s;
        amuse_ctx* ctx;
        if (B < 1 || T < 1 || T > AMUSE_MAX_STEPS) return 2;
        sched_raw = (char*)slurp(dir, "sched.bin", sizeof(int) * (size_t)(1 + T) + sizeof(float) * (8u * (size_t)T + 128u));
        cond = (float*)slurp(dir, "cond.f32", sizeof(float) * 3u * (size_t)B * AMUSE_COND_DIM);
        s.n_steps = T;
        s.timesteps = (const int*)(sched_raw + sizeof(int));
        s.coef = (const float*)(sched_raw + sizeof(int) * (size_t)(1 + T));
        s.freqs = s.coef + 8u * (size_t)T;
        ctx = amuse_create(0, den, AMUSE_DENOISER_PARAMS, prior, AMUSE_PRIOR_PARAMS);
        if (!ctx) {
            fprintf(stderr, "amuse_create: %s\n", amuse_last_error());
            return 1;
        }
        check(amuse_set_schedule(ctx, &s, NULL), "amuse_set_schedule");
        /* the job's launch plan, pinned the way a caller that shards the job over GPUs pins it on every shard (here: one shard = the job) */
        check(amuse_plan(AMUSE_ARCH_ENC, AMUSE_PREC_BF16, B, 5, &g, &dpath, NULL, NULL), "amuse_plan");
        check(amuse_set_clips_per_group(ctx, g), "amuse_set_clips_per_group");
        check(amuse_set_decode_path(ctx, dpath), "amuse_set_decode_path");
        d_cond = to_device(cond, 3u * (size_t)B * AMUSE_COND_DIM);
        if (hipMalloc((void**)&d_lat, sizeof(float) * (size_t)B * AMUSE_D_MODEL) != 0 ||
            hipMalloc((void**)&d_poses, sizeof(float) * (size_t)B * AMUSE_N_FRAMES * AMUSE_N_JOINTS * 3u) != 0 ||
            hipMalloc((void**)&d_trans, sizeof(float) * (size_t)B * AMUSE_N_FRAMES * 3u) != 0)
            return 2;
        check(amuse_diffusion_backward(ctx, d_cond, d_cond + (size_t)B * AMUSE_COND_DIM, d_cond + 2u * (size_t)B * AMUSE_COND_DIM, B,
                                       AMUSE_PREC_BF16, AMUSE_QUAT_P3D, seed, 0, NULL, NULL, d_lat, d_poses, d_trans, NULL),
              "amuse_diffusion_backward");
        if (hipDeviceSynchronize() != 0) return 2;
        dump(dir, "out_latents.f32", d_lat, sizeof(float) * (size_t)B * AMUSE_D_MODEL);
        dump(dir, "out_poses.f32", d_poses, sizeof(float) * (size_t)B * AMUSE_N_FRAMES * AMUSE_N_JOINTS * 3u);
        dump(dir, "out_trans.f32", d_trans, sizeof(float) * (size_t)B * AMUSE_N_FRAMES * 3u);
        amuse_destroy(ctx);
        hipFree(d_cond);
        hipFree(d_lat);
        hipFree(d_poses);
        hipFree(d_trans);
        free(den);
        free(prior);
        free(head);
        free(sched_raw);
        free(cond);
        printf("RUN_OK B=%d T=%d\n", B, T);
        return 0;
    }
    fprintf(stderr, "usage: client abi | client plan CLIPS PREC TOKENS | client run DIR B SEED\n");
    return 2;
}
