// Stand-ins for the HIP runtime and for the kernel launchers, so that the library's HOST code (context construction, weight
// packing into MFMA-fragment streams, workspace management, argument checking: amuse_api.hip, amuse_audio_api.hip) can run under
// AddressSanitizer / UBSan on a machine without a GPU.  "Device" memory is host memory; launches are no-ops.  Test
// infrastructure only (tests/test_host_asan.py builds it) - never linked into libamuse_hip.so.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../amuse_amd/csrc/amuse_audio.hpp"
#include "../../amuse_amd/csrc/amuse_kernels.hpp"

static long g_live = 0;
long amuse_stub_live_allocations() { return g_live; }

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); ++g_live; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (p) { free(p); --g_live; } return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); ++g_live; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); --g_live; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); ++g_live; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); --g_live; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
}

namespace amuse {
hipError_t launch_sample(const SampleArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t launch_sample8(const SampleArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_sample8x(const SampleArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_sample8h(const SampleArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_time_tokens(const int*, int, const float*, const float*, const float*, const float*, const float*, const float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_cond_tokens(const CondArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_repack(const float*, const int*, void*, size_t, int, hipStream_t) { return hipSuccess; }
hipError_t launch_add_noise(const float*, const float*, const float*, const float*, float*, int, hipStream_t, int) { return hipSuccess; }
hipError_t launch_counter_normal(uint64_t, uint64_t, int, int, int, float*, hipStream_t, int) { return hipSuccess; }
hipError_t launch_vae_rows(const VaeRowsArgs&, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_rows8x(const VaeRowsArgs&, hipStream_t, int) { return hipSuccess; }
hipError_t launch_vae_attn(const VaeAttnArgs&, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_fused(const VaeFusedArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_fusedh(const VaeFusedArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_den_fused(const DenFusedArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_den_fusedh(const DenFusedArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_fusedx(const VaeFusedXArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_den_fusedx(const DenFusedXArgs&, hipStream_t) { return hipSuccess; }
hipError_t launch_sample_dec(const SampleDecArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t launch_mem_kv(const float*, int, const float*, const float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_feats_to_smplx(const float*, size_t, int, float*, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_smplx_to_feats(const float*, const float*, size_t, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_latent(const float*, const float*, float*, float*, float*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_vae_ca(const float*, const float*, const float*, const float*, const float*, float*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_gemm(const GemmArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t launch_fbank(const float*, int, int, const float*, const float*, const int*, float, float, float*, hipStream_t) { return hipSuccess; }
hipError_t launch_im2col(const float*, unsigned short*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_ast_tokens(const float*, const float*, const float*, float*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_ln_bf16(const float*, const float*, const float*, float, unsigned short*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_tile_bf16(const unsigned short*, unsigned short*, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_untile_bf16(const unsigned short*, unsigned short*, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_untile_f32(const float*, float*, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t launch_ast_attn(const unsigned short*, const unsigned short*, unsigned short*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_ast_pool(const float*, const float*, const float*, int, float*, int, hipStream_t) { return hipSuccess; }
hipError_t launch_ast_head(const float*, int, const float*, const float*, const unsigned short*, const float*, float*, int, hipStream_t) { return hipSuccess; }
}  // namespace amuse
