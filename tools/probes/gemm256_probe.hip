// Candidate GEMM for the audio front-end, in isolation: 256 x 256 x 64 macro tile, 8 waves (2 feature halves x 4 row
// quarters, 128 x 64 per wave = 128 accumulator registers), both operands through double-buffered padded LDS, one
// barrier per k-tile.  C[row][feature] = sum_k X[row][k] W[feature][k]  (bf16 in, fp32 out).
// Question it answers: does the larger tile lift the 128 x 128 kernel's load-path / LDS-read limits?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDSK = 72;                       // padded row: 64 + 8 halfwords
constexpr int TILE = 256, KT = 64;
constexpr int BUF = 2 * TILE * LDSK * 2;       // bytes per buffer (W tile + X tile)
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int STORE>
__global__ __launch_bounds__(512) void k_gemm256(const unsigned short* __restrict__ X, const unsigned short* __restrict__ W,
                                                 float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, j = lane & 15;
  const int wf = wave >> 2, wr = wave & 3;     // feature half, row quarter
  const int tn = blockIdx.x % (N / TILE), tm = blockIdx.x / (N / TILE);
  const int lrow = t >> 1, lhalf = t & 1;      // loader mapping: 2 threads per 128-byte row
  const uint4* gW = reinterpret_cast<const uint4*>(W + (size_t)(tn * TILE + lrow) * K + 32 * lhalf);
  const uint4* gX = reinterpret_cast<const uint4*>(X + (size_t)(tm * TILE + lrow) * K + 32 * lhalf);
  auto lds_w = [&](int b) { return reinterpret_cast<unsigned short*>(smem + b * BUF); };
  auto lds_x = [&](int b) { return reinterpret_cast<unsigned short*>(smem + b * BUF + TILE * LDSK * 2); };
  uint4 rw[4], rx[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { rw[c] = gW[c]; rx[c] = gX[c]; }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    *reinterpret_cast<uint4*>(lds_w(0) + lrow * LDSK + 32 * lhalf + 8 * c) = rw[c];
    *reinterpret_cast<uint4*>(lds_x(0) + lrow * LDSK + 32 * lhalf + 8 * c) = rx[c];
  }
  __syncthreads();
  f32x4 acc[8][4];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / KT;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    // unconditional (arrays written under a condition go to scratch): the last iteration re-fetches its own tile
    const int kn = (kt + 1 < nk) ? kt + 1 : kt;
#pragma unroll
    for (int c = 0; c < 4; ++c) { rw[c] = gW[kn * 8 + c]; rx[c] = gX[kn * 8 + c]; }
    const unsigned short* Wb = lds_w(b) + (size_t)(128 * wf + j) * LDSK + 8 * g;
    const unsigned short* Xb = lds_x(b) + (size_t)(64 * wr + j) * LDSK + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 wfr[8], xfr[4];
#pragma unroll
      for (int x = 0; x < 8; ++x) wfr[x] = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(16 * x) * LDSK + 32 * s);
#pragma unroll
      for (int y = 0; y < 4; ++y) xfr[y] = *reinterpret_cast<const bf16x8*>(Xb + (size_t)(16 * y) * LDSK + 32 * s);
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[x], xfr[y], acc[x][y], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<uint4*>(lds_w(b ^ 1) + lrow * LDSK + 32 * lhalf + 8 * c) = rw[c];
      *reinterpret_cast<uint4*>(lds_x(b ^ 1) + lrow * LDSK + 32 * lhalf + 8 * c) = rx[c];
    }
    __syncthreads();
  }
  if (STORE) {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int row = tm * TILE + 64 * wr + 16 * y + j, f = tn * TILE + 128 * wf + 16 * x + 4 * g;
        *reinterpret_cast<f32x4*>(C + (size_t)row * N + f) = acc[x][y];
      }
  } else {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][3];
    if (s == 12345.678f) C[0] = s;
  }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = 38912;   // 32 clips x 1216 rows
  const int shapes[4][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm256<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm256<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
  for (auto& sh : shapes) {
    const int N = sh[0], K = sh[1];
    std::vector<unsigned short> hx((size_t)M * K), hw((size_t)N * K);
    unsigned s = 1234567u;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.1f);
    unsigned short *dx, *dw; float* dc;
    CHECK(hipMalloc(&dx, hx.size() * 2)); CHECK(hipMalloc(&dw, hw.size() * 2)); CHECK(hipMalloc(&dc, (size_t)M * N * 4));
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const int grid = (M / TILE) * (N / TILE);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int store = 1; store >= 0; --store) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) {
          if (store) hipLaunchKernelGGL(k_gemm256<1>, dim3(grid), dim3(512), 2 * BUF, 0, dx, dw, dc, M, N, K);
          else hipLaunchKernelGGL(k_gemm256<0>, dim3(grid), dim3(512), 2 * BUF, 0, dx, dw, dc, M, N, K);
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("N=%d K=%d store=%d: %.1f us  %.0f TFLOP/s\n", N, K, store, ms / 5 * 1e3, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12);
      }
    }
    // spot check
    std::vector<float> hc(16);
    double maxerr = 0;
    for (int q = 0; q < 8; ++q) {
      const int row = (q * 4801 + 17) % M, f = (q * 977 + 5) % N;
      float got; CHECK(hipMemcpy(&got, dc + (size_t)row * N + f, 4, hipMemcpyDeviceToHost));
      double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)row * K + k]) * bf2f(hw[(size_t)f * K + k]);
      maxerr = fmax(maxerr, fabs(ref - got));
    }
    printf("   spot-check max err %.3e\n", maxerr);
    CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(dc));
  }
  return 0;
}
