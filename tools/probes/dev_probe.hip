#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../amuse_amd/csrc/amuse_dev.hpp"
using namespace amuse;
__global__ void k(float* out, const float* in) {
  int l = threadIdx.x & 63;
  float v = in[l];
  out[l] = allreduce_g_sum(v);
  out[64 + l] = allreduce_g_max(v);
  out[128 + l] = v + __shfl_xor(v, 16) ;
}
__global__ void k2(float* out, const float* in) {   // exchange_sum test: 256 threads
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* exch = reinterpret_cast<f32x4*>(smem);
  int lane = threadIdx.x & 63; int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 part[kTiles];
  for (int t = 0; t < kTiles; ++t) for (int m = 0; m < 4; ++m) part[t][m] = in[((wave * kTiles + t) * 64 + lane) * 4 + m];
  int parity = 0;
  exchange_sum(part, exch, parity, wave, lane);
  for (int t = 0; t < kTiles; ++t) for (int m = 0; m < 4; ++m) out[((wave * kTiles + t) * 64 + lane) * 4 + m] = part[t][m];
}
int main() {
  float h[64], *d, *o; for (int i = 0; i < 64; i++) h[i] = sinf(i * 1.3f) * 3;
  hipMalloc(&d, 256); hipMalloc(&o, 1024); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(o, d); float r[256]; hipMemcpy(r, o, 768, hipMemcpyDeviceToHost);
  double me = 0, mm = 0;
  for (int l = 0; l < 64; l++) { int c = l & 15; float s = ((h[c] + h[c + 16]) + (h[c + 32] + h[c + 48])); float mx = fmaxf(fmaxf(h[c], h[c + 16]), fmaxf(h[c + 32], h[c + 48]));
    me = fmax(me, fabs(r[l] - s)); mm = fmax(mm, fabs(r[64 + l] - mx)); }
  printf("allreduce sum err %g max err %g  (r[0]=%g expect %g)\n", me, mm, r[0], h[0] + h[16] + h[32] + h[48]);
  const int N = 4 * kTiles * 64 * 4; float* hi = new float[N]; for (int i = 0; i < N; i++) hi[i] = cosf(i * 0.37f);
  float *di, *dо; hipMalloc(&di, N * 4); hipMalloc(&dо, N * 4); hipMemcpy(di, hi, N * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, kExchBytes);
  k2<<<1, 256, kExchBytes>>>(dо, di); float* ho = new float[N]; hipMemcpy(ho, dо, N * 4, hipMemcpyDeviceToHost);
  double ee = 0; for (int w = 0; w < 4; w++) for (int i = 0; i < kTiles * 64 * 4; i++) { float s = ((hi[i] + hi[kTiles * 256 + i]) + hi[2 * kTiles * 256 + i]) + hi[3 * kTiles * 256 + i]; ee = fmax(ee, fabs(ho[w * kTiles * 256 + i] - s)); }
  printf("exchange err %g\n", ee);
}
