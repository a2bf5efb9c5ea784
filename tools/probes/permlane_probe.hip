#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned l = threadIdx.x;
  auto a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
  out[l] = a[0]; out[64 + l] = a[1];
  auto b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
  out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* n[4] = {"p16.vdst", "p16.src ", "p32.vdst", "p32.src "};
  for (int r = 0; r < 4; r++) { printf("%s:", n[r]); for (int i = 0; i < 64; i += 1) printf(" %u", h[r * 64 + i]); printf("\n"); }
}
