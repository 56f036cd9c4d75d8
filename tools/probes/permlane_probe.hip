// Probe of v_permlane32_swap via the builtin on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned x = threadIdx.x + 100;
  auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  unsigned h[128]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8) printf("lane %2d: r0=%u r1=%u\n", l, h[2*l], h[2*l+1]);
  return 0;
}
