// Probe of ds_read_b64_tr_b16 lane/element mapping on gfx950: lane l supplies the address of shorts 4l..4l+3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  auto p = (__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 256 * 2);
  k<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (L%2d,e%d)", h[l*4+j] / 4, h[l*4+j] % 4); printf("\n"); }
  return 0;
}
