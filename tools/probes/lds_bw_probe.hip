// LDS read-bandwidth probe on gfx950: W waves per workgroup, each issuing batches of N ds_read_b128 (lane-linear, conflict-free)
// per lgkmcnt(0) wait.  Prints cycles per wave-instruction and B/clk/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <int N>
__global__ __launch_bounds__(512) void k(unsigned long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"((j % 8) * 1024) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < N; ++j) acc ^= v[j];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345 && acc[1] == 7) out[1000] = acc[2];
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 8192 * 8);
  const int iters = 2000;
  for (int waves = 1; waves <= 8; waves *= 2) {
    k<12><<<256, waves * 64>>>(d, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    const double cyc = (double)h / iters / 12;
    printf("waves/CU %d, 12 reads per wait: %.1f cycles per wave-instruction, %.0f B/clk/CU\n", waves, cyc, waves * 1024.0 / cyc);
  }
  k<4><<<256, 256>>>(d, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("waves/CU 4, 4 reads per wait: %.1f cycles per wave-instruction\n", (double)h / iters / 4);
  return 0;
}
