// LDS-DMA (buffer_load_dwordx4 ... lds) throughput per CU by piece shape and number of issuing waves, L2-resident source laid
// out like a GEMM operand (row stride LD bytes).  hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// SHAPE: bytes per row segment fetched by consecutive lanes (64, 128, 256, 1024).  WAVES: issuing waves per workgroup (of 8).
// MODE 0: LDS-DMA; MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128.  INFL: pieces kept in flight per wave.
template <int SHAPE, int MODE, int INFL>
__global__ __launch_bounds__(512) void probe(const unsigned char* src, int ld, int iters, int waves, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= waves) return;
  constexpr int LPR = SHAPE / 16;     // lanes per row segment
  constexpr int ROWS = 64 / LPR;      // rows per piece
  const int prow = lane / LPR, pch = lane % LPR;
  // workgroup b streams a 256-row panel; wave w takes rows [w*32, w*32+32) of it, walking K (columns) in steps of SHAPE bytes
  const unsigned char* base = src + (size_t)(blockIdx.x % 8) * 256 * ld + (size_t)wave * 32 * ld;  // one 2 MB panel per XCD: L2 hits
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)(32 * ld), 0x00020000);
  unsigned voff = prow * ld + pch * 16;
  unsigned acc = 0;
  const int kwrap = ld / SHAPE;
  for (int it = 0; it < iters; ++it) {
    const int k = it % kwrap;
    const int rblk = (it / kwrap) % (32 / ROWS > 0 ? 32 / ROWS : 1);
    const unsigned soff = (unsigned)(k * SHAPE + rblk * ROWS * ld);
    unsigned char* dst = smem + wave * 16384 + (it % 16) * 1024;
    if (MODE == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)dst, 16, voff, soff, 0, 0);
      if (INFL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (INFL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (INFL == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
      *(u32x4*)(dst + lane * 16) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc += smem[threadIdx.x * 4];
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int SHAPE, int MODE, int INFL>
void run(const char* name, const unsigned char* src, int ld, int waves, unsigned* sink) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<SHAPE, MODE, INFL><<<256, 512>>>(src, ld, 64, waves, sink);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    probe<SHAPE, MODE, INFL><<<256, 512>>>(src, ld, iters, waves, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = 256.0 * waves * iters * 1024.0;
  printf("%-28s waves=%d infl=%2d: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip  %.0f ns per piece per wave\n", name, waves, INFL, best * 1e3,
         bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12, best * 1e6 / iters);
}

int main() {
  const int ld = 8192;  // bytes per operand row
  unsigned char* src;
  unsigned* sink;
  hipMalloc(&src, (size_t)8 * 256 * ld + 4096);
  hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)8 * 256 * ld);
  for (int waves : {4, 8}) {
    run<64, 0, 8>("dma 16 rows x 64 B", src, ld, waves, sink);
    run<128, 0, 8>("dma 8 rows x 128 B", src, ld, waves, sink);
    run<256, 0, 8>("dma 4 rows x 256 B", src, ld, waves, sink);
    run<1024, 0, 8>("dma 1 row x 1024 B", src, ld, waves, sink);
    run<128, 0, 4>("dma 8 rows x 128 B", src, ld, waves, sink);
    run<128, 0, 16>("dma 8 rows x 128 B", src, ld, waves, sink);
    run<64, 0, 16>("dma 16 rows x 64 B", src, ld, waves, sink);
    run<64, 1, 8>("reg 16 rows x 64 B", src, ld, waves, sink);
    run<128, 1, 8>("reg 8 rows x 128 B", src, ld, waves, sink);
    run<1024, 1, 8>("reg 1 row x 1024 B", src, ld, waves, sink);
  }
  return 0;
}
