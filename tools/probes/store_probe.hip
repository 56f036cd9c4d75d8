// Epilogue store patterns of a 256 x 256 fp32 tile per workgroup (8 waves, each 64 rows x 128 columns as 32 x 64 blocks), whole chip:
//   A  lane (row r = lane>>3, chunk c = lane&7) stores float4 at columns 8c and 8c+4  (two half-filled 128-byte lines per row and instruction)
//   B  same lanes, float4 at columns 4c and 32+4c                                     (one full 128-byte line per row and instruction)
//   C  lane (row r = lane>>4, piece q = lane&15) stores float4 at column 4q            (one full 256-byte row run per instruction)
// hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int PAT>
__global__ __launch_bounds__(512) void probe(float* out, int ld, int tiles_n, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  for (int rep = 0; rep < reps; ++rep) {
    const int tile = blockIdx.x + rep * gridDim.x;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    float* base = out + (size_t)(tm * 256 + wm * 64) * ld + tn * 256 + wn * 128;
    const float4 v = make_float4(lane, wave, rep, 1.f);
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* blk = base + (size_t)(i * 32) * ld + jp * 64;
        if (PAT == 0 || PAT == 1) {
          const int prow = lane >> 3, c = lane & 7;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float* rp = blk + (size_t)(k * 8 + prow) * ld;
            if (PAT == 0) {
              *(float4*)(rp + 8 * c) = v;
              *(float4*)(rp + 8 * c + 4) = v;
            } else {
              *(float4*)(rp + 4 * c) = v;
              *(float4*)(rp + 32 + 4 * c) = v;
            }
          }
        } else {
          const int prow = lane >> 4, q = lane & 15;
#pragma unroll
          for (int k = 0; k < 8; ++k) *(float4*)(blk + (size_t)(k * 4 + prow) * ld + 4 * q) = v;
        }
      }
  }
}

template <int PAT>
void run(const char* name, float* out, int M, int N) {
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n, reps = tiles / 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<PAT><<<256, 512>>>(out, N, tiles_n, reps);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    probe<PAT><<<256, 512>>>(out, N, tiles_n, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-40s %dx%d: %8.1f us  %.2f TB/s\n", name, M, N, best * 1e3, (double)M * N * 4 / (best * 1e-3) / 1e12);
}

int main() {
  float* out;
  const int M = 16384, N = 4096;
  hipMalloc(&out, (size_t)M * N * 4);
  run<0>("A: float4 at 8c, 8c+4 (half lines)", out, M, N);
  run<1>("B: float4 at 4c, 32+4c (full lines)", out, M, N);
  run<2>("C: 16 lanes per row (256-byte runs)", out, M, N);
  run<0>("A again", out, M, N);
  return 0;
}
