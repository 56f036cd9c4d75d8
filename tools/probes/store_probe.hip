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
        if (PAT == 3 || PAT == 4) {
          const int prow = lane >> 3, c = lane & 7;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float* rp = blk + (size_t)(k * 8 + prow) * ld;
            if (PAT == 3) {
              typedef __attribute__((ext_vector_type(4))) float f32x4_t;
              const f32x4_t vv = {v.x, v.y, v.z, v.w};
              __builtin_nontemporal_store(vv, (f32x4_t*)(rp + 4 * c));
              __builtin_nontemporal_store(vv, (f32x4_t*)(rp + 32 + 4 * c));
            } else {
              typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
              __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)rp, (short)0, 1 << 20, 0x00020000);
              const u32x4_t u = __builtin_bit_cast(u32x4_t, v);
              __builtin_amdgcn_raw_buffer_store_b128(u, rs_, 16 * c, 0, 16 /* sc1 */);
              __builtin_amdgcn_raw_buffer_store_b128(u, rs_, 128 + 16 * c, 0, 16 /* sc1 */);
            }
          }
        } else if (PAT == 0 || PAT == 1) {
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
  const int tiles_n = N / 256, tiles = (M / 256) * tiles_n;
  if (tiles < 256) {  // one tile per workgroup, fewer workgroups than CUs: what a single M = 2048 launch writes (incl. the end-of-kernel write-back)
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0);
      probe<PAT><<<tiles, 512>>>(out, N, tiles_n, 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-40s %dx%d (%d workgroups): %8.1f us  %.2f TB/s\n", name, M, N, tiles, best * 1e3, (double)M * N * 4 / (best * 1e-3) / 1e12);
    return;
  }
  const int reps = tiles / 256;
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
  // a single launch's output: 2048 x 3072 fp32 = 25 MB from 96 workgroups (hipEvent pair around ONE launch: includes launch latency ~2-3 us)
  run<0>("A, one qkv-sized output", out, 2048, 3072);
  run<1>("B, one qkv-sized output", out, 2048, 3072);
  run<3>("B + nontemporal, qkv-sized", out, 2048, 3072);
  run<4>("B + sc1 write-through, qkv-sized", out, 2048, 3072);
  run<3>("B + nontemporal, big", out, M, N);
  run<0>("A, 2048 x 1024 (8 MB)", out, 2048, 1024);
  run<1>("B, 2048 x 1024 (8 MB)", out, 2048, 1024);
  run<3>("B + nontemporal, 2048 x 1024", out, 2048, 1024);
  run<0>("A, 2048 x 4096 (33 MB)", out, 2048, 4096);
  return 0;
}
