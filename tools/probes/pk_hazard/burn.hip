// Neighbour for tools/probes/pk_hazard/standalone.py: an MFMA + LDS + VMEM busy loop shaped like the library's ping-pong GEMM workgroups
// (512 threads, one workgroup per CU, ~100 KB of LDS, 8 independent accumulator chains per wave) -- no meaning, only load.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
extern "C" __global__ __launch_bounds__(512) void burn_mfma(const uint4* src, float* sink, int iters, int mode) {
  __shared__ uint4 s[6144];  // 96 KB
  const int t = threadIdx.x;
  for (int i = t; i < 6144; i += 512) s[i] = src[(blockIdx.x * 6144 + i) & 0xfffff];
  __syncthreads();
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    union { uint4 u; bf16x8 b; } a, b;
    a.u = s[(t * 7 + it * 13) % 6144];
    b.u = s[(t * 3 + it * 29 + 1) % 6144];
    if (mode & 2) a.u.x ^= src[(blockIdx.x * 512 + t + it * 4096) & 0xfffff].x;  // + a stream of global loads
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, acc[j], 0, 0, 0);
    if (mode & 1) {  // + vector work between the MFMAs
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j][0] = acc[j][0] * 1.0001f + 0.5f;
    }
  }
  float v = 0.f;
  for (int j = 0; j < 8; ++j)
    for (int r = 0; r < 16; ++r) v += acc[j][r];
  if (v == 1234.5f) sink[blockIdx.x * 512 + t] = v;
}

// One-wave workgroups of bf16 MFMAs and nothing else: launched 2048-wide they put two MFMA-issuing waves on every SIMD and leave
// registers, LDS and wave slots for any other kernel's workgroups -- a neighbour that is certain to share SIMDs with them.
extern "C" __global__ __launch_bounds__(64) void burn_wave_dep(float* sink, int iters) {  // the same with ONE dependent accumulator chain
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + j + threadIdx.x); b[j] = (__bf16)(0.5f); }
  for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  if (acc[0] == 1234.5f) sink[blockIdx.x] = acc[0];
}
extern "C" __global__ __launch_bounds__(64) void burn_wave(float* sink, int iters) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + j + threadIdx.x); b[j] = (__bf16)(0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float v = 0.f;
  for (int j = 0; j < 4; ++j) v += acc[j][0];
  if (v == 1234.5f) sink[blockIdx.x] = v;
}
