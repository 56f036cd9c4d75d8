// Do the matrix pipe and the vector ALU of a gfx950 SIMD run beside each other?
//
// Two findings of the kernels ask for the hardware fact: the N-channel composite (the time of
// v_mfma_f32_32x32x2_f32 ADDS to the VALU time, interleaved or not) and the ViT attention (round 3 ablation: removing the bf16 MFMAs
// saves exactly their pipe time, 10 of 38.5 us, although they are issued between the softmax instructions).  This program times, on
// every CU at once, workgroups of 8 waves (two per SIMD) that issue
//   M   only matrix instructions (independent accumulators, back to back)
//   V   only vector instructions (independent v_fma_f32 chains; or v_exp_f32)
//   MV  both, interleaved in EVERY wave (one matrix instruction, then its share of vector instructions)
//   M|V both, by wave: waves 0-3 only matrix, waves 4-7 only vector instructions (each SIMD holds one wave of each kind)
// for the bf16 instruction of the GEMM / attention kernels (v_mfma_f32_32x32x16_bf16, 8 passes) and for the f32 one of the composite
// (v_mfma_f32_32x32x2_f32, 16 passes).  The vector share per matrix instruction is chosen so that both pipes have the same work
// (bf16: 8 FMAs = 32 cycles per wave; f32: 16 FMAs = 64 cycles).  Overlap shows as t(MV) ~ max(t(M), t(V)), none as t(M) + t(V).
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pipe_overlap_probe.hip -o /tmp/pipe_probe && /tmp/pipe_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// KIND 0: bf16 32x32x16 (NV = 8 vector instructions per matrix instruction), 1: f32 32x32x2 (NV = 16).  VOP 0: v_fma_f32, 1: v_exp_f32
// mode 0 = M, 1 = V, 2 = MV (every wave both), 3 = M|V (waves 0-3 matrix, 4-7 vector), 4 = M|V (even waves matrix, odd waves vector)
template <int KIND, int VOP, int NV = (KIND == 0 ? 8 : 16)>
__global__ __launch_bounds__(512) void probe(int mode, int iters, float* sink, float seed) {
  const int wave = threadIdx.x >> 6;
  const bool do_m = mode == 0 || mode == 2 || (mode == 3 && wave < 4) || (mode == 4 && (wave & 1) == 0);
  const bool do_v = mode == 1 || mode == 2 || (mode == 3 && wave >= 4) || (mode == 4 && (wave & 1) == 1);
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = seed + 0.001f * (float)(j + threadIdx.x);
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = b[e] = (__bf16)(seed + 0.01f * e);
  const float fa = seed, fb = seed * 0.5f;
  const float m1 = 0.999f, a1 = 0.001f;
  // (three loop bodies: the choice is wave-uniform and made outside the loop, so that no branch sits between the instructions timed)
  if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m1), "v"(a1));
          else asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        }
        // (the volatile statements keep their order among themselves; tying the accumulator in keeps the matrix instruction between them)
        asm volatile("" : "+v"(acc[i]));
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
        asm volatile("" : "+v"(acc[i]));
      }
    }
  } else if (do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m1), "v"(a1));
          else asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j];
  if (s == 12345.678f) sink[0] = s;
}

template <int KIND, int VOP, int NV = (KIND == 0 ? 8 : 16)>
static float run(int mode, int iters, float* sink, int threads = 512) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<KIND, VOP, NV>), dim3(256), dim3(threads), 0, 0, mode, iters, sink, 0.5f);  // warm-up
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<KIND, VOP, NV>), dim3(256), dim3(threads), 0, 0, mode, iters, sink, 0.5f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  return best * 1e3f;
}

template <int KIND, int VOP>
static void table(const char* what, int iters, float* sink) {
  const float tm = run<KIND, VOP>(0, iters, sink), tv = run<KIND, VOP>(1, iters, sink), tmv = run<KIND, VOP>(2, iters, sink);
  const float tm1 = run<KIND, VOP>(0, iters, sink, 256), tv1 = run<KIND, VOP>(1, iters, sink, 256);  // four waves: one per SIMD
  const float t3 = run<KIND, VOP>(3, iters, sink), t4 = run<KIND, VOP>(4, iters, sink);
  printf("%s\n", what);
  printf("   2 waves / SIMD, same work in both:   M %7.1f us   V %7.1f us   MV (every wave issues both) %7.1f us = %.2f x (M + V) = %.2f x max(M, V)\n", tm, tv, tmv,
         tmv / (tm + tv), tmv / (tm > tv ? tm : tv));
  printf("   1 wave / SIMD (4-wave workgroups):   M %7.1f us   V %7.1f us\n", tm1, tv1);
  printf("   by wave (8 waves, half of them M):   waves 0-3 M, 4-7 V %7.1f us   even waves M, odd waves V %7.1f us   (sum of the 1-wave times %.1f, max %.1f)\n", t3, t4,
         tm1 + tv1, tm1 > tv1 ? tm1 : tv1);
}

// the bf16 instruction with NV vector instructions behind each: what share of the smaller pipe's time is hidden, at two waves and at one wave per SIMD
template <int NV>
static void sweep(int iters, float* sink) {
  const float tm = run<0, 0, NV>(0, iters, sink), tv = run<0, 0, NV>(1, iters, sink), tmv = run<0, 0, NV>(2, iters, sink);
  const float tm1 = run<0, 0, NV>(0, iters, sink, 256), tv1 = run<0, 0, NV>(1, iters, sink, 256), tmv1 = run<0, 0, NV>(2, iters, sink, 256);
  auto hid = [](float m, float v, float mv) { return (m + v - mv) / (m < v ? m : v); };
  printf("   %2d v_fma_f32 per MFMA:  2 waves / SIMD  M %7.1f  V %7.1f  MV %7.1f us (%.0f %% of the smaller pipe's time hidden)   |   1 wave / SIMD  M %7.1f  V %7.1f  MV %7.1f us (%.0f %%)\n",
         NV, tm, tv, tmv, 100.f * hid(tm, tv, tmv), tm1, tv1, tmv1, 100.f * hid(tm1, tv1, tmv1));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* sink;
  CK(hipMalloc(&sink, 64));
  printf("256 workgroups x 8 waves (2 per SIMD), %d iterations x 4 matrix instructions per wave (+ their vector share)\n", iters);
  table<0, 0>("v_mfma_f32_32x32x16_bf16 + 8 v_fma_f32 each", iters, sink);
  table<0, 1>("v_mfma_f32_32x32x16_bf16 + 8 v_exp_f32 each", iters, sink);
  table<1, 0>("v_mfma_f32_32x32x2_f32  + 16 v_fma_f32 each", iters, sink);
  table<1, 1>("v_mfma_f32_32x32x2_f32  + 16 v_exp_f32 each", iters, sink);
  printf("v_mfma_f32_32x32x16_bf16 followed by NV independent v_fma_f32 in every wave\n");
  sweep<4>(iters, sink);
  sweep<8>(iters, sink);
  sweep<12>(iters, sink);
  sweep<16>(iters, sink);
  sweep<24>(iters, sink);
  sweep<32>(iters, sink);
  return 0;
}
