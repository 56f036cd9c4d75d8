// Shared device/host helpers for libsiu3r_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/siu3r_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

void siu3r_set_error(const char* fmt, ...);

#define SIU3R_CHECK(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      siu3r_set_error(__VA_ARGS__);   \
      return 1;                       \
    }                                 \
  } while (0)

#define SIU3R_LAUNCH_CHECK(name)                                          \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      siu3r_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return 2;                                                           \
    }                                                                     \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> packed bf16 pair (x in the low half): one v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
__device__ __host__ inline uint32_t pack_bf16x2(float x, float y) {
#if __HIP_DEVICE_COMPILE__
  f32x2_t v = {x, y};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#else
  auto rne = [](float f) -> uint32_t {
    union { float f; uint32_t u; } v;
    v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (v.u >> 16) | 0x40;
    return (v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16;
  };
  return (rne(x) & 0xffffu) | (rne(y) << 16);
#endif
}
__device__ __host__ inline u16 f32_to_bf16_bits(float f) {
#if __HIP_DEVICE_COMPILE__
  return (u16)(pack_bf16x2(f, 0.f) & 0xffffu);
#endif
  union { float f; uint32_t u; } v;
  v.f = f;
  uint32_t u = v.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((u >> 16) & 1u);
  return (u16)((u + r) >> 16);
}
__device__ __host__ inline float bf16_bits_to_f32(u16 b) {
  union { float f; uint32_t u; } v;
  v.u = ((uint32_t)b) << 16;
  return v.f;
}

// load/store one scalar of runtime dtype
__device__ inline float load_as_f32(const void* p, int dtype, int64_t i) {
  return dtype == SIU3R_F32 ? ((const float*)p)[i] : bf16_bits_to_f32(((const u16*)p)[i]);
}
__device__ inline void store_from_f32(void* p, int dtype, int64_t i, float v) {
  if (dtype == SIU3R_F32)
    ((float*)p)[i] = v;
  else
    ((u16*)p)[i] = f32_to_bf16_bits(v);
}

// 8 consecutive elements (16 B bf16 / 32 B f32), aligned
struct f32x8 {
  float v[8];
};
__device__ inline f32x8 load8_as_f32(const void* p, int dtype, int64_t i) {
  f32x8 r;
  if (dtype == SIU3R_F32) {
    const float4* q = (const float4*)((const float*)p + i);
    float4 a = q[0], b = q[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  } else {
    uint4 a = *(const uint4*)((const u16*)p + i);
    uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r.v[2 * j] = bf16_bits_to_f32((u16)(w[j] & 0xffff));
      r.v[2 * j + 1] = bf16_bits_to_f32((u16)(w[j] >> 16));
    }
  }
  return r;
}
__device__ inline void store8_from_f32(void* p, int dtype, int64_t i, const f32x8& r) {
  if (dtype == SIU3R_F32) {
    float4* q = (float4*)((float*)p + i);
    q[0] = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
  } else {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = pack_bf16x2(r.v[2 * j], r.v[2 * j + 1]);
    *(uint4*)((u16*)p + i) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// pack 8 floats to bf16x8 (hi) and optionally the residual (lo)
__device__ inline uint4 pack_bf16x8(const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    w[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ inline void split_bf16x8(const float* f, uint4& hi, uint4& lo) {
  float r[8];
  uint32_t h[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
    r[2 * j] = f[2 * j] - bf16_bits_to_f32((u16)(h[j] & 0xffffu));
    r[2 * j + 1] = f[2 * j + 1] - bf16_bits_to_f32((u16)(h[j] >> 16));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = pack_bf16x8(r);
}

// the split of the GEMM K loop and of the pre-split planes (siu3r_gemm_params.c_x3): hi = the UPPER 16 BITS, lo = bf16(f - hi)
__device__ inline void split_trunc_bf16x8(const float* f, uint4& hi, uint4& lo) {
  float r[8];
  uint32_t h[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t u0 = __float_as_uint(f[2 * j]), u1 = __float_as_uint(f[2 * j + 1]);
#if __HIP_DEVICE_COMPILE__
    h[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
#else
    h[j] = (u1 & 0xffff0000u) | (u0 >> 16);
#endif
    r[2 * j] = f[2 * j] - __uint_as_float(u0 & 0xffff0000u);
    r[2 * j + 1] = f[2 * j + 1] - __uint_as_float(u1 & 0xffff0000u);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = pack_bf16x8(r);
}

__device__ inline bf16x8 as_bf16x8(uint4 u) {
  union { uint4 u; bf16x8 b; } c;
  c.u = u;
  return c.b;
}

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// blockIdx.z -> (outer, inner) batch index and the operand offsets of siu3r_gemm_params' two-level batching
struct siu3r_zoff {
  int zo, zi;
  int64_t a, w, c, r, bias;
};
__device__ __forceinline__ siu3r_zoff siu3r_batch_offsets(const siu3r_gemm_params& p, int z) {
  siu3r_zoff o;
  if (p.bmod > 0) {
    o.zo = z / p.bmod;
    o.zi = z - o.zo * p.bmod;
    o.a = (int64_t)o.zo * p.sa + (int64_t)o.zi * p.sa_i;
    o.c = (int64_t)o.zo * p.sc + (int64_t)o.zi * p.sc_i;
    o.r = (int64_t)o.zo * p.sr + (int64_t)o.zi * p.sr_i;
    o.w = (int64_t)o.zi * p.sw;
    o.bias = (int64_t)o.zi * p.sbias;
  } else {
    o.zo = z;
    o.zi = 0;
    o.a = (int64_t)z * p.sa;
    o.c = (int64_t)z * p.sc;
    o.r = (int64_t)z * p.sr;
    o.w = (int64_t)z * p.sw;
    o.bias = 0;
  }
  return o;
}
