// Minimal cross-wave test of the packed add: in every workgroup waves 0-3 ("victims", one per SIMD) run
//     v_pk_add_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]        (d.lo = a.lo + b.hi, d.hi = a.hi + b.lo)
// in a loop and check every result; waves 4-7 ("aggressors", the same four SIMDs) run one kind of instruction in a loop:
//   0 nothing   1 v_cvt_pk_bf16_f32   2 v_perm_b32   3 v_pk_mul_f32   4 v_pk_add_f32 (no op_sel)   5 bf16 MFMA   6 the hi / lo split of the
//   bf16x3 GEMMs (and, sub, perm, cvt_pk)   7 v_pk_fma_f32
// hipcc --offload-arch=gfx950 -O3 tools/probes/pk_hazard/xwave.hip -o /tmp/xwave && /tmp/xwave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void xwave(unsigned long long* err, unsigned long long* first, int iters, int mode, int victim_form) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {  // victims
    unsigned long long bad = 0;
    f2 a = {1.0f + lane, 2.0f + 0.5f * lane}, b = {10.0f + 3.0f * lane, 20.0f + 7.0f * lane};
    for (int it = 0; it < iters; ++it) {
      f2 d;
      if (victim_form == 0)
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
      else if (victim_form == 1)
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      else
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
      float e0, e1;
      if (victim_form == 0) { e0 = a.x + b.y; e1 = a.y + b.x; }
      else if (victim_form == 1) { e0 = a.x + b.x; e1 = a.y + b.y; }
      else { e0 = a.x * b.y; e1 = a.y * b.x; }
      asm volatile("" : "+v"(e0), "+v"(e1));
      if (d.x != e0 || d.y != e1) {
        if (!bad) { first[0] = ((unsigned long long)__float_as_uint(d.x) << 32) | __float_as_uint(d.y); first[1] = ((unsigned long long)__float_as_uint(e0) << 32) | __float_as_uint(e1); }
        ++bad;
      }
      a.x += 0.25f; b.y -= 0.125f;
    }
    if (bad) atomicAdd(err, bad);
  } else {  // aggressors
    float x = 1.0f + lane, y = 2.0f + lane;
    uint32_t u = 0x3f800000u + lane, w = 0;
    f2 p = {x, y}, q = {y, x};
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(1.0f + j); fb[j] = (__bf16)(0.5f); }
    for (int it = 0; it < iters; ++it) {
      switch (mode) {
        case 1: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x), "v"(y)); x += 1.f; break;
        case 2: asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(u), "v"(w), "v"(0x07060302u)); u += 3; break;
        case 3: asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(q)); break;
        case 4: asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(q)); break;
        case 5: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0); break;
        case 6: {
          uint32_t hi;
          float lo0, lo1;
          asm volatile("v_perm_b32 %0, %3, %4, %5\n\tv_and_b32 %1, 0xffff0000, %4\n\tv_and_b32 %2, 0xffff0000, %3\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %3, %2\n\tv_cvt_pk_bf16_f32 %1, %1, %2"
                       : "=&v"(hi), "=&v"(lo0), "=&v"(lo1) : "v"(x), "v"(y), "v"(0x07060302u));
          w ^= hi ^ __float_as_uint(lo0);
          x += 1.f;
        } break;
        case 7: asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(p), "v"(q)); break;
        default: break;
      }
    }
    float s = p.x + p.y + x + acc[0];
    if (s == 12345.678f || w == 0xdeadbeefu) first[3] = w;
  }
}

int main() {
  unsigned long long *err, *first;
  hipMalloc(&err, 8); hipMalloc(&first, 32);
  const char* names[] = {"nothing", "v_cvt_pk_bf16_f32", "v_perm_b32", "v_pk_mul_f32", "v_pk_add_f32 (plain)", "bf16 MFMA 32x32x16", "hi / lo split sequence (and, sub, perm, cvt_pk)", "v_pk_fma_f32"};
  const char* vn[] = {"v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 (no op_sel)", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]"};
  for (int vf = 0; vf < 3; ++vf)
    for (int mode = 0; mode < 8; ++mode) {
      unsigned long long h[5] = {0, 0, 0, 0, 0};
      hipMemset(err, 0, 8); hipMemset(first, 0, 32);
      const int iters = 200000;
      hipLaunchKernelGGL(xwave, dim3(1024), dim3(512), 0, 0, err, first, iters, mode, vf);
      hipDeviceSynchronize();
      hipMemcpy(h, err, 8, hipMemcpyDeviceToHost); hipMemcpy(h + 1, first, 32, hipMemcpyDeviceToHost);
      printf("victim %-44s | aggressor %-48s : %llu wrong of %.3g results", vn[vf], names[mode], h[0], 1024.0 * 4 * 64 * iters);
      if (h[0]) printf("   first: got (%08llx, %08llx) expected (%08llx, %08llx)", h[1] >> 32, h[1] & 0xffffffffull, h[2] >> 32, h[2] & 0xffffffffull);
      printf("\n");
    }
  return 0;
}
