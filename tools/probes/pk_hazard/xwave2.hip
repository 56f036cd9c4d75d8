// Which forms of a packed fp32 instruction go wrong beside another wave's MFMAs (follow-up of xwave.hip): victims (waves 0-3) run ONE form
// of v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 in a loop and check both halves; aggressors (waves 4-7, the same SIMDs) run one MFMA type.
// hipcc --offload-arch=gfx950 -O3 tools/probes/pk_hazard/xwave2.hip -o /tmp/xwave2 && /tmp/xwave2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define FORMS(X) \
  X(0, "", 0, 0, 1, 1) \
  X(1, " op_sel:[0,1]", 0, 1, 1, 1) \
  X(2, " op_sel:[1,0]", 1, 0, 1, 1) \
  X(3, " op_sel:[1,1]", 1, 1, 1, 1) \
  X(4, " op_sel_hi:[1,0]", 0, 0, 1, 0) \
  X(5, " op_sel_hi:[0,1]", 0, 0, 0, 1) \
  X(6, " op_sel_hi:[0,0]", 0, 0, 0, 0) \
  X(7, " op_sel:[0,1] op_sel_hi:[1,0]", 0, 1, 1, 0) \
  X(8, " op_sel:[1,0] op_sel_hi:[0,1]", 1, 0, 0, 1)

template <int F>
__device__ __forceinline__ unsigned long long victim(int iters, int lane, int op, unsigned long long* first, int same_wave) {
  unsigned long long bad = 0;
  f32x16 vacc;
  for (int r = 0; r < 16; ++r) vacc[r] = 0.f;
  bf16x8 va, vb;
  for (int j = 0; j < 8; ++j) { va[j] = (__bf16)(1.0f + j); vb[j] = (__bf16)(0.25f); }
  f2 a = {1.0f + lane, 2.0f + 0.5f * lane}, b = {10.0f + 3.0f * lane, 20.0f + 7.0f * lane};
  for (int it = 0; it < iters; ++it) {
    f2 d;
    if (same_wave) vacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, vacc, 0, 0, 0);  // the victim's OWN MFMA right in front of the packed op
    int s0l = 0, s1l = 0, s0h = 1, s1h = 1;
#define X(ID, TXT, A, B, C, D) \
    if (F == ID) { s0l = A; s1l = B; s0h = C; s1h = D; \
      if (op == 0) asm volatile("v_pk_add_f32 %0, %1, %2" TXT : "=v"(d) : "v"(a), "v"(b)); \
      else asm volatile("v_pk_mul_f32 %0, %1, %2" TXT : "=v"(d) : "v"(a), "v"(b)); }
    FORMS(X)
#undef X
    const float x0 = s0l ? a.y : a.x, x1 = s1l ? b.y : b.x, y0 = s0h ? a.y : a.x, y1 = s1h ? b.y : b.x;
    float e0 = op == 0 ? x0 + x1 : x0 * x1, e1 = op == 0 ? y0 + y1 : y0 * y1;
    asm volatile("" : "+v"(e0), "+v"(e1));
    if (d.x != e0 || d.y != e1) {
      if (!bad) { first[0] = ((unsigned long long)(d.x != e0) << 1) | (unsigned long long)(d.y != e1); }
      ++bad;
    }
    a.x += 0.25f; b.y -= 0.125f; a.y += 0.5f; b.x += 0.375f;
  }
  if (vacc[0] == 12345.678f) first[2] = 1;
  return bad;
}

__global__ __launch_bounds__(512) void xwave2(unsigned long long* err, unsigned long long* first, int iters, int form, int op, int agg, int same_wave) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    unsigned long long bad = 0;
    switch (form) {
#define X(ID, TXT, A, B, C, D) case ID: bad = victim<ID>(iters, lane, op, first, same_wave); break;
      FORMS(X)
#undef X
    }
    if (bad) atomicAdd(err, bad);
  } else {
    f32x16 acc; f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(1.0f + j); fb[j] = (__bf16)(0.5f); }
    float sa = 1.f + lane, sb2 = 0.5f;
    for (int it = 0; it < iters; ++it) {
      if (agg == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      else if (agg == 1) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc4, 0, 0, 0);
      else if (agg == 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa, sb2, acc, 0, 0, 0);
      else if (agg == 3) { asm volatile("s_nop 7"); }
    }
    if (acc[0] + acc4[0] == 12345.678f) first[3] = 1;
  }
}

int main() {
  unsigned long long *err, *first;
  (void)hipMalloc(&err, 8); (void)hipMalloc(&first, 32);
  const char* fn[] = {
#define X(ID, TXT, A, B, C, D) TXT,
      FORMS(X)
#undef X
  };
  const char* an[] = {"bf16 MFMA 32x32x16", "bf16 MFMA 16x16x32", "f32 MFMA 32x32x2", "s_nop"};
  for (int op = 0; op < 2; ++op)
    for (int agg = 0; agg < 4; ++agg)
      for (int form = 0; form < 9; ++form) {
        unsigned long long h[5] = {0, 0, 0, 0, 0};
        (void)hipMemset(err, 0, 8); (void)hipMemset(first, 0, 32);
        const int iters = 100000;
        hipLaunchKernelGGL(xwave2, dim3(1024), dim3(512), 0, 0, err, first, iters, form, op, agg, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, err, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h + 1, first, 32, hipMemcpyDeviceToHost);
        printf("%s %-52s | %-20s : %10llu wrong of %.3g%s\n", op ? "mul" : "add", fn[form][0] ? fn[form] : " (default selection)", an[agg], h[0], 1024.0 * 4 * 64 * iters,
               h[0] ? (h[1] == 2 ? "  (low half)" : h[1] == 1 ? "  (high half)" : "  (both halves)") : "");
      }
  // the victim's own MFMAs (no aggressor waves: they idle): is the hazard also inside one wave?
  for (int op = 0; op < 2; ++op)
    for (int form = 0; form < 9; ++form) {
      unsigned long long h[5] = {0, 0, 0, 0, 0};
      (void)hipMemset(err, 0, 8); (void)hipMemset(first, 0, 32);
      const int iters = 100000;
      hipLaunchKernelGGL(xwave2, dim3(1024), dim3(512), 0, 0, err, first, iters, form, op, 3, 1);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, err, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h + 1, first, 32, hipMemcpyDeviceToHost);
      printf("%s %-52s | %-20s : %10llu wrong of %.3g%s\n", op ? "mul" : "add", fn[form][0] ? fn[form] : " (default selection)", "OWN bf16 MFMA in front", h[0], 1024.0 * 4 * 64 * iters,
             h[0] ? (h[1] == 2 ? "  (low half)" : h[1] == 1 ? "  (high half)" : "  (both halves)") : "");
    }
  return 0;
}
