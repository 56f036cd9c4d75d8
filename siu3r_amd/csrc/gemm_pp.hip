// Host side of the 8-wave ping-pong GEMM (kernel template: gemm_pp_kernel.h; one tile shape and operand mode per translation unit: gemm_pp_t{1,2,3}{x,b}.hip)
// and the skinny remainder-row kernel.
#include "gemm_pp_kernel.h"

namespace siu3r_gemm_pp {
// the remainder rows as a launch of their own (the ping-pong kernels can also carry them as extra workgroups: gemm_pp_kernel.h)
template <bool X3, bool LNF>
__global__ __launch_bounds__(512, 2) void gemm_skinny_kernel(const siu3r_gemm_params p) {
#if __HIP_DEVICE_COMPILE__
  __shared__ __attribute__((aligned(16))) unsigned char smem[SKINNY_SMEM_BYTES];
  skinny_rows_body<X3, LNF>(p, smem, blockIdx.x, blockIdx.z);
#endif
}

// ---- the same remainder rows when there are at most FOUR of them (M = 2 x 1025 = 8 x 256 + 2; 1025 = 4 x 256 + 1 per decoder side) ------------
// (used for long K only, 2048 <= kpad <= 4096: see the launcher.)
// A matrix-vector product has no use for MFMA: the W panel is the only traffic, and the fragment-shaped loads of the kernel above fetch
// it as 64 scattered 16-byte pieces per instruction (12.8 us at K = 1024, ~40 at K = 4096 -- which kept fc2 on a ninth row of tiles).  Here a wave streams each W row in fully coalesced 1 KiB loads (8 columns' worth of one chunk in flight behind the 8 being
// consumed), the <= 4 A rows sit in LDS as fp32 and are read once per chunk for all 8 columns, products are plain fp32 FMAs of the
// operands (bf16x3: a = hi + lo as the MFMA path splits it; a * w_hi and a * w_lo accumulate separately), a lane-level partial sum per
// (column, row) is reduced across the wave at the end, and wave 0 hands the 64-column block to the common row pass in accumulator layout.
constexpr int GV_ROWS = 4, GV_KMAX = 4096;
template <bool X3, bool LNF>
__global__ __launch_bounds__(512) void gemm_skinny_gemv_kernel(const siu3r_gemm_params p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int ESZ = X3 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float s_a[GV_ROWS * GV_KMAX];
  __shared__ __attribute__((aligned(16))) float s_res[GV_ROWS][64];
  __shared__ __attribute__((aligned(16))) unsigned char s_stage[siu3r_epi_pp::WAVE_STAGE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = p.m, N = p.n, K = p.k, kpad = p.kpad;
  const int row0 = p.m_main, R = M - row0, col0 = blockIdx.x * 64, z = blockIdx.z;
  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const unsigned char* Ab = (const unsigned char*)p.a + zof.a * ESZ;
  const unsigned char* Wb = X3 ? (const unsigned char*)p.w_x3 + zof.w * 4 : (const unsigned char*)p.w_hi + zof.w * 2;
  const int WROW = X3 ? kpad * 4 : kpad * 2;
  // A rows -> LDS (fp32, zero beyond K)
  for (int i = t; i < R * (kpad / 4); i += 512) {
    const int r = i / (kpad / 4), k = (i - r * (kpad / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) {  // K % 4 == 0 (launcher)
      const unsigned char* src = Ab + ((int64_t)(row0 + r) * p.lda + k) * ESZ;
      if (X3 && p.a_x3) {  // pre-split planes: a = hi + lo (the 16-bit-mantissa value the MFMA path multiplies)
        const unsigned char* seg = Ab + (int64_t)(row0 + r) * p.lda * 4 + (k >> 5) * 128 + (k & 31) * 2;
        const uint2 h = *(const uint2*)seg, l = *(const uint2*)(seg + 64);
        v = make_float4(__uint_as_float(h.x << 16) + __uint_as_float(l.x << 16), __uint_as_float(h.x & 0xffff0000u) + __uint_as_float(l.x & 0xffff0000u),
                        __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16), __uint_as_float(h.y & 0xffff0000u) + __uint_as_float(l.y & 0xffff0000u));
      } else if (X3) {
        // (the same value from fp32 rows: hi + bf16(a - hi), so that a launch gives the same bits whichever form its A operand has)
        v = *(const float4*)src;
        float* vf = (float*)&v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hi = __uint_as_float(__float_as_uint(vf[e]) & 0xffff0000u);
          vf[e] = hi + __uint_as_float(pack_bf16x2(vf[e] - hi, 0.f) << 16);
        }
      } else {
        const uint2 u = *(const uint2*)src;
        v = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
      }
    }
    *(float4*)(s_a + r * kpad + k) = v;
  }
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, (short)0, (int)((int64_t)N * WROW), RSRC_FLAGS);
  const int nchunk = (WROW + 1023) / 1024;
  unsigned voff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int n = col0 + wave * 8 + c;
    if (n > N - 1) n = N - 1;
    voff[c] = (unsigned)((int64_t)n * WROW + lane * 16);
  }
  // a lane's 16 bytes of a chunk: bf16x3 -> 8 values of ONE plane (hi or lo) of k0 .. k0 + 8 in the [hi 32 | lo 32] segment; bf16 -> 8 consecutive k
  const int k_lane = X3 ? (lane >> 3) * 32 + (lane & 3) * 8 : lane * 8;
  constexpr int K_PER_CHUNK = X3 ? 256 : 512;
  float acc[8][GV_ROWS];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) acc[c][r] = 0.f;
  auto issue = [&](u32x4 (&w)[8], int ch) {
    const bool in = ch * 1024 + lane * 16 < WROW;
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = __builtin_amdgcn_raw_buffer_load_b128(rW, in ? voff[c] : OOB, ch * 1024, 0);
  };
  __syncthreads();
  auto consume = [&](const u32x4 (&w)[8], int ch) {
    int k0 = ch * K_PER_CHUNK + k_lane;
    if (k0 > kpad - 8) k0 = 0;  // (a lane beyond the row end loaded zeros)
    float a[GV_ROWS][8];
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) {
      if (r < R) {
        const float4 x = *(const float4*)(s_a + r * kpad + k0), y = *(const float4*)(s_a + r * kpad + k0 + 4);
        a[r][0] = x.x; a[r][1] = x.y; a[r][2] = x.z; a[r][3] = x.w; a[r][4] = y.x; a[r][5] = y.y; a[r][6] = y.z; a[r][7] = y.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[r][e] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float wv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wv[2 * e] = __uint_as_float(w[c][e] << 16);
        wv[2 * e + 1] = __uint_as_float(w[c][e] & 0xffff0000u);
      }
#pragma unroll
      for (int r = 0; r < GV_ROWS; ++r)
        if (r < R) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[c][r] = __builtin_fmaf(a[r][e], wv[e], acc[c][r]);
        }
    }
  };
  u32x4 w0[8], w1[8];
  issue(w0, 0);
  for (int ch = 0; ch < nchunk; ch += 2) {
    if (ch + 1 < nchunk) issue(w1, ch + 1);
    consume(w0, ch);
    if (ch + 2 < nchunk) issue(w0, ch + 2);
    if (ch + 1 < nchunk) consume(w1, ch + 1);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int r = 0; r < GV_ROWS; ++r) {
      float v = acc[c][r];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) s_res[r][wave * 8 + c] = v;
    }
  __syncthreads();
  if (wave != 0) return;
  // accumulator layout of a 32 x 32 block: lane -> column lane % 32, register i -> row 8 (i / 4) + 4 (lane / 32) + i % 4
  f32x16 out[1][2];
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 8 * (i >> 2) + 4 * lh + (i & 3);
      out[0][j][i] = row < GV_ROWS ? s_res[row][j * 32 + l31] : 0.f;
    }
  siu3r_epi_pp::wave_rows<1, 2, LNF>(p, out, (float*)s_stage, row0, col0, M, z, lane);
#endif
}

}  // namespace siu3r_gemm_pp

#define SIU3R_PP_DECL(T) void siu3r_gemm_pp_go_##T(const siu3r_gemm_params& p, int mode, bool lnf, dim3 grid, hipStream_t s)
SIU3R_PP_DECL(t1x); SIU3R_PP_DECL(t1b); SIU3R_PP_DECL(t2x); SIU3R_PP_DECL(t2b); SIU3R_PP_DECL(t3x); SIU3R_PP_DECL(t3b);
#undef SIU3R_PP_DECL

// ---- host side: applicability, names, launches.  The decision (family, tile, split-K, skinny rows) is siu3r_gemm_plan's (gemm.hip).
// A-mode the ping-pong kernels run this problem in (0 dense, 1 conv tap cursor, 2 small-cin conv), or -1: outside their range
int siu3r_gemm_pp_mode(const siu3r_gemm_params& p) {
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  const bool bf = !x3 && !p.w_lo && p.a_dtype == SIU3R_BF16;
  if (!x3 && !bf) return -1;
  const int64_t lim = 0xfffff000ll;
  const int esz = x3 ? 4 : 2, kstep = x3 ? 16 : 32;
  if ((int64_t)p.n * p.kpad * (x3 ? 4 : 2) >= lim) return -1;
  if (p.a_mode == 0) {
    if (((int64_t)(p.m - 1) * p.lda + p.k) * esz >= lim || p.relu_in) return -1;
    return 0;
  }
  if (p.a_mode != 1 || p.ln_stats) return -1;
  const int64_t img = (int64_t)(p.m / (p.oh * p.ow)) * p.ih * p.iw * p.cin * esz + (int64_t)(p.pad * p.iw + p.pad) * p.cin * esz;
  if (img >= lim || p.cin % (16 / esz) != 0) return -1;
  if (p.cin % kstep == 0 && p.kh * p.kw <= 31) return 1;
  return p.relu_in ? -1 : 2;
}

void siu3r_gemm_pp_name(const siu3r_gemm_params& p, int cfg, char* buf, int n) {
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  const int mode = siu3r_gemm_pp_mode(p);
  snprintf(buf, n, "siu3r_gemm_pp::gemm_pp_kernel<%s, %d, %d, %d, %s, %s%s>", x3 ? "true" : "false", cfg == 3 ? 1 : 2, cfg == 1 ? 4 : 2, mode,
           (mode == 1 && p.relu_in) ? "true" : "false", (mode == 0 && p.ln_stats) ? "true" : "false", (x3 && mode <= 1 && p.a_x3) ? ", true" : "");
}

// can a ping-pong launch of this problem read A as pre-split planes (siu3r_gemm_params.a_x3)?  bf16x3, dense rows, whole 64-deep K
// tiles (no zero-padded tail inside a [hi 32 | lo 32] segment), rows and batch items that start on a segment boundary
bool siu3r_gemm_pp_a_x3_ok(const siu3r_gemm_params& p) {
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  if (!x3 || ((uintptr_t)p.a % 16) != 0 || p.sa % 32 != 0 || p.sa_i % 32 != 0) return false;
  const int mode = siu3r_gemm_pp_mode(p);
  if (mode == 0) return p.k == p.kpad && p.lda % 32 == 0;
  return mode == 1 && p.cin % 32 == 0 && !p.relu_in;  // the tap-cursor gather over an NHWC map of whole segments (a zero-padded K tail lands on a masked tap)
}
// can it write its output as planes (c_x3)?  The fast fp32 row pass of gemm_epilogue_pp.h must be the one that runs (wave_rows()'s
// conditions, for every batch item), and rows / batch items must start on a segment boundary
bool siu3r_gemm_pp_c_x3_ok(const siu3r_gemm_params& p, int cfg) {
  if (siu3r_epi_pp::g_force_general || siu3r_gemm_pp_mode(p) < 0) return false;
  const int MI_ = cfg == 3 ? 1 : 2;
  const int64_t res_b = p.r_dtype == SIU3R_F32 ? 4 : 2;
  const bool res_ok = !p.residual || (p.r_dtype == SIU3R_F32 && (p.ldr * res_b) % 16 == 0 && ((uintptr_t)p.residual % 16) == 0 && (p.sr * res_b) % 16 == 0 &&
                                       (p.sr_i * res_b) % 16 == 0 && (int64_t)(32 * MI_) * p.ldr < (1 << 28));
  const bool up_ok = !p.up_src || (p.up_dtype == SIU3R_F32 && ((uintptr_t)p.up_src % 16) == 0 && (int64_t)p.m * p.n < (1 << 29));
  return p.out_mode == 0 && up_ok && !p.c_aux && p.c_dtype == SIU3R_F32 && p.n % 64 == 0 && (int64_t)(32 * MI_) * p.ldc < (1 << 28) && res_ok &&
         p.ldc % 32 == 0 && p.sc % 32 == 0 && p.sc_i % 32 == 0 && ((uintptr_t)p.c % 16) == 0 && ((uintptr_t)p.c_x3 % 16) == 0 &&
         p.c_x3_col0 >= 0 && p.c_x3_col0 % 64 == 0;
}

// tiled launch with tile cfg (SIU3R_TILE_PP_*); p.splitk, p.m_main as planned
// does the ping-pong launch of this problem carry its remainder rows itself (dense A, MFMA remainder rows: short K)?
static const bool g_no_fold = getenv("SIU3R_GEMM_NO_FOLD") != nullptr;
bool siu3r_gemm_pp_folds_skinny(const siu3r_gemm_params& p) {
  return !g_no_fold && p.m_main > 0 && p.m_main < p.m && p.a_mode == 0 && p.kpad <= 2048 && siu3r_gemm_pp_mode(p) == 0;
}

int siu3r_gemm_pp_launch(const siu3r_gemm_params& pin, int cfg, void* stream) {
  using namespace siu3r_gemm_pp;
  siu3r_gemm_params p = pin;
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  const int mode = siu3r_gemm_pp_mode(p);
  if (mode < 0) return 1;
  const int MI_ = cfg == 3 ? 1 : 2, NJ_ = cfg == 1 ? 4 : 2;
  const int bm = 128 * MI_, bn = 64 * NJ_;
  const int mrows = p.m_main > 0 ? p.m_main : p.m;
  const int tm = (mrows + bm - 1) / bm, tn = (p.n + bn - 1) / bn;
  // 8 regions (gy x gx), one per XCD: first the factorisation whose largest region holds the fewest tiles (an XCD with more workgroups
  // than CUs runs a second round while the others idle: 5 row tiles cut 2 x 4 put 18 tiles per batch item on four XCDs and 12 on the
  // rest), then the smallest per-XCD operand footprint
  int best = -1;
  long best_cost = 0;
  for (int gx = 1; gx <= 8; gx *= 2) {
    const int gy = 8 / gx;
    const int rm = (tm + gy - 1) / gy, rn = (tn + gx - 1) / gx;
    const long cost = (long)rm * rn * 1000000 + (long)rm * bm + (long)rn * bn;
    if (best < 0 || cost < best_cost) {
      best = gx;
      best_cost = cost;
    }
  }
  p.map_gx = best;
  p.map_rm = (tm + (8 / best) - 1) / (8 / best);
  p.map_rn = (tn + best - 1) / best;
  dim3 grid(8 * p.map_rm * p.map_rn, p.splitk > 1 ? p.splitk : 1, p.batch > 0 ? p.batch : 1), block(512);
  p.sk_gx = 0;
  if (siu3r_gemm_pp_folds_skinny(p)) {
    p.sk_gx = (int)grid.x;
    grid.x += (unsigned)((p.n + 63) / 64);
  }
  hipStream_t s = (hipStream_t)stream;
  const bool lnf = p.ln_stats != nullptr;
#ifdef SIU3R_PP_MINI  // tuning builds (tools/ab_pp.sh): one instantiation, compiled here
  hipLaunchKernelGGL((gemm_pp_kernel<SIU3R_PP_MINI, false, false>), grid, block, 0, s, p);
#else
  // one translation unit per tile shape and operand mode (they compile in parallel): gemm_pp_t1{x,b}.hip 256 x 256, _t2 256 x 128, _t3 128 x 128
  if (cfg == 1) (x3 ? siu3r_gemm_pp_go_t1x : siu3r_gemm_pp_go_t1b)(p, mode, lnf, grid, s);
  else if (cfg == 2) (x3 ? siu3r_gemm_pp_go_t2x : siu3r_gemm_pp_go_t2b)(p, mode, lnf, grid, s);
  else (x3 ? siu3r_gemm_pp_go_t3x : siu3r_gemm_pp_go_t3b)(p, mode, lnf, grid, s);
#endif
  SIU3R_LAUNCH_CHECK("siu3r_gemm(pp)");
  return 0;
}

// rows [p.m_main, p.m) (at most 32) of a dense problem
int siu3r_gemm_skinny_launch(const siu3r_gemm_params& p, void* stream) {
  using namespace siu3r_gemm_pp;
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  dim3 grid((p.n + 63) / 64, 1, p.batch > 0 ? p.batch : 1), block(512);
  hipStream_t s = (hipStream_t)stream;
  const bool lnf = p.ln_stats != nullptr;
#ifndef SIU3R_PP_MINI
  // at most four rows: the matrix-vector kernel (16-byte aligned A rows, K % 4 == 0, the rows fit its LDS)
  const int rows = p.m - p.m_main, esz = x3 ? 4 : 2;
  static const bool no_gemv = getenv("SIU3R_GEMM_NO_GEMV") != nullptr;
  // (measured, tools/mb_skinny.py: the MFMA kernel above wins at K = 1024 -- 12.7 vs 14.7 us, both mostly fixed cost -- and loses at
  // K = 4096: 37 vs 26 us)
  if (!no_gemv && rows >= 1 && rows <= GV_ROWS && p.kpad >= 2048 && p.kpad <= GV_KMAX && p.k % 4 == 0 && ((int64_t)p.lda * esz) % 16 == 0 &&
      ((uintptr_t)p.a % 16) == 0 && (p.sa * esz) % 16 == 0 && (p.sa_i * esz) % 16 == 0) {
    if (x3 && lnf) hipLaunchKernelGGL((gemm_skinny_gemv_kernel<true, true>), grid, block, 0, s, p);
    else if (x3) hipLaunchKernelGGL((gemm_skinny_gemv_kernel<true, false>), grid, block, 0, s, p);
    else if (lnf) hipLaunchKernelGGL((gemm_skinny_gemv_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_skinny_gemv_kernel<false, false>), grid, block, 0, s, p);
    SIU3R_LAUNCH_CHECK("siu3r_gemm(skinny gemv)");
    return 0;
  }
  if (x3 && lnf) hipLaunchKernelGGL((gemm_skinny_kernel<true, true>), grid, block, 0, s, p);
  else if (x3) hipLaunchKernelGGL((gemm_skinny_kernel<true, false>), grid, block, 0, s, p);
  else if (lnf) hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_skinny_kernel<false, false>), grid, block, 0, s, p);
#endif
  SIU3R_LAUNCH_CHECK("siu3r_gemm(skinny)");
  return 0;
}
