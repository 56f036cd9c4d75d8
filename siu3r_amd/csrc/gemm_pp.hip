// Host side of the 8-wave ping-pong GEMM (kernel template: gemm_pp_kernel.h; one tile shape and operand mode per translation unit: gemm_pp_t{1,2,3}{x,b}.hip)
// and the skinny remainder-row kernel.
#include "gemm_pp_kernel.h"

namespace siu3r_gemm_pp {
// ---- the last <= 32 rows of a dense problem (M = 2 x 1025 tokens = 8 x 256 + 2: a ninth row of tiles that holds two rows would cost a
// second round of workgroups).  A workgroup multiplies those rows by 64 columns over the whole K: its 8 waves take K slices, load
// their MFMA fragments straight from global memory (no LDS ring: the W panel is streamed once, 16 bytes per lane, everything of a
// slice in flight at once), and wave 0 adds the partial blocks through LDS and runs the row pass.  N / 64 workgroups of ~4 us.
template <bool X3, bool LNF>
__global__ __launch_bounds__(512, 2) void gemm_skinny_kernel(const siu3r_gemm_params p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int ESZ = X3 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * 8192 + siu3r_epi_pp::WAVE_STAGE_BYTES];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int M = p.m, N = p.n, K = p.k, kpad = p.kpad;
  const int row0 = p.m_main, col0 = blockIdx.x * 64, z = blockIdx.z;
  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const unsigned char* Ab = (const unsigned char*)p.a + zof.a * ESZ;
  const unsigned char* Wb = X3 ? (const unsigned char*)p.w_x3 + zof.w * 4 : (const unsigned char*)p.w_hi + zof.w * 2;
  const int WROW = X3 ? kpad * 4 : kpad * 2;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, (short)0, (int)(((int64_t)(M - 1) * p.lda + K) * ESZ), RSRC_FLAGS);
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, (short)0, (int)((int64_t)N * WROW), RSRC_FLAGS);
  int m = row0 + l31;
  if (m > M - 1) m = M - 1;
  // k16 steps; A: the lane half's 8 values of row m; W: rows n0, n0 + 32 (bf16x3: hi and lo halves of the [hi 32 | lo 32] segment)
  const unsigned a_voff = (unsigned)((int64_t)m * p.lda * ESZ + lh * (8 * ESZ));
  unsigned w_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int n = col0 + j * 32 + l31;
    if (n > N - 1) n = N - 1;
    w_voff[j] = (unsigned)((int64_t)n * WROW + lh * 16);
  }
  const int ns_all = kpad / 16;                     // k16 steps (kpad % 64 == 0)
  const int per = ((ns_all / 4 + 7) / 8) * 4;       // steps per wave, a multiple of 4
  const int s_begin = wave * per, s_end = min(ns_all, s_begin + per);
  f32x16 acc[1][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
  for (int s0 = s_begin; s0 < s_end; s0 += 4) {
    u32x4 fa[4][2], fw[4][2][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int sg = s0 + u;
      const bool kin = sg * 16 + lh * 8 < K;  // (K % 8 == 0; W is zero padded, A must not be read beyond K)
      if (X3) {
        fa[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rA, kin ? a_voff : OOB, sg * 64, 0);
        fa[u][1] = __builtin_amdgcn_raw_buffer_load_b128(rA, kin ? a_voff + 16 : OOB, sg * 64, 0);
      } else {
        fa[u][0] = __builtin_amdgcn_raw_buffer_load_b128(rA, kin ? a_voff : OOB, sg * 32, 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (X3) {
          const int so = (sg >> 1) * 128 + (sg & 1) * 32;
          fw[u][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rW, w_voff[j], so, 0);
          fw[u][j][1] = __builtin_amdgcn_raw_buffer_load_b128(rW, w_voff[j] + 64, so, 0);
        } else {
          fw[u][j][0] = __builtin_amdgcn_raw_buffer_load_b128(rW, w_voff[j], sg * 32, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      union U8 { u32x4 u; bf16x8 h; };
      if (X3) {
        U8 ah, al, bh[2], bl[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int x0 = e < 2 ? fa[u][0][2 * e] : fa[u][1][2 * e - 4], x1 = e < 2 ? fa[u][0][2 * e + 1] : fa[u][1][2 * e - 3];
          ah.u[e] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
          al.u[e] = pack_bf16x2(__uint_as_float(x0) - __uint_as_float(x0 & 0xffff0000u), __uint_as_float(x1) - __uint_as_float(x1 & 0xffff0000u));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bh[j].u = fw[u][j][0];
          bl[j].u = fw[u][j][1];
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bh[j].h, acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.h, bl[j].h, acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.h, bh[j].h, acc[0][j], 0, 0, 0);
        }
      } else {
        U8 a, b;
        a.u = fa[u][0];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b.u = fw[u][j][0];
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc[0][j], 0, 0, 0);
        }
      }
    }
  }
  // partial blocks -> wave 0 (same lane, same register <-> same address: conflict-free 16-byte accesses)
  float* part = (float*)smem;
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int qv = 0; qv < 4; ++qv) {
        float4 v = make_float4(acc[0][j][4 * qv], acc[0][j][4 * qv + 1], acc[0][j][4 * qv + 2], acc[0][j][4 * qv + 3]);
        *(float4*)(part + ((wave * 8 + j * 4 + qv) * 64 + lane) * 4) = v;
      }
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll 1
  for (int w = 1; w < 8; ++w) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int qv = 0; qv < 4; ++qv) {
        const float4 v = *(const float4*)(part + ((w * 8 + j * 4 + qv) * 64 + lane) * 4);
        acc[0][j][4 * qv] += v.x;
        acc[0][j][4 * qv + 1] += v.y;
        acc[0][j][4 * qv + 2] += v.z;
        acc[0][j][4 * qv + 3] += v.w;
      }
  }
  siu3r_epi_pp::wave_rows<1, 2, LNF>(p, acc, (float*)(smem + 8 * 8192), row0, col0, M, z, lane);
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
  snprintf(buf, n, "siu3r_gemm_pp::gemm_pp_kernel<%s, %d, %d, %d, %s, %s>", x3 ? "true" : "false", cfg == 3 ? 1 : 2, cfg == 1 ? 4 : 2, mode,
           (mode == 1 && p.relu_in) ? "true" : "false", (mode == 0 && p.ln_stats) ? "true" : "false");
}

// tiled launch with tile cfg (SIU3R_TILE_PP_*); p.splitk, p.m_main as planned
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
  if (x3 && lnf) hipLaunchKernelGGL((gemm_skinny_kernel<true, true>), grid, block, 0, s, p);
  else if (x3) hipLaunchKernelGGL((gemm_skinny_kernel<true, false>), grid, block, 0, s, p);
  else if (lnf) hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_skinny_kernel<false, false>), grid, block, 0, s, p);
#endif
  SIU3R_LAUNCH_CHECK("siu3r_gemm(skinny)");
  return 0;
}
