// Epilogue of the 8-wave ping-pong GEMM (gemm_pp.hip): the same fused operations as gemm_epilogue.h (bias, folded LayerNorm, GELU /
// ReLU, residual, RoPE2D, conv-transpose pixel shuffle, x2 bilinear upsample-add, row statistics, bf16 copy, deterministic split-K)
// for 128 MI x 64 NJ tiles held by 4 x 2 waves of 32 MI x 32 NJ each.
//
// Wave-private staging: a wave transposes ONE 32-row x 64-column pair of accumulator blocks at a time through its own 8.5 KiB of LDS
// (lane = column  ->  lane = (row, 8-column chunk)), so that every global access is a 16-byte vector and a tile row is written in
// 256-byte runs, and then moves on to its next pair.  No workgroup barrier after the first one, the accumulators are released pair by
// pair (the 256 x 256 kernel holds 128 of them: a workgroup-wide row pass on top of those spilled ~200 registers), and a 64-column
// group (one RoPE head, one row-statistics group) is always inside one pair.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"  // gelu_fast

namespace siu3r_epi_pp {

#ifndef SIU3R_EPI_GENERAL
#define SIU3R_EPI_GENERAL 0  // A/B builds: 1 = every launch takes the general row pass
#endif
constexpr bool g_force_general = SIU3R_EPI_GENERAL != 0;
constexpr int NT = 512;
constexpr int LDW = 68;                           // floats per staged row (64 + 4: 16-byte aligned rows, spreads banks)
constexpr int WAVE_STAGE_BYTES = 32 * LDW * 4 + 64 * 2 * 4;  // 32 x 64 block pair + (mean, rstd) of the wave's 64 rows
constexpr int staging_bytes() { return 8 * WAVE_STAGE_BYTES + 64; }

// Split-K over workgroups (deterministic): slabs as 16-byte write-through stores, relaxed ticket; the tile's last arriver sums the slabs
// in slice order into acc, resets the ticket for the next launch that uses this counter (same stream: see ops.py) and returns true;
// the other slices return false.
template <int MI, int NJ>
__device__ __forceinline__ bool splitk_reduce(const siu3r_gemm_params& p, f32x16 (&acc)[MI][NJ], unsigned char* smem, int64_t tile_id, int t) {
  constexpr int BN = 64 * NJ, BM = 128 * MI;
  int* s_flag = (int*)(smem + 8 * WAVE_STAGE_BYTES);
#if __HIP_DEVICE_COMPILE__
  {
    const int S = p.splitk;
    float* slabs = p.sk_ws + tile_id * S * (BM * BN);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    constexpr int NV = MI * NJ * 4;
    __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, (short)0, (int)(S * BM * BN * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          union { u32x4_t u; float f[4]; } v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v.f[e] = acc[i][j][4 * q + e];
          const int vec = (i * NJ + j) * 4 + q;
          __builtin_amdgcn_raw_buffer_store_b128(v.u, rs_, (int)((((int)blockIdx.y * NV + vec) * NT + t) * 16), 0, 16 /* sc1 */);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) *s_flag = __hip_atomic_fetch_add(p.sk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_flag != S - 1) return false;
    if (t == 0) __hip_atomic_store(p.sk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int sl = 0; sl < S; ++sl) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int vec = (i * NJ + j) * 4 + q;
            union { u32x4_t u; float f[4]; } v;
            v.u = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(((sl * NV + vec) * NT + t) * 16), 0, 16 /* sc1 */);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v.f[e];
          }
    }
  }
#endif
  return true;
}


// Fast row pass for the common output form -- row-major fp32 C with N % 64 == 0 and 16-byte aligned rows, optional fp32 residual of the
// same kind, no pixel shuffle / upsample-add / bf16 copy -- which is every GEMM of the encoder, the decoder and most of the heads in the
// bf16x3 mode.  The general pass below decides all of that per element chunk (64-bit index arithmetic, alignment tests, dtype switches:
// ~2000 executed instructions per wave and tile, as long as the stores themselves take); here
//   * rows are addressed through buffer resources over the wave's own rows with 32-bit offsets, an out-of-range offset standing in for
//     "row beyond M" (stores dropped, loads return 0: no branches in the row loop);
//   * a lane holds columns [4c, 4c+4) and [32+4c, 32+4c+4) of the 64-column group instead of [8c, 8c+8): each store instruction then
//     writes one FULL 128-byte line per row (two half-filled lines before: 15-25 % slower on a launch-sized burst, tools/probes/store_probe.hip);
//   * the feature tests (RoPE, statistics, activation, residual) are wave-uniform and sit outside the element loops.
template <int MI, int NJ, bool LNF, bool BF>
__device__ __forceinline__ void wave_rows_fast(const siu3r_gemm_params& p, f32x16 (&acc)[MI][NJ], float* ws, int row_w0, int col_w0, int m_end, int z,
                                               int lane, const siu3r_zoff& zof) {
#if __HIP_DEVICE_COMPILE__
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  typedef __attribute__((ext_vector_type(4))) float f32x4_t;
  const int l31 = lane & 31, lh = lane >> 5;
  const int prow = lane >> 3, c = lane & 7;
  const float* addp = LNF ? p.ln_c2 + zof.bias : (p.bias ? p.bias + zof.bias : nullptr);
  const float* c1p = LNF ? p.ln_c1 + zof.bias : nullptr;
  float* w_ln = ws + 32 * LDW;
  const int M = m_end, N = p.n;
  if (LNF) {
    if (lane < 32 * MI) {
      const int m = row_w0 + lane;
      float mu = 0.f, rstd = 0.f;
      if (m < M) {
        const float2* sp = (const float2*)p.ln_stats + ((int64_t)zof.zo * p.ln_sz + (int64_t)zof.zi * p.ln_sz_i + (int64_t)m * p.ln_ldm) * p.ln_tiles;
        const int Cn = p.k, last = Cn - 64 * (p.ln_tiles - 1);
        float2 part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = i < p.ln_tiles ? sp[i] : make_float2(0.f, 0.f);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < p.ln_tiles) sum += part[i].x * (float)(i == p.ln_tiles - 1 ? last : 64);
        mu = sum / (float)Cn;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < p.ln_tiles) {
            const float d = part[i].x - mu;
            m2 += part[i].y + d * d * (float)(i == p.ln_tiles - 1 ? last : 64);
          }
        rstd = rsqrtf(m2 / (float)Cn + p.ln_eps);
      }
      w_ln[2 * lane] = mu;
      w_ln[2 * lane + 1] = rstd;
    }
  }
  constexpr unsigned OOB = 0x80000000u;  // beyond num_records: the store is dropped, the load returns 0
  const bool res_bf = p.r_dtype == SIU3R_BF16;
  // c_x3: the output ALSO (or, c == NULL, only) as pre-split bf16x3 planes for the next GEMM's A operand (siu3r_hip.h): same row offsets
  // as the fp32 output, a 32-column segment = one 128-byte line [hi 32 | lo 32]
  // c_x3_col0: planes for the 64-column groups at or behind that column only; with c_x3 == c those groups are NOT written as fp32 (a mixed
  // buffer: the q | k | v projection keeps q in fp32 and hands k, v to the attention kernel pre-split)
  const bool st_c_any = p.c != nullptr, x3o_any = !BF && p.c_x3 != nullptr, mixed = x3o_any && p.c_x3 == p.c;
  const bool st_c = st_c_any;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned char*)(st_c ? p.c : p.c_x3) + (zof.c + (int64_t)row_w0 * p.ldc) * (BF ? 2 : 4)), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned char*)(x3o_any ? p.c_x3 : p.c) + (zof.c + (int64_t)row_w0 * p.ldc) * 4), (short)0, x3o_any ? 0x7fffffff : 0, 0x00020000);
  const bool has_res = p.residual != nullptr;
  const int r_esz = res_bf ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_res ? (const unsigned char*)p.residual + (zof.r + (int64_t)row_w0 * p.ldr) * r_esz : (const unsigned char*)p.c), (short)0, has_res ? 0x7fffffff : 0, 0x00020000);
  // 8 residual values of a row group: fp32 as two 16-byte loads at the lane's two column runs, bf16 (BF layout only) as one
  auto load_res = [&](unsigned roff, f32x4_t& lo, f32x4_t& hi) {
    if (BF && res_bf) {
      const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        lo[2 * e] = __builtin_bit_cast(float, w[e] << 16);
        lo[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xffff0000u);
        hi[2 * e] = __builtin_bit_cast(float, w[2 + e] << 16);
        hi[2 * e + 1] = __builtin_bit_cast(float, w[2 + e] & 0xffff0000u);
      }
    } else {
      lo = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 0, 0));
      hi = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rr, roff, BF ? 16 : 128, 0));
    }
  };
  // x2 bilinear upsample-add (align_corners; the DPT stems, dpt_block.py:230-235): fp32 source map [b, oh/2, ow/2, N] behind a buffer resource
  const bool up = !BF && p.up_src != nullptr;
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(up ? p.up_src : p.c), (short)0, up ? 0x7fffffff : 0, 0x00020000);
  const int act = p.act;
  const bool stats = p.stats_out != nullptr;
  const bool upper = c >= 4;  // this lane's columns are the second member of their RoPE pairs (d & 16)
  const int pc4 = 4 * (c ^ 4);
  // BF (bf16 C): a lane keeps the 8 consecutive columns [8c, 8c+8) -- 16 bytes, so that 8 lanes already fill a 128-byte line -- and
  // RoPE / statistics are not offered (fp32 outputs only: the general pass takes those)
  const int o_lo = BF ? 8 * c : 4 * c, o_hi = BF ? 8 * c + 4 : 32 + 4 * c;

#pragma unroll
  for (int jp = 0; jp < NJ / 2; ++jp) {
    const int g0 = col_w0 + jp * 64;  // the 64-column group (one RoPE head, one statistics group)
    if (g0 < N) {                     // (wave-uniform; N % 64 == 0: a group is whole or absent)
      const int n_lo = g0 + o_lo, n_hi = g0 + o_hi;
      const bool x3o = x3o_any && g0 >= p.c_x3_col0;  // (wave-uniform)
      const bool c_here = st_c_any && !(mixed && x3o);
      const bool rope = !BF && p.rope_cos != nullptr && g0 < p.rope_ncols;
      f32x4_t b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = b_lo, c1_lo = b_lo, c1_hi = b_lo;
      if (addp) {
        b_lo = *(const f32x4_t*)(addp + n_lo);
        b_hi = *(const f32x4_t*)(addp + n_hi);
      }
      if (LNF) {
        c1_lo = *(const f32x4_t*)(c1p + n_lo);
        c1_hi = *(const f32x4_t*)(c1p + n_hi);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int mblk = row_w0 + i * 32;
        if (mblk < M) {  // (wave-uniform)
          // ---- global reads of the block pair's four row groups go out first (residual, RoPE positions) where the registers allow it:
          // the 256 x 256 tile holds 128 accumulators and reads them per row group instead
          constexpr bool HOIST = MI * NJ <= 4;
          f32x4_t r_lo[HOIST ? 4 : 1], r_hi[HOIST ? 4 : 1];
          int pos0[4], pos1[4];
          unsigned coff[4];
          bool rok[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int lr = i * 32 + k * 8 + prow;
            const bool ok = row_w0 + lr < M;
            rok[k] = ok;
            coff[k] = (ok && c_here) ? (unsigned)((lr * (int)p.ldc + n_lo) * (BF ? 2 : 4)) : OOB;
            if constexpr (HOIST) {
              const unsigned roff = ok ? (unsigned)((lr * (int)p.ldr + n_lo) * r_esz) : OOB;
              load_res(roff, r_lo[k], r_hi[k]);
            }
            pos0[k] = pos1[k] = 0;
            if (rope && ok) {
              const int64_t* pp = p.rope_pos + ((int64_t)z * p.m + row_w0 + lr) * 2;
              pos0[k] = (int)pp[0];
              pos1[k] = (int)pp[1];
            }
          }
          // ---- transpose the pair through the wave's LDS (same wave: LDS operations complete in order, a counted wait is the only
          // synchronisation; the previous pair's reads were waited for before its last use)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDW + jj * 32 + l31] = acc[i][2 * jp + jj][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int lr = i * 32 + k * 8 + prow;
            const float* rowp = ws + (k * 8 + prow) * LDW;
            f32x4_t v_lo = *(const f32x4_t*)(rowp + o_lo), v_hi = *(const f32x4_t*)(rowp + o_hi);
            f32x4_t q_lo = v_lo, q_hi = v_hi;
            if (rope) {
              q_lo = *(const f32x4_t*)(rowp + pc4);
              q_hi = *(const f32x4_t*)(rowp + 32 + pc4);
            }
            if (k == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area may be rewritten
            float ln_mu = 0.f, ln_rs = 1.f;
            if (LNF) {
              ln_mu = w_ln[2 * lr];
              ln_rs = w_ln[2 * lr + 1];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v_lo[e] = ln_rs * (v_lo[e] - ln_mu * c1_lo[e]) + b_lo[e];
                v_hi[e] = ln_rs * (v_hi[e] - ln_mu * c1_hi[e]) + b_hi[e];
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v_lo[e] += b_lo[e];
                v_hi[e] += b_hi[e];
              }
            }
            if (rope) {
              // the partner columns' bias / LayerNorm terms (cache hits; kept out of the registers that live across the row loop)
              f32x4_t pb_lo = {0.f, 0.f, 0.f, 0.f}, pb_hi = pb_lo, pc1_lo = pb_lo, pc1_hi = pb_lo;
              if (addp) {
                pb_lo = *(const f32x4_t*)(addp + g0 + pc4);
                pb_hi = *(const f32x4_t*)(addp + g0 + 32 + pc4);
              }
              if (LNF) {
                pc1_lo = *(const f32x4_t*)(c1p + g0 + pc4);
                pc1_hi = *(const f32x4_t*)(c1p + g0 + 32 + pc4);
              }
              // columns d = 4c + e (axis 0, positions pos0) and 32 + 4c + e (axis 1, pos1); table entry d & 15, partner d ^ 16
              const f32x4_t cs0 = *(const f32x4_t*)(p.rope_cos + (int64_t)pos0[k] * 16 + 4 * (c & 3)), sn0 = *(const f32x4_t*)(p.rope_sin + (int64_t)pos0[k] * 16 + 4 * (c & 3));
              const f32x4_t cs1 = *(const f32x4_t*)(p.rope_cos + (int64_t)pos1[k] * 16 + 4 * (c & 3)), sn1 = *(const f32x4_t*)(p.rope_sin + (int64_t)pos1[k] * 16 + 4 * (c & 3));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float o0 = LNF ? ln_rs * (q_lo[e] - ln_mu * pc1_lo[e]) + pb_lo[e] : q_lo[e] + pb_lo[e];
                const float o1 = LNF ? ln_rs * (q_hi[e] - ln_mu * pc1_hi[e]) + pb_hi[e] : q_hi[e] + pb_hi[e];
                v_lo[e] = upper ? (v_lo[e] * cs0[e] + o0 * sn0[e]) : (v_lo[e] * cs0[e] - o0 * sn0[e]);
                v_hi[e] = upper ? (v_hi[e] * cs1[e] + o1 * sn1[e]) : (v_hi[e] * cs1[e] - o1 * sn1[e]);
              }
            }
            if (act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v_lo[e] = siu3r_epi::gelu_fast(v_lo[e]);
                v_hi[e] = siu3r_epi::gelu_fast(v_hi[e]);
              }
            } else if (act == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v_lo[e] = fmaxf(v_lo[e], 0.f);
                v_hi[e] = fmaxf(v_hi[e], 0.f);
              }
            }
            if (up) {
              const int m = min(row_w0 + lr, M - 1);
              const int ohw = p.oh * p.ow, bi = m / ohw, rr_ = m - bi * ohw;
              const int oy = rr_ / p.ow, ox = rr_ - oy * p.ow;
              const int sh = p.oh >> 1, sw = p.ow >> 1;
              const float fy = (p.oh > 1) ? (float)(sh - 1) / (float)(p.oh - 1) * oy : 0.f;
              const float fx = (p.ow > 1) ? (float)(sw - 1) / (float)(p.ow - 1) * ox : 0.f;
              const int y0 = (int)fy, x0 = (int)fx;
              const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
              const float ly = fy - y0, lx = fx - x0;
              const int sb = bi * sh * sw;
              const unsigned u00 = (unsigned)(((sb + y0 * sw + x0) * N + n_lo) * 4), u01 = (unsigned)(((sb + y0 * sw + x1) * N + n_lo) * 4);
              const unsigned u10 = (unsigned)(((sb + y1 * sw + x0) * N + n_lo) * 4), u11 = (unsigned)(((sb + y1 * sw + x1) * N + n_lo) * 4);
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) {
                const f32x4_t a_ = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ru, u00, hh * 128, 0));
                const f32x4_t b_ = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ru, u01, hh * 128, 0));
                const f32x4_t c_ = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ru, u10, hh * 128, 0));
                const f32x4_t d_ = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ru, u11, hh * 128, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float add = (1.f - ly) * ((1.f - lx) * a_[e] + lx * b_[e]) + ly * ((1.f - lx) * c_[e] + lx * d_[e]);
                  if (hh == 0) v_lo[e] += add;
                  else v_hi[e] += add;
                }
              }
            }
            if constexpr (HOIST) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {  // (zeros without a residual: num_records 0)
                v_lo[e] += r_lo[k][e];
                v_hi[e] += r_hi[k][e];
              }
            } else if (has_res) {
              const unsigned roff = rok[k] ? (unsigned)((lr * (int)p.ldr + n_lo) * r_esz) : OOB;
              f32x4_t a_, b_;
              load_res(roff, a_, b_);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v_lo[e] += a_[e];
                v_hi[e] += b_[e];
              }
            }
            if (!BF && stats) {
              // (mean, centred sum of squares) of this row's 64-column group, reduced over the group's 8 lanes
              float s1 = ((v_lo[0] + v_lo[1]) + (v_lo[2] + v_lo[3])) + ((v_hi[0] + v_hi[1]) + (v_hi[2] + v_hi[3]));
              s1 += __shfl_xor(s1, 1);
              s1 += __shfl_xor(s1, 2);
              s1 += __shfl_xor(s1, 4);
              const float mean = s1 * (1.f / 64.f);
              float s2 = 0.f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float d0 = v_lo[e] - mean, d1 = v_hi[e] - mean;
                s2 += d0 * d0 + d1 * d1;
              }
              s2 += __shfl_xor(s2, 1);
              s2 += __shfl_xor(s2, 2);
              s2 += __shfl_xor(s2, 4);
              if (c == 0 && rok[k]) {
                const int64_t row = (int64_t)zof.zo * p.st_sz + (int64_t)zof.zi * p.st_sz_i + (int64_t)(row_w0 + lr) * p.st_ldm;
                ((float2*)p.stats_out)[row * (N >> 6) + (g0 >> 6)] = make_float2(mean, s2);
              }
            }
            if constexpr (BF) {
              u32x4_t w;
              w[0] = pack_bf16x2(v_lo[0], v_lo[1]);
              w[1] = pack_bf16x2(v_lo[2], v_lo[3]);
              w[2] = pack_bf16x2(v_hi[0], v_hi[1]);
              w[3] = pack_bf16x2(v_hi[2], v_hi[3]);
              __builtin_amdgcn_raw_buffer_store_b128(w, rc, coff[k], 0, 0);
            } else {
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v_lo), rc, coff[k], 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v_hi), rc, coff[k] + 128, 0, 0);  // (immediate offset: see the plane stores below)
              if (x3o) {
                // hi = upper 16 bits, lo = bf16(v - hi): the expressions of the K loop's in-register split (gemm_pp_kernel.h), so that a
                // consumer of the planes multiplies bit-identical operands.  A lane holds 4 values of each of the group's two segments =
                // 8 bytes of a plane; lane pairs (c, c ^ 1) trade halves so that the even lane stores 16 bytes of the hi plane and the odd
                // lane 16 bytes of the lo plane: 8 lanes x 16 B = the row's full 128-byte segment line per store instruction
                const bool odd = (c & 1) != 0;
                const unsigned xoff = rok[k] ? (unsigned)((lr * (int)p.ldc + g0) * 4 + (odd ? 64 : 0) + 8 * (c & ~1)) : OOB;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                  const f32x4_t v = hh == 0 ? v_lo : v_hi;
                  unsigned int b[4], hi2[2], lo2[2];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float f = v[e];  // (a copy: __builtin_bit_cast of the vector ELEMENT lvalue read element 0 for every e)
                    b[e] = __float_as_uint(f);
                  }
#pragma unroll
                  for (int e = 0; e < 2; ++e) {
                    hi2[e] = __builtin_amdgcn_perm(b[2 * e + 1], b[2 * e], 0x07060302u);
                    lo2[e] = pack_bf16x2(__uint_as_float(b[2 * e]) - __uint_as_float(b[2 * e] & 0xffff0000u), __uint_as_float(b[2 * e + 1]) - __uint_as_float(b[2 * e + 1] & 0xffff0000u));
                  }
                  unsigned int mine[2], got[2];
#pragma unroll
                  for (int e = 0; e < 2; ++e) {
                    mine[e] = odd ? lo2[e] : hi2[e];
                    got[e] = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(odd ? hi2[e] : lo2[e]), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
                  }
                  u32x4_t w;
                  w[0] = odd ? got[0] : mine[0];
                  w[1] = odd ? got[1] : mine[1];
                  w[2] = odd ? mine[0] : got[0];
                  w[3] = odd ? mine[1] : got[1];
                  // (the second line's +128 goes into the instruction's IMMEDIATE offset, not into soffset: with an SGPR soffset hipcc assumes
                  // that a 16-byte store's data registers may be overwritten by the very next VALU instruction -- the ISA manual says so --
                  // and the next row group's LDS address landed in w[0] of lanes 12..15 of every row of 16 before the store had read it:
                  // 448 wrong plane words of 4 M on the 256 x 256 tile, none on 256 x 128.  With an immediate offset the compiler keeps its
                  // wait state.)
                  __builtin_amdgcn_raw_buffer_store_b128(w, rx, xoff + hh * 128, 0, 0);
                }
              }
            }
          }
        }
      }
    }
  }
#endif
}

// Row pass of ONE wave: its 32 MI x 32 NJ accumulator blocks start at (row_w0, col_w0) of the output; rows >= m_end are not stored
// (they belong to another launch or lie beyond M).  ws: the wave's WAVE_STAGE_BYTES of LDS.
template <int MI, int NJ, bool LNF>
__device__ __forceinline__ void wave_rows(const siu3r_gemm_params& p, f32x16 (&acc)[MI][NJ], float* ws, int row_w0, int col_w0, int m_end, int z,
                                          int lane) {
  static_assert(NJ % 2 == 0, "block pairs");
  const int l31 = lane & 31, lh = lane >> 5;
  unsigned char* Cb = (unsigned char*)p.c;
  const unsigned char* Rb = (const unsigned char*)p.residual;
  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const int64_t c_boff = zof.c, r_boff = zof.r;
  {
    // (wave-uniform) the common output forms take the fast pass: row-major C, N % 64 == 0, 16-byte aligned rows, no pixel shuffle /
    // upsample-add / bf16 copy; fp32 C with everything else, bf16 C without RoPE
    const int64_t res_b = p.r_dtype == SIU3R_F32 ? 4 : 2;
    const bool res_ok = !p.residual || ((((int64_t)p.ldr * res_b) & 15) == 0 && ((r_boff * res_b + (int64_t)(uintptr_t)p.residual) & 15) == 0 &&
                                         (int64_t)(32 * MI) * p.ldr < (1 << 28) && (p.r_dtype == SIU3R_F32 || p.c_dtype == SIU3R_BF16));
    const bool up_ok = !p.up_src || (p.up_dtype == SIU3R_F32 && p.c_dtype == SIU3R_F32 && (((int64_t)(uintptr_t)p.up_src) & 15) == 0 &&
                                     (int64_t)p.m * p.n < (1 << 29) /* the half-resolution source is a quarter of the output: 4 B offsets < 2^31 */);
    const bool common = p.out_mode == 0 && up_ok && !p.c_aux && (p.n & 63) == 0 && (int64_t)(32 * MI) * p.ldc < (1 << 28) && res_ok && !g_force_general;
    if (common && p.c_dtype == SIU3R_F32 && (p.ldc & 3) == 0 && ((c_boff * 4 + (int64_t)(uintptr_t)p.c) & 15) == 0) {
      wave_rows_fast<MI, NJ, LNF, false>(p, acc, ws, row_w0, col_w0, m_end, z, lane, zof);
      return;
    }
    if (common && !p.up_src && p.c_dtype == SIU3R_BF16 && (p.ldc & 7) == 0 && ((c_boff * 2 + (int64_t)(uintptr_t)p.c) & 15) == 0 && !p.rope_cos && !p.stats_out) {
      wave_rows_fast<MI, NJ, LNF, true>(p, acc, ws, row_w0, col_w0, m_end, z, lane, zof);
      return;
    }
  }
  const float* biasp = p.bias ? p.bias + zof.bias : nullptr;
  constexpr bool ln = LNF;
  const float* c1p = ln ? p.ln_c1 + zof.bias : nullptr;
  const float* c2p = ln ? p.ln_c2 + zof.bias : nullptr;
  const float* addp = ln ? c2p : biasp;
  const int c_esz = p.c_dtype == SIU3R_F32 ? 4 : 2;
  const int r_esz = p.r_dtype == SIU3R_F32 ? 4 : 2;
  const int M = m_end, N = p.n;
  const int Mtot = p.m;  // rows per batch item (RoPE position index)
  float* w_ln = ws + 32 * LDW;                           // (mean, rstd) of the wave's 32 MI rows
  const int prow = lane >> 3, chunk = lane & 7;          // row pass: 8 rows x 8 chunks per step

  if (ln) {
    // lane r merges the 64-column (mean, M2) partials of row r of the wave (Chan's formula)
    if (lane < 32 * MI) {
      const int m = row_w0 + lane;
      float mu = 0.f, rstd = 0.f;
      if (m < M) {
        const float2* sp = (const float2*)p.ln_stats + ((int64_t)zof.zo * p.ln_sz + (int64_t)zof.zi * p.ln_sz_i + (int64_t)m * p.ln_ldm) * p.ln_tiles;
        const int Cn = p.k, last = Cn - 64 * (p.ln_tiles - 1);
        float2 part[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) part[i] = i < p.ln_tiles ? sp[i] : make_float2(0.f, 0.f);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < p.ln_tiles) sum += part[i].x * (float)(i == p.ln_tiles - 1 ? last : 64);
        mu = sum / (float)Cn;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < p.ln_tiles) {
            const float d = part[i].x - mu;
            m2 += part[i].y + d * d * (float)(i == p.ln_tiles - 1 ? last : 64);
          }
        rstd = rsqrtf(m2 / (float)Cn + p.ln_eps);
      }
      w_ln[2 * lane] = mu;
      w_ln[2 * lane + 1] = rstd;
    }
  }

#pragma unroll
  for (int jp = 0; jp < NJ / 2; ++jp) {
    // ---- column invariants of this 64-column group
    const int n0 = col_w0 + jp * 64 + chunk * 8;
    const bool col_ok = n0 < N;
    const int nv = col_ok ? min(8, N - n0) : 0;
    int co0 = n0, kidx = 0;
    if (p.out_mode == 1) {
      kidx = n0 / p.cout;
      co0 = n0 - kidx * p.cout;
    }
    const bool full = nv == 8 && (p.out_mode == 0 || co0 + 8 <= p.cout);
    const bool rope = p.rope_cos != nullptr && n0 < p.rope_ncols && col_ok;
    const int pc = chunk ^ 2;  // RoPE partner chunk: 16 columns away inside the 64-wide head
    const int d0 = n0 & 63, axis = d0 >> 5;
    const bool upper = (d0 & 16) != 0;
    float bias[8], pbias[8], c1[8], pc1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = pbias[e] = c1[e] = pc1[e] = 0.f;
    if (col_ok) {
      if (addp) {
        if (nv == 8 && (((uintptr_t)(addp + co0)) & 15) == 0) {
          const float4 a = *(const float4*)(addp + co0), b = *(const float4*)(addp + co0 + 4);
          bias[0] = a.x; bias[1] = a.y; bias[2] = a.z; bias[3] = a.w; bias[4] = b.x; bias[5] = b.y; bias[6] = b.z; bias[7] = b.w;
        } else {
          _Pragma("unroll") for (int e = 0; e < 8; ++e)
            if (e < nv) bias[e] = addp[co0 + e];
        }
      }
      if (ln) {
        _Pragma("unroll") for (int e = 0; e < 8; ++e)
          if (e < nv) c1[e] = c1p[n0 + e];
      }
      if (rope) {
        const int pn0 = n0 - chunk * 8 + pc * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (addp) pbias[e] = addp[pn0 + e];
          if (ln) pc1[e] = c1p[pn0 + e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mblk = row_w0 + i * 32;
      if (mblk < M) {  // (wave-uniform)
      // ---- global reads of the block pair's four row groups go out first: residual, RoPE positions
      f32x8 res[4];
      int64_t pos[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = mblk + k * 8 + prow;
        const bool ok = col_ok && m < M;
#pragma unroll
        for (int e = 0; e < 8; ++e) res[k].v[e] = 0.f;
        pos[k] = 0;
        if (!ok) continue;
        if (Rb && p.out_mode == 0) {
          const int64_t ridx = r_boff + (int64_t)m * p.ldr + n0;
          if (full && (((uintptr_t)Rb + ridx * r_esz) & 15) == 0) {
            res[k] = load8_as_f32(Rb, p.r_dtype, ridx);
          } else {
            _Pragma("unroll") for (int e = 0; e < 8; ++e)
              if (e < nv) res[k].v[e] = load_as_f32(Rb, p.r_dtype, ridx + e);
          }
        }
        if (rope) pos[k] = p.rope_pos[((int64_t)z * Mtot + m) * 2 + axis];
      }
      // ---- transpose the pair through the wave's LDS (same wave: LDS operations of a wave complete in order, a counted wait is the
      // only synchronisation needed; the previous pair's reads were waited for before its last use)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) ws[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDW + jj * 32 + l31] = acc[i][2 * jp + jj][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float4 rc[4][2], rs[4][2];
      if (rope) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float* cs_ = p.rope_cos + pos[k] * 16 + (d0 & 8);
          const float* sn_ = p.rope_sin + pos[k] * 16 + (d0 & 8);
          rc[k][0] = *(const float4*)cs_; rc[k][1] = *(const float4*)(cs_ + 4);
          rs[k][0] = *(const float4*)sn_; rs[k][1] = *(const float4*)(sn_ + 4);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int lr = i * 32 + k * 8 + prow;  // row inside the wave
        const int m = row_w0 + lr;
        float v[8], pv[8];
        {
          const float* rowp = ws + (k * 8 + prow) * LDW;
          const float4 a = *(const float4*)(rowp + chunk * 8), b = *(const float4*)(rowp + chunk * 8 + 4);
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = 0.f;
          if (rope) {
            const float4 c = *(const float4*)(rowp + pc * 8), d = *(const float4*)(rowp + pc * 8 + 4);
            pv[0] = c.x; pv[1] = c.y; pv[2] = c.z; pv[3] = c.w; pv[4] = d.x; pv[5] = d.y; pv[6] = d.z; pv[7] = d.w;
          }
        }
        if (k == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area may be rewritten
        if (m >= M) continue;                  // (uniform over the 8 lanes of a row)
        if (!col_ok && !p.stats_out) continue;
        float ln_mu = 0.f, ln_rs = 1.f;
        if (ln) {
          ln_mu = w_ln[2 * lr];
          ln_rs = w_ln[2 * lr + 1];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ln_rs * (v[e] - ln_mu * c1[e]) + bias[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias[e];
        }
        if (rope) {
          const float cc[8] = {rc[k][0].x, rc[k][0].y, rc[k][0].z, rc[k][0].w, rc[k][1].x, rc[k][1].y, rc[k][1].z, rc[k][1].w};
          const float ss[8] = {rs[k][0].x, rs[k][0].y, rs[k][0].z, rs[k][0].w, rs[k][1].x, rs[k][1].y, rs[k][1].z, rs[k][1].w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float o = ln ? ln_rs * (pv[e] - ln_mu * pc1[e]) + pbias[e] : pv[e] + pbias[e];
            v[e] = upper ? (v[e] * cc[e] + o * ss[e]) : (v[e] * cc[e] - o * ss[e]);
          }
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = siu3r_epi::gelu_fast(v[e]);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        int64_t oidx;
        if (p.out_mode == 0) {
          oidx = (int64_t)m * p.ldc + n0;
        } else {
          const int ihw = p.ih * p.iw;
          const int b = m / ihw, rr = m - b * ihw;
          const int iy = rr / p.iw, ix = rr - iy * p.iw;
          const int ky = kidx / p.up, kx = kidx - ky * p.up;
          oidx = (((int64_t)b * (p.ih * p.up) + iy * p.up + ky) * (p.iw * p.up) + ix * p.up + kx) * p.cout + co0;
        }
        if (p.up_src && col_ok) {
          const int ohw = p.oh * p.ow;
          const int b = m / ohw, rr = m - b * ohw;
          const int oy = rr / p.ow, ox = rr - oy * p.ow;
          const int sh = p.oh >> 1, sw = p.ow >> 1;
          const float fy = (p.oh > 1) ? (float)(sh - 1) / (float)(p.oh - 1) * oy : 0.f;
          const float fx = (p.ow > 1) ? (float)(sw - 1) / (float)(p.ow - 1) * ox : 0.f;
          const int y0 = (int)fy, x0 = (int)fx;
          const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
          const float ly = fy - y0, lx = fx - x0;
          const int64_t sb = (int64_t)b * sh * sw;
          const int64_t i00 = (sb + (int64_t)y0 * sw + x0) * N + n0, i01 = (sb + (int64_t)y0 * sw + x1) * N + n0;
          const int64_t i10 = (sb + (int64_t)y1 * sw + x0) * N + n0, i11 = (sb + (int64_t)y1 * sw + x1) * N + n0;
          if (full && (N & 7) == 0) {
            const f32x8 a = load8_as_f32(p.up_src, p.up_dtype, i00), b_ = load8_as_f32(p.up_src, p.up_dtype, i01);
            const f32x8 c = load8_as_f32(p.up_src, p.up_dtype, i10), d = load8_as_f32(p.up_src, p.up_dtype, i11);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              v[e] += (1.f - ly) * ((1.f - lx) * a.v[e] + lx * b_.v[e]) + ly * ((1.f - lx) * c.v[e] + lx * d.v[e]);
          } else {
            _Pragma("unroll") for (int e = 0; e < 8; ++e)
              if (e < nv) v[e] += (1.f - ly) * ((1.f - lx) * load_as_f32(p.up_src, p.up_dtype, i00 + e) + lx * load_as_f32(p.up_src, p.up_dtype, i01 + e)) +
                      ly * ((1.f - lx) * load_as_f32(p.up_src, p.up_dtype, i10 + e) + lx * load_as_f32(p.up_src, p.up_dtype, i11 + e));
          }
        }
        if (Rb) {
          if (p.out_mode == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += res[k].v[e];
          } else {
            _Pragma("unroll") for (int e = 0; e < 8; ++e)
              if (e < nv) v[e] += load_as_f32(Rb, p.r_dtype, r_boff + oidx + e);
          }
        }
        if (p.stats_out) {
          // (mean, centred sum of squares) of this row's 64-column group, reduced over the group's 8 lanes
          const int n64 = min(64, N - (n0 & ~63));
          float s1 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s1 += e < nv ? v[e] : 0.f;
          s1 += __shfl_xor(s1, 1);
          s1 += __shfl_xor(s1, 2);
          s1 += __shfl_xor(s1, 4);
          const float mean = s1 / (float)max(n64, 1);
          float s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = v[e] - mean;
            s2 += e < nv ? d * d : 0.f;
          }
          s2 += __shfl_xor(s2, 1);
          s2 += __shfl_xor(s2, 2);
          s2 += __shfl_xor(s2, 4);
          if (chunk == 0 && n64 > 0) {
            const int64_t row = (int64_t)zof.zo * p.st_sz + (int64_t)zof.zi * p.st_sz_i + (int64_t)m * p.st_ldm;
            ((float2*)p.stats_out)[row * ((N + 63) >> 6) + (n0 >> 6)] = make_float2(mean, s2);
          }
          if (!col_ok) continue;
        }
        const int64_t cidx = c_boff + oidx;
        if (full && (((uintptr_t)Cb + cidx * c_esz) & 15) == 0) {
          f32x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.v[e] = v[e];
          store8_from_f32(Cb, p.c_dtype, cidx, o);
          if (p.c_aux) store8_from_f32(p.c_aux, SIU3R_BF16, cidx, o);
        } else {
          _Pragma("unroll") for (int e = 0; e < 8; ++e)
            if (e < nv) {
              store_from_f32(Cb, p.c_dtype, cidx + e, v[e]);
              if (p.c_aux) store_from_f32(p.c_aux, SIU3R_BF16, cidx + e, v[e]);
            }
        }
      }
      }
    }
  }
}

}  // namespace siu3r_epi_pp
