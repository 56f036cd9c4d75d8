// Shared GEMM epilogue for gemm.hip / gemm_dma.hip.
//
// The MFMA accumulators (lane = one output column, 16 rows per 32x32 tile) are staged through LDS (whole tile)
// and written back row-wise: every lane owns 8 consecutive columns of one row, so bias / residual /
// output move as 16-byte (bf16) or 2x16-byte (fp32) vectors and a row of the tile is one contiguous burst,
// instead of 2-4 byte scalar stores per lane (which made the old epilogue cost more than a K=1024 main loop).
// Fused here: bias, exact-erf GELU / ReLU, residual add, RoPE2D on q|k columns, conv-transpose pixel shuffle,
// bilinear x2 upsample-add (GS head), dtype conversion.  Ragged edges and unaligned rows fall back to scalars.
#pragma once
#include "common.h"

namespace siu3r_epi {

// GELU(x) = x Phi(x) = 0.5 (x + |x| erf(|x| / sqrt 2)) with erf(u) = 1 - (1 + a1 u + ... + a6 u^6)^-16
// (Abramowitz-Stegun 7.1.28, |error| <= 3e-7: fp32-epsilon class).  One quarter-rate op (the reciprocal) per element
// instead of erff()'s or 7.1.26's two (reciprocal + exponential); the 1/sqrt 2 is folded into the coefficients.
// (1 + ...)^16 overflows to +inf beyond |x| ~ 25, where 1/inf = 0 gives erf = 1 exactly.
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x);
  float q = 4.30638e-5f * 0.125f;                        // a6 / 2^3
  q = q * ax + 2.765672e-4f * 0.17677669529663687f;      // a5 / 2^2.5
  q = q * ax + 1.520143e-4f * 0.25f;                     // a4 / 2^2
  q = q * ax + 9.2705272e-3f * 0.35355339059327373f;     // a3 / 2^1.5
  q = q * ax + 4.22820123e-2f * 0.5f;                    // a2 / 2
  q = q * ax + 7.05230784e-2f * 0.70710678118654752f;    // a1 / 2^0.5
  q = q * ax + 1.0f;
  q *= q;
  q *= q;
  q *= q;
  q *= q;
  const float e = 1.0f - __builtin_amdgcn_rcpf(q);
  return 0.5f * (ax * e + x);
}

template <int NI, int KSPL = 1>
constexpr int staging_bytes() { return 128 * (64 * NI + 4) * 4 + 2048; }  // tile + per-row (mean, rstd) and per-column c1 of a folded LayerNorm

// The whole 128 x BN accumulator tile is staged once (two raw barriers); every thread then owns ONE 8-column chunk
// (t % CHUNKS, so its bias / RoPE axis are loop invariants) of NTASK rows.  All global reads of a group of 4 rows --
// residual, RoPE positions, then the cos/sin rows -- are issued before any of them is used, and the first group's are
// issued before the staging barriers, so the tile pays one memory latency instead of one per row (the per-row
// dependent loads of the previous version cost as much as a K = 1024 main loop).
// KSPL = 2: the workgroup has 8 waves; waves 4..7 hold the accumulators of the second half of every K tile for the same
// sub-tiles as waves 0..3.  They hand their partial sums to waves 0..3 through the staging area first (same lane, same
// register <-> same address), then the tile is staged once and all 512 threads run the row pass.
// LNF (compile time): this launch carries a folded LayerNorm (siu3r_gemm_params.ln_*).  A separate instantiation, so that the plain
// epilogue keeps its register budget (110 VGPRs = two 8-wave workgroups per CU for the 128 x 64 kernel).
template <int NI, int KSPL = 1, bool LNF = false>
__device__ __forceinline__ void run(const siu3r_gemm_params& p, f32x16 (&acc)[2][NI], unsigned char* smem, int tile_m,
                                    int tile_n, int z, int t) {
  constexpr int BN = 64 * NI, BM = 128;
  constexpr int LDC = BN + 4;            // floats per staged row (16-B aligned rows, spreads banks)
  constexpr int CHUNKS = BN / 8;         // 8-column chunks per row
  constexpr int RPP = 256 * KSPL / CHUNKS;  // rows per pass of the workgroup's threads
  constexpr int NTASK = BM / RPP;        // rows per thread
  constexpr int G = NTASK < 4 ? NTASK : 4, NG = NTASK / G;
  float* cs = (float*)smem;
  const int lane = t & 63, wave = (t >> 6) & 3, kh = t >> 8;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  unsigned char* Cb = (unsigned char*)p.c;
  const unsigned char* Rb = (const unsigned char*)p.residual;
  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const int64_t c_boff = zof.c, r_boff = zof.r;
  const float* biasp = p.bias ? p.bias + zof.bias : nullptr;
  // ---- LayerNorm folded into this GEMM: rstd_m * (acc - mean_m * c1[n]) + c2[n]
  constexpr bool ln = LNF;
  const float* c1p = ln ? p.ln_c1 + zof.bias : nullptr;
  const float* c2p = ln ? p.ln_c2 + zof.bias : nullptr;
  float* s_ln = cs + BM * LDC;   // [BM][2] (mean, rstd) behind the staged tile
  float* s_c1 = s_ln + 2 * BM;   // [BN] c1 of the tile's columns
  const int c_esz = p.c_dtype == SIU3R_F32 ? 4 : 2;
  const int r_esz = p.r_dtype == SIU3R_F32 ? 4 : 2;
  const int M = p.m, N = p.n;

  const int chunk = t % CHUNKS, row0 = t / CHUNKS;
  const int n0 = tile_n * BN + chunk * 8;
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

  // ---- loop invariants: bias of this chunk (and of the RoPE partner chunk)
  float bias[8], pbias[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias[e] = pbias[e] = 0.f;
  const float* addp = ln ? c2p : biasp;  // the additive per-column term: bias, or the folded LayerNorm's c2
  if (addp && col_ok) {
    if (nv == 8 && (((uintptr_t)(addp + co0)) & 15) == 0) {
      const float4 a = *(const float4*)(addp + co0), b = *(const float4*)(addp + co0 + 4);
      bias[0] = a.x; bias[1] = a.y; bias[2] = a.z; bias[3] = a.w; bias[4] = b.x; bias[5] = b.y; bias[6] = b.z; bias[7] = b.w;
    } else {
      _Pragma("unroll") for (int e = 0; e < 8; ++e)
        if (e < nv) bias[e] = addp[co0 + e];
    }
    if (rope) {
      const int pn0 = tile_n * BN + pc * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) pbias[e] = addp[pn0 + e];
    }
  }
  // statistics of the tile's rows: TPR threads per row merge the 64-column (mean, M2) partials of the row (Chan's formula; counts are
  // 64 except for the last partial).  The loads go out here, the merge runs behind the first barrier (the ring is then free)
  constexpr int LN_TMAX = 16, TPR = 2 * KSPL, LN_PER = LN_TMAX / TPR;
  float2 lnp[ln ? LN_PER : 1];
  const int ln_r = t / TPR, ln_part = t - ln_r * TPR;
  const bool ln_row = ln && tile_m * BM + ln_r < M;
  if (ln) {
#pragma unroll
    for (int i = 0; i < LN_PER; ++i) lnp[i] = make_float2(0.f, 0.f);
    if (ln_row) {
      const float2* sp = (const float2*)p.ln_stats + ((int64_t)zof.zo * p.ln_sz + (int64_t)zof.zi * p.ln_sz_i + (int64_t)(tile_m * BM + ln_r) * p.ln_ldm) * p.ln_tiles;
#pragma unroll
      for (int i = 0; i < LN_PER; ++i)
        if (ln_part * LN_PER + i < p.ln_tiles) lnp[i] = sp[ln_part * LN_PER + i];
    }
  }
  f32x8 res[G];
  int64_t pos[G];
  auto issue_loads = [&](int g) {
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int lr = row0 + (g * G + k) * RPP;
      const int m = tile_m * BM + lr;
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
      if (rope) pos[k] = p.rope_pos[((int64_t)z * M + m) * 2 + axis];
    }
  };

  issue_loads(0);
  // ---- stage the tile.  Raw barriers: the loads above stay in flight across them
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  lds_barrier();  // main loop no longer reads the ring
  if (ln) {
    const int Cn = p.k, last = Cn - 64 * (p.ln_tiles - 1);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_PER; ++i) {
      const int ti = ln_part * LN_PER + i;
      if (ti < p.ln_tiles) sum += lnp[i].x * (float)(ti == p.ln_tiles - 1 ? last : 64);
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
    const float mu = sum / (float)Cn;
    float m2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_PER; ++i) {
      const int ti = ln_part * LN_PER + i;
      if (ti < p.ln_tiles) {
        const float d = lnp[i].x - mu;
        m2 += lnp[i].y + d * d * (float)(ti == p.ln_tiles - 1 ? last : 64);
      }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) m2 += __shfl_xor(m2, o);
    if (ln_part == 0) {
      s_ln[2 * ln_r] = ln_row ? mu : 0.f;
      s_ln[2 * ln_r + 1] = ln_row ? rsqrtf(m2 / (float)Cn + p.ln_eps) : 0.f;
    }
    if (t < BN) s_c1[t] = (tile_n * BN + t < N) ? c1p[tile_n * BN + t] : 0.f;
  }
  if (KSPL == 2) {
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wn * (32 * NI) + j * 32 + l31] = acc[i][j][r];
    }
    lds_barrier();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[i][j][r] += cs[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDC + wn * (32 * NI) + j * 32 + l31];
    }
    lds_barrier();  // the partials have been read: the area may be overwritten with the totals
  }
  // ---- split-K over workgroups: publish this slice's partial tile; the tile's last arriver sums all slices in slice order.
  // Slabs travel as 16-byte WRITE-THROUGH (sc1) stores and are read back with sc1 loads (L2-served, never a stale L1 line): no
  // release / acquire fence is needed -- a fence would write back / invalidate whole caches under every concurrent kernel's feet
  // (MI355X_MICROARCH.md, hand-off price list: "sc1 stores AND sc1 loads both sides") -- only "my stores have left" (vmcnt(0)) before
  // the relaxed ticket.
#if __HIP_DEVICE_COMPILE__
  if (p.splitk > 1) {
    const int S = p.splitk, tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int64_t tile_id = ((int64_t)z * tiles_m + tile_m) * tiles_n + tile_n;
    float* slabs = p.sk_ws + tile_id * S * (BM * BN);
    int* s_flag = (int*)(s_c1 + 160);
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    constexpr int NV = 2 * NI * 4;  // 16-byte vectors of accumulators per thread
    __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc((void*)slabs, (short)0, (int)(S * BM * BN * 4), 0x00020000);
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            union { u32x4_t u; float f[4]; } v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v.f[e] = acc[i][j][4 * q + e];
            const int vec = (i * NI + j) * 4 + q;
            __builtin_amdgcn_raw_buffer_store_b128(v.u, rs_, (int)((((int)blockIdx.y * NV + vec) * 256 + (t & 255)) * 16), 0, 16 /* sc1 */);
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) *s_flag = __hip_atomic_fetch_add(p.sk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_flag != S - 1) return;  // (uniform) an earlier slice: done
    if (t == 0) __hip_atomic_store(p.sk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      for (int sl = 0; sl < S; ++sl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int vec = (i * NI + j) * 4 + q;
              union { u32x4_t u; float f[4]; } v;
              v.u = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(((sl * NV + vec) * 256 + (t & 255)) * 16), 0, 16 /* sc1 */);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v.f[e];
            }
      }
    }
  }
#endif
  if (kh == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wn * (32 * NI) + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          cs[lr * LDC + col] = acc[i][j][r];
        }
      }
  }
  lds_barrier();
  if (p.trace && t == 0) p.trace[(size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + 5] = __builtin_readcyclecounter();

#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g > 0) issue_loads(g);
    float4 rc[G][2], rs[G][2];
    if (rope) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        const float* cs_ = p.rope_cos + pos[k] * 16 + (d0 & 8);
        const float* sn_ = p.rope_sin + pos[k] * 16 + (d0 & 8);
        rc[k][0] = *(const float4*)cs_; rc[k][1] = *(const float4*)(cs_ + 4);
        rs[k][0] = *(const float4*)sn_; rs[k][1] = *(const float4*)(sn_ + 4);
      }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int lr = row0 + (g * G + k) * RPP;
      const int m = tile_m * BM + lr;
      if (m >= M) continue;                       // (uniform over the 8 / 16 threads of a row)
      if (!col_ok && !p.stats_out) continue;
      float v[8];
      {
        const float4 a = *(const float4*)(cs + lr * LDC + chunk * 8);
        const float4 b = *(const float4*)(cs + lr * LDC + chunk * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      float ln_mu = 0.f, ln_rs = 1.f;
      if (ln) {
        ln_mu = s_ln[2 * lr];
        ln_rs = s_ln[2 * lr + 1];
        const float4 ca = *(const float4*)(s_c1 + chunk * 8), cb = *(const float4*)(s_c1 + chunk * 8 + 4);
        const float c1[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ln_rs * (v[e] - ln_mu * c1[e]) + bias[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias[e];
      }
      // ---- RoPE2D on q|k columns
      if (rope) {
        float4 a = *(const float4*)(cs + lr * LDC + pc * 8);
        float4 b = *(const float4*)(cs + lr * LDC + pc * 8 + 4);
        const float pv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const float cc[8] = {rc[k][0].x, rc[k][0].y, rc[k][0].z, rc[k][0].w, rc[k][1].x, rc[k][1].y, rc[k][1].z, rc[k][1].w};
        const float ss[8] = {rs[k][0].x, rs[k][0].y, rs[k][0].z, rs[k][0].w, rs[k][1].x, rs[k][1].y, rs[k][1].z, rs[k][1].w};
        float pc1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ln) {
          const float4 ca = *(const float4*)(s_c1 + pc * 8), cb = *(const float4*)(s_c1 + pc * 8 + 4);
          pc1[0] = ca.x; pc1[1] = ca.y; pc1[2] = ca.z; pc1[3] = ca.w; pc1[4] = cb.x; pc1[5] = cb.y; pc1[6] = cb.z; pc1[7] = cb.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float o = ln ? ln_rs * (pv[e] - ln_mu * pc1[e]) + pbias[e] : pv[e] + pbias[e];
          v[e] = upper ? (v[e] * cc[e] + o * ss[e]) : (v[e] * cc[e] - o * ss[e]);
        }
      }
      // ---- activation
      if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      // ---- output index of element 0 of the chunk
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
      // ---- fused bilinear x2 (align_corners=True) upsample-add of a low-res NHWC map with n channels
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
      // ---- residual (conv-transpose scatter: read at the scattered position, not prefetched)
      if (Rb) {
        if (p.out_mode == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += res[k].v[e];
        } else {
          _Pragma("unroll") for (int e = 0; e < 8; ++e)
            if (e < nv) v[e] += load_as_f32(Rb, p.r_dtype, r_boff + oidx + e);
        }
      }
      // ---- row statistics of the output for the LayerNorm folded into the NEXT GEMM: (mean, centred sum of squares) of this
      // row's 64-column group, reduced over the group's 8 threads (consecutive lanes), written by the first of them
      if (p.stats_out) {
        const int n64 = min(64, N - (n0 & ~63));   // valid columns of the group (> 0 for the group's first chunk)
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
        if ((chunk & 7) == 0 && n64 > 0) {
          const int64_t row = (int64_t)zof.zo * p.st_sz + (int64_t)zof.zi * p.st_sz_i + (int64_t)m * p.st_ldm;
          ((float2*)p.stats_out)[row * ((N + 63) >> 6) + (n0 >> 6)] = make_float2(mean, s2);
        }
        if (!col_ok) continue;
      }
      // ---- store
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
            store_from_f32(Cb, p.c_dtype, cidx + e, v[e]);  // cout % 8 == 0: no straddling
            if (p.c_aux) store_from_f32(p.c_aux, SIU3R_BF16, cidx + e, v[e]);
          }
      }
    }
  }
}

}  // namespace siu3r_epi
