// Shared GEMM epilogue for gemm.hip / gemm_dma.hip.
//
// The MFMA accumulators (lane = one output column, 16 rows per 32x32 tile) are staged through LDS in two
// 64-row halves and written back row-wise: every lane owns 8 consecutive columns of one row, so bias / residual /
// output move as 16-byte (bf16) or 2x16-byte (fp32) vectors and a row of the tile is one contiguous burst,
// instead of 2-4 byte scalar stores per lane (which made the old epilogue cost more than a K=1024 main loop).
// Fused here: bias, exact-erf GELU / ReLU, residual add, RoPE2D on q|k columns, conv-transpose pixel shuffle,
// bilinear x2 upsample-add (GS head), dtype conversion.  Ragged edges and unaligned rows fall back to scalars.
#pragma once
#include "common.h"

namespace siu3r_epi {

// erf with |error| <= 1.5e-7 (Abramowitz-Stegun 7.1.26): fp32-epsilon class, an order cheaper than erff()
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = 1.0f / (1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float y = 1.0f - poly * __expf(-ax * ax);
  return x < 0.f ? -y : y;
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

template <int NI>
__device__ __forceinline__ void run(const siu3r_gemm_params& p, f32x16 (&acc)[2][NI], unsigned char* smem, int tile_m,
                                    int tile_n, int z, int t) {
  constexpr int BN = 64 * NI, BM = 128;
  constexpr int LDC = BN + 4;          // floats per staged row (16-B aligned rows, spreads banks)
  constexpr int CHUNKS = BN / 8;       // 8-column chunks per row
  constexpr int TASKS = 64 * CHUNKS;   // per half
  float* cs = (float*)smem;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  unsigned char* Cb = (unsigned char*)p.c;
  const unsigned char* Rb = (const unsigned char*)p.residual;
  const int64_t c_boff = (int64_t)z * p.sc, r_boff = (int64_t)z * p.sr;
  const int c_esz = p.c_dtype == SIU3R_F32 ? 4 : 2;

  for (int half = 0; half < 2; ++half) {
    __syncthreads();  // main loop (or previous half) no longer reads the staging area
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = wn * (32 * NI) + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        cs[lr * LDC + col] = half == 0 ? acc[0][j][r] : acc[1][j][r];
      }
    }
    __syncthreads();
    for (int task = t; task < TASKS; task += 256) {
      const int chunk = task % CHUNKS, lr = task / CHUNKS;
      const int n0 = tile_n * BN + chunk * 8;
      const int m = tile_m * BM + (lr >> 5) * 64 + half * 32 + (lr & 31);
      if (n0 >= p.n || m >= p.m) continue;
      const int nv = min(8, p.n - n0);
      float v[8];
      {
        const float4 a = *(const float4*)(cs + lr * LDC + chunk * 8);
        const float4 b = *(const float4*)(cs + lr * LDC + chunk * 8 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
      // ---- bias (indexed by output channel; conv-transpose: n = (ky*up+kx)*cout + co)
      int co0 = n0, kidx = 0;
      if (p.out_mode == 1) {
        kidx = n0 / p.cout;
        co0 = n0 - kidx * p.cout;
      }
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nv) v[e] += p.bias[co0 + e];
      }
      // ---- RoPE2D on q|k columns: partner chunk is 16 columns away (chunk ^ 2 inside the 64-wide head)
      if (p.rope_cos != nullptr && n0 < p.rope_ncols) {
        const int pc = chunk ^ 2;
        const float4 a = *(const float4*)(cs + lr * LDC + pc * 8);
        const float4 b = *(const float4*)(cs + lr * LDC + pc * 8 + 4);
        float pv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const int pn0 = tile_n * BN + pc * 8;
        const int d0 = n0 & 63, axis = d0 >> 5;
        const bool upper = (d0 & 16) != 0;
        const int64_t pos = p.rope_pos[((int64_t)z * p.m + m) * 2 + axis];
        const float* cs_ = p.rope_cos + pos * 16 + (d0 & 8);
        const float* sn_ = p.rope_sin + pos * 16 + (d0 & 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float pb = p.bias ? p.bias[pn0 + e] : 0.f;
          const float c = cs_[e], s = sn_[e], o = pv[e] + pb;
          v[e] = upper ? (v[e] * c + o * s) : (v[e] * c - o * s);
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
      const bool full = nv == 8 && (p.out_mode == 0 || co0 + 8 <= p.cout);
      // ---- fused bilinear x2 (align_corners=True) upsample-add of a low-res NHWC map with n channels
      if (p.up_src) {
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
        const int64_t i00 = (sb + (int64_t)y0 * sw + x0) * p.n + n0, i01 = (sb + (int64_t)y0 * sw + x1) * p.n + n0;
        const int64_t i10 = (sb + (int64_t)y1 * sw + x0) * p.n + n0, i11 = (sb + (int64_t)y1 * sw + x1) * p.n + n0;
        if (full && (p.n & 7) == 0) {
          const f32x8 a = load8_as_f32(p.up_src, p.up_dtype, i00), b_ = load8_as_f32(p.up_src, p.up_dtype, i01);
          const f32x8 c = load8_as_f32(p.up_src, p.up_dtype, i10), d = load8_as_f32(p.up_src, p.up_dtype, i11);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] += (1.f - ly) * ((1.f - lx) * a.v[e] + lx * b_.v[e]) + ly * ((1.f - lx) * c.v[e] + lx * d.v[e]);
        } else {
          for (int e = 0; e < nv; ++e)
            v[e] += (1.f - ly) * ((1.f - lx) * load_as_f32(p.up_src, p.up_dtype, i00 + e) + lx * load_as_f32(p.up_src, p.up_dtype, i01 + e)) +
                    ly * ((1.f - lx) * load_as_f32(p.up_src, p.up_dtype, i10 + e) + lx * load_as_f32(p.up_src, p.up_dtype, i11 + e));
        }
      }
      // ---- residual
      if (Rb) {
        const int64_t ridx = r_boff + ((p.out_mode == 0) ? (int64_t)m * p.ldr + n0 : oidx);
        const int r_esz = p.r_dtype == SIU3R_F32 ? 4 : 2;
        if (full && (((uintptr_t)Rb + ridx * r_esz) & 15) == 0) {
          const f32x8 r = load8_as_f32(Rb, p.r_dtype, ridx);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r.v[e];
        } else {
          for (int e = 0; e < nv; ++e) v[e] += load_as_f32(Rb, p.r_dtype, ridx + e);
        }
      }
      // ---- store
      const int64_t cidx = c_boff + oidx;
      if (full && (((uintptr_t)Cb + cidx * c_esz) & 15) == 0) {
        f32x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = v[e];
        store8_from_f32(Cb, p.c_dtype, cidx, o);
      } else {
        for (int e = 0; e < nv; ++e) store_from_f32(Cb, p.c_dtype, cidx + e, v[e]);  // cout % 8 == 0: no straddling
      }
    }
  }
}

}  // namespace siu3r_epi
