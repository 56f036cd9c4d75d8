// MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// C[M,N] = epilogue( A[M,K] x W[N,K]^T ), bf16 MFMA operands, fp32 accumulate.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 tiles.
// Operands are register-staged (global -> VGPR -> LDS) with the next K-tile's loads issued before
// the current tile's MFMAs and a double-buffered, XOR-swizzled LDS image (one barrier per K-step).
// bf16x3 mode (SPLIT): A is fp32, split on the fly into hi+lo bf16; W carries a pre-split lo plane;
// three MFMAs per product recover ~fp32 accuracy (fp32 residual of bf16 rounding is 2^-17 relative).
//
// A addressing modes: dense rows, NHWC implicit conv gather (any KHxKW/stride/pad, Cin % 8 == 0),
// NCHW fp32 16x16 patchify (coalesced patch-embed im2col, reference croco/patch_embed.py:19-29).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand plane per stage

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // row stride 128 B, 16-B chunks XOR-swizzled so that ds_read_b128 lane groups hit 16 distinct slots
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int A_F32, int SPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(const siu3r_gemm_params p) {
  constexpr int PLANES = SPLIT ? 2 : 1;
  // stage s: [A_hi | B_hi | (A_lo | B_lo)]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * PLANES * TILE_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int z = blockIdx.z;

  const int esz_a = A_F32 ? 4 : 2;
  const unsigned char* Ab = (const unsigned char*)p.a + (int64_t)z * p.sa * esz_a;
  const u16* Wh = (const u16*)p.w_hi + (int64_t)z * p.sw;
  const u16* Wl = SPLIT ? (const u16*)p.w_lo + (int64_t)z * p.sw : nullptr;

  // ---- per-thread load geometry: chunk (8 k-elements) x 4 rows
  const int chunk = t & 7, row0 = t >> 3;
  int64_t a_base[4];
  int a_iy0[4], a_ix0[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = tile_m * BM + row0 + 32 * i;
    a_ok[i] = m < p.m;
    a_iy0[i] = a_ix0[i] = 0;
    a_base[i] = 0;
    if (!a_ok[i]) continue;
    if (p.a_mode == 0) {
      a_base[i] = (int64_t)m * p.lda;
    } else {
      int ohw = p.oh * p.ow;
      int b = m / ohw, r = m - b * ohw;
      int oy = r / p.ow, ox = r - oy * p.ow;
      if (p.a_mode == 1) {
        a_base[i] = (int64_t)b * p.ih * p.iw * p.cin;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
      } else {  // patchify NCHW: 3 channels, 16x16 patches
        a_base[i] = (int64_t)b * 3 * p.ih * p.iw + (int64_t)(oy * 16) * p.iw + ox * 16;
      }
    }
  }
  int64_t w_base[4];
  bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = tile_n * BN + row0 + 32 * i;
    w_ok[i] = n < p.n;
    w_base[i] = (int64_t)n * p.kpad + chunk * 8;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // register staging
  float4 ra0[4], ra1[4];  // fp32 A: 8 floats per row-chunk
  uint4 rab[4];           // bf16 A
  uint4 rwh[4], rwl[4];

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK + chunk * 8;
    const bool k_ok = k0 < p.k;
    int ky = 0, kx = 0, c0 = 0;
    int64_t koff = 0;
    if (p.a_mode == 1) {
      int tap = k0 / p.cin;
      c0 = k0 - tap * p.cin;
      ky = tap / p.kw;
      kx = tap - ky * p.kw;
    } else if (p.a_mode == 2) {
      int c = k0 >> 8, kyy = (k0 >> 4) & 15, kxx = k0 & 15;
      koff = (int64_t)c * p.ih * p.iw + (int64_t)kyy * p.iw + kxx;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = a_ok[i] && k_ok;
      int64_t off = 0;
      if (p.a_mode == 0) {
        off = a_base[i] + k0;
      } else if (p.a_mode == 1) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        ok = ok && iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw;
        off = a_base[i] + ((int64_t)iy * p.iw + ix) * p.cin + c0;
      } else {
        off = a_base[i] + koff;
      }
      if (A_F32) {
        if (ok) {
          const float4* q = (const float4*)(Ab + off * 4);
          ra0[i] = q[0];
          ra1[i] = q[1];
        } else {
          ra0[i] = make_float4(0, 0, 0, 0);
          ra1[i] = make_float4(0, 0, 0, 0);
        }
      } else {
        rab[i] = ok ? *(const uint4*)(Ab + off * 2) : make_uint4(0, 0, 0, 0);
      }
      if (w_ok[i]) {
        rwh[i] = *(const uint4*)(Wh + w_base[i] + (int64_t)kt * BK);
        if (SPLIT) rwl[i] = *(const uint4*)(Wl + w_base[i] + (int64_t)kt * BK);
      } else {
        rwh[i] = make_uint4(0, 0, 0, 0);
        if (SPLIT) rwl[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };

  auto store_tile = [&](int stage) {
    unsigned char* sA = smem + stage * (2 * PLANES * TILE_BYTES);
    unsigned char* sB = sA + TILE_BYTES;
    unsigned char* sAl = sA + 2 * TILE_BYTES;
    unsigned char* sBl = sA + 3 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row0 + 32 * i;
      const int o = lds_off(r, chunk);
      if (A_F32) {
        float f[8] = {ra0[i].x, ra0[i].y, ra0[i].z, ra0[i].w, ra1[i].x, ra1[i].y, ra1[i].z, ra1[i].w};
        if (p.relu_in) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        if (SPLIT) {
          uint4 hi, lo;
          split_bf16x8(f, hi, lo);
          *(uint4*)(sA + o) = hi;
          *(uint4*)(sAl + o) = lo;
        } else {
          *(uint4*)(sA + o) = pack_bf16x8(f);
        }
      } else {
        uint4 v = rab[i];
        if (p.relu_in) {
          uint32_t* w = (uint32_t*)&v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t x = w[j];
            if (x & 0x8000u) x &= 0xffff0000u;
            if (x & 0x80000000u) x &= 0x0000ffffu;
            w[j] = x;
          }
        }
        *(uint4*)(sA + o) = v;
      }
      *(uint4*)(sB + o) = rwh[i];
      if (SPLIT) *(uint4*)(sBl + o) = rwl[i];
    }
  };

  const int nkt = p.kpad / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const unsigned char* sA = smem + cur * (2 * PLANES * TILE_BYTES);
    const unsigned char* sB = sA + TILE_BYTES;
    const unsigned char* sAl = sA + 2 * TILE_BYTES;
    const unsigned char* sBl = sA + 3 * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + lh;
      bf16x8 fa[2], fb[2], fal[2], fbl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        const int rb = wn * 64 + i * 32 + l31;
        fa[i] = as_bf16x8(*(const uint4*)(sA + lds_off(ra, c)));
        fb[i] = as_bf16x8(*(const uint4*)(sB + lds_off(rb, c)));
        if (SPLIT) {
          fal[i] = as_bf16x8(*(const uint4*)(sAl + lds_off(ra, c)));
          fbl[i] = as_bf16x8(*(const uint4*)(sBl + lds_off(rb, c)));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fb[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fbl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nkt) {
      store_tile(cur ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue
  unsigned char* Cb = (unsigned char*)p.c;
  const unsigned char* Rb = (const unsigned char*)p.residual;
  const int64_t c_boff = (int64_t)z * p.sc, r_boff = (int64_t)z * p.sr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = tile_n * BN + wn * 64 + j * 32 + l31;
    if (n >= p.n) continue;
    int co = n, kidx = 0;
    if (p.out_mode == 1) {
      kidx = n / p.cout;
      co = n - kidx * p.cout;
    }
    const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = tile_m * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.m) continue;
        float v = acc[i][j][r] + bv;
        if (p.act == 1)
          v = gelu_erf(v);
        else if (p.act == 2)
          v = fmaxf(v, 0.f);
        int64_t oidx;
        if (p.out_mode == 0) {
          oidx = (int64_t)m * p.ldc + n;
        } else {
          const int ihw = p.ih * p.iw;
          const int b = m / ihw, rr = m - b * ihw;
          const int iy = rr / p.iw, ix = rr - iy * p.iw;
          const int ky = kidx / p.up, kx = kidx - ky * p.up;
          oidx = (((int64_t)b * (p.ih * p.up) + iy * p.up + ky) * (p.iw * p.up) + ix * p.up + kx) * p.cout + co;
        }
        if (p.up_src) {
          // + bilinear x2 (align_corners=True) sample of a low-res NHWC map with n channels
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
          const float v00 = load_as_f32(p.up_src, p.up_dtype, (sb + (int64_t)y0 * sw + x0) * p.n + n);
          const float v01 = load_as_f32(p.up_src, p.up_dtype, (sb + (int64_t)y0 * sw + x1) * p.n + n);
          const float v10 = load_as_f32(p.up_src, p.up_dtype, (sb + (int64_t)y1 * sw + x0) * p.n + n);
          const float v11 = load_as_f32(p.up_src, p.up_dtype, (sb + (int64_t)y1 * sw + x1) * p.n + n);
          v += (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        if (Rb) {
          const int64_t ridx = (p.out_mode == 0) ? (int64_t)m * p.ldr + n : oidx;
          v += load_as_f32(Rb, p.r_dtype, r_boff + ridx);
        }
        store_from_f32(Cb, p.c_dtype, c_boff + oidx, v);
      }
    }
  }
}

}  // namespace

extern "C" int siu3r_gemm(const siu3r_gemm_params* pp, void* stream) {
  const siu3r_gemm_params& p = *pp;
  SIU3R_CHECK(p.a && p.w_hi && p.c, "siu3r_gemm: null operand pointer");
  SIU3R_CHECK(p.m > 0 && p.n > 0 && p.k > 0, "siu3r_gemm: empty problem (m=%d n=%d k=%d)", p.m, p.n, p.k);
  SIU3R_CHECK(p.kpad % BK == 0 && p.kpad >= p.k, "siu3r_gemm: kpad=%d must be a multiple of 64 and >= k=%d", p.kpad, p.k);
  SIU3R_CHECK(p.a_dtype == SIU3R_BF16 || p.a_dtype == SIU3R_F32, "siu3r_gemm: bad a_dtype %d", p.a_dtype);
  SIU3R_CHECK(p.c_dtype == SIU3R_BF16 || p.c_dtype == SIU3R_F32, "siu3r_gemm: bad c_dtype %d", p.c_dtype);
  SIU3R_CHECK(!(p.w_lo && p.a_dtype != SIU3R_F32), "siu3r_gemm: bf16x3 mode needs fp32 activations");
  SIU3R_CHECK(p.a_mode >= 0 && p.a_mode <= 2, "siu3r_gemm: bad a_mode %d", p.a_mode);
  if (p.a_mode == 0) {
    SIU3R_CHECK(p.k % 8 == 0 && p.lda % 8 == 0, "siu3r_gemm: dense A needs k %% 8 == 0 and lda %% 8 == 0 (k=%d lda=%ld)", p.k, (long)p.lda);
  } else if (p.a_mode == 1) {
    SIU3R_CHECK(p.cin % 8 == 0, "siu3r_gemm: conv gather needs cin %% 8 == 0 (cin=%d)", p.cin);
    SIU3R_CHECK(p.k == p.kh * p.kw * p.cin, "siu3r_gemm: conv k=%d != kh*kw*cin", p.k);
    SIU3R_CHECK(p.m % (p.oh * p.ow) == 0, "siu3r_gemm: conv m=%d not a multiple of oh*ow", p.m);
  } else {
    SIU3R_CHECK(p.a_dtype == SIU3R_F32 && p.k == 768 && p.ih == p.oh * 16 && p.iw == p.ow * 16,
                "siu3r_gemm: patchify mode needs fp32 NCHW input, k=768, H,W multiples of 16");
  }
  if (p.out_mode == 1) SIU3R_CHECK(p.n == p.up * p.up * p.cout && p.m % (p.ih * p.iw) == 0, "siu3r_gemm: bad conv-transpose geometry");
  if (p.up_src) SIU3R_CHECK(p.a_mode == 1 && p.out_mode == 0 && p.oh % 2 == 0 && p.ow % 2 == 0, "siu3r_gemm: up_src needs conv mode with even output size");
  const int tiles = ((p.m + BM - 1) / BM) * ((p.n + BN - 1) / BN);
  dim3 grid(tiles, 1, p.batch > 0 ? p.batch : 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (p.w_lo) {
    hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, block, 0, s, p);
  } else if (p.a_dtype == SIU3R_F32) {
    hipLaunchKernelGGL((gemm_kernel<1, 0>), grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<0, 0>), grid, block, 0, s, p);
  }
  SIU3R_LAUNCH_CHECK("siu3r_gemm");
  return 0;
}
