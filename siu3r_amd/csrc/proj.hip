// Tall-and-skinny projections at the end of the DPT heads, bf16x3: out[M, N] = x[M, K] W^T + b with M = 2 x 512^2 rows per pair, K = 256 or 128
// and N = 83 (Gaussian head: head.4, reference src/models/heads/dpt_block.py:384-391 "gs_params" head, last Conv2d(feature_dim, num_channels, 1))
// or N <= 32 (pts3d head: dpt_block.py:357-369 "regression" head, last Conv2d(last_dim, num_channels, 1)).
//
// These are HBM streams (1 KiB read and 332 bytes written per row; 710 MB per pair for the Gaussian head), not GEMM work: on the
// implicit-GEMM kernels the 83 ragged columns cost a second column tile and an unaligned row pass (262 us, 2.7 TB/s: tools/mb_head4.py).
// Here a workgroup streams 32 RG rows per step: the rows arrive as 16-byte pieces in registers (issued one tile ahead, under the MFMAs of
// the current one), are split ONCE into hi / lo bf16 planes when they are stored to LDS (padded rows: conflict-free fragment reads),
// every wave owns one 32-column block of W -- its fragments (K / 16 steps x hi / lo) stay in registers for the whole kernel -- and 32
// rows, reads its A fragments from LDS, runs 3 MFMAs per step, adds the bias and stores its 32 x 32 block (a lane = one column: 128
// contiguous bytes per row and store instruction).  Same products as siu3r_gemm (A_hi W_hi + A_lo W_hi + A_hi W_lo), another K order.
#include "common.h"

namespace {

struct ProjParams {
  const float* x;      // [Z, M, K] fp32, rows contiguous
  const uint4* wfrag;  // [G][NB][K / 16][2 planes][64 lanes] x 8 bf16 (ops.pack_proj)
  const float* bias;   // [G, NB * 32] (zero padded) or null
  float* out;          // [Z, M, N] fp32, row stride ldc
  int64_t M, out_zs;  // out_zs: floats between the outputs of consecutive z
  int Z, G, N, ldc;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// K: inner dimension; NB: 32-column blocks of W; RG: 32-row groups per tile.  Workgroup = NB * RG waves.
template <int K, int NB, int RG>
__global__ __launch_bounds__(64 * NB * RG) void proj_rows_x3_kernel(const ProjParams p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int NT = 64 * NB * RG, ROWS = 32 * RG, KS = K / 16;
  constexpr int LD = K + 8;                       // bf16 elements per staged plane row (16-byte aligned, spreads the banks)
  constexpr int PIECES = ROWS * K / 4;            // 16-byte fp32 pieces per tile
  constexpr int PPT = (PIECES + NT - 1) / NT;     // pieces per thread (the last round may be ragged)
  __shared__ __attribute__((aligned(16))) unsigned short s_hi[ROWS * LD], s_lo[ROWS * LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nb = wave % NB, rg = wave / NB;
  const int z = blockIdx.y, g = z % p.G;
  const int l31 = lane & 31, lh = lane >> 5;

  bf16x8 bh[KS], bl[KS];
  {
    const uint4* wf = p.wfrag + ((size_t)(g * NB + nb) * KS * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bh[s] = as_bf16x8(wf[(2 * s) * 64]);
      bl[s] = as_bf16x8(wf[(2 * s + 1) * 64]);
    }
  }
  const int col = nb * 32 + l31;
  const float bv = (p.bias && col < p.N) ? p.bias[g * NB * 32 + col] : 0.f;

  const int64_t ntiles = (p.M + ROWS - 1) / ROWS;
  const float* xz = p.x + (int64_t)z * p.M * K;
  float* oz = p.out + (int64_t)z * p.out_zs;
  float4 nx[PPT];
  auto fetch = [&](int64_t tile) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = i * NT + t;                       // piece of the tile: row pc / (K / 4), 4 columns from 4 (pc % (K / 4))
      const int64_t row = tile * ROWS + pc / (K / 4);
      nx[i] = (pc < PIECES && row < p.M) ? *(const float4*)(xz + row * K + 4 * (pc % (K / 4))) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    lds_barrier();  // every wave has read its fragments of the previous tile
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pc = i * NT + t, r = pc / (K / 4), c = 4 * (pc % (K / 4));
      if (pc >= PIECES) break;
      const float4 v = nx[i];
      const uint32_t u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
      uint2 h, l;
      h.x = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
      h.y = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
      l.x = pack_bf16x2(v.x - __uint_as_float(u0 & 0xffff0000u), v.y - __uint_as_float(u1 & 0xffff0000u));
      l.y = pack_bf16x2(v.z - __uint_as_float(u2 & 0xffff0000u), v.w - __uint_as_float(u3 & 0xffff0000u));
      *(uint2*)(s_hi + r * LD + c) = h;
      *(uint2*)(s_lo + r * LD + c) = l;
    }
    lds_barrier();
    if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);  // the next tile's rows travel during the MFMAs

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned short *ah_p = s_hi + (rg * 32 + l31) * LD + 8 * lh, *al_p = s_lo + (rg * 32 + l31) * LD + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 ah = as_bf16x8(*(const uint4*)(ah_p + 16 * s)), al = as_bf16x8(*(const uint4*)(al_p + 16 * s));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s], acc, 0, 0, 0);
    }
    if (col < p.N) {
      const int64_t row0 = tile * ROWS + rg * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) oz[row * p.ldc + col] = acc[r] + bv;
      }
    }
  }
#endif
}

}  // namespace

extern "C" int siu3r_proj_rows_x3(const float* x, const void* wfrag, const float* bias, float* out, int Z, int G, int64_t M, int K, int N, int ldc,
                                  int64_t out_z_stride, void* stream) {
  SIU3R_CHECK(x && wfrag && out, "proj_rows_x3: null pointer");
  SIU3R_CHECK(Z > 0 && G > 0 && Z % G == 0 && M > 0 && N > 0 && ldc >= N, "proj_rows_x3: bad shape (Z=%d G=%d N=%d ldc=%d)", Z, G, N, ldc);
  ProjParams p{x, (const uint4*)wfrag, bias, out, M, out_z_stride > 0 ? out_z_stride : M * ldc, Z, G, N, ldc};
  const int nb = (N + 31) / 32;
  // one workgroup per CU (its W fragments live in registers), the Z problems share the chip
  auto grid = [&](int rows) {
    const int64_t nt = (M + rows - 1) / rows;
    int64_t per = 256 / Z;
    if (per < 1) per = 1;
    return dim3((unsigned)(nt < per ? nt : per), Z);
  };
  hipStream_t s = (hipStream_t)stream;
  if (K == 256 && nb == 3)
    hipLaunchKernelGGL((proj_rows_x3_kernel<256, 3, 2>), grid(64), dim3(384), 0, s, p);
  else if (K == 128 && nb == 1)
    hipLaunchKernelGGL((proj_rows_x3_kernel<128, 1, 4>), grid(128), dim3(256), 0, s, p);
  else {
    siu3r_set_error("proj_rows_x3: no instantiation for K=%d N=%d (K 256 with N 65..96, K 128 with N <= 32)", K, N);
    return 1;
  }
  SIU3R_LAUNCH_CHECK("siu3r_proj_rows_x3");
  return 0;
}
