// Pipelined attention for the ViT blocks (head_dim 64, no mask): the attention of reference croco/blocks.py:94-112 (self) and
// 149-169 (cross) after the projection GEMM has applied RoPE2D, i.e. O = softmax(Q K^T * scale) V on strided [B, N, H, 64] views.
//
// Why a second kernel next to attention.hip's attn_fast_kernel: one pair has 2 x 16 (view, head) problems of 1025 queries x 1025 keys.
// That is 2 wave-jobs (32 queries x one 32-key half of every key tile) per SIMD, and attn_fast_kernel spends them in 544 four-wave
// workgroups (2.1 per CU, a 1-query straggler tile per problem, every 64-query workgroup re-reading and re-splitting all of K / V)
// whose waves run their tile chain (S MFMAs -> softmax -> P V MFMAs) serially.  Here
//   * a workgroup is 8 waves = 128 queries x 2 key halves, so 1024 queries of a problem are exactly 8 workgroups and a pair fills
//     the 256 CUs once, with K / V tiles fetched and split once per 128 queries;
//   * query 1024 (the intrinsics token: Nq = 128 n + 1) does not get a tile of its own: the last workgroup of the problem carries it
//     as an fp32 side path on the K / V rows it is staging anyway (one key per 8-lane group and tile, online softmax per group,
//     merged through LDS at the end);
//   * the chain is software-pipelined inside each wave: the S MFMAs of tile kt+1 are issued between the softmax instructions of tile
//     kt, the P V MFMAs of tile kt between the instructions that split / store tile kt+2 (three K/V stages in LDS, one barrier per tile);
//   * workgroups of one (batch, head) problem are mapped to ONE XCD (linear id -> XCD is id % 8), so that its K / V (512 KiB fp32) is
//     fetched into one L2 instead of eight.
// Fragment layouts (S^T = K Q^T so that a lane owns a query column, "virtual k" ordering of P, V transposed in the LDS read) are those
// of attention.hip.  X3 = bf16x3 mode (fp32 tensors, hi/lo split operands, 3 MFMAs per product); otherwise bf16 tensors.
#include <stdlib.h>

#include "common.h"

// tuning aid (tools/ab_attn.sh): 1 = no MFMAs, 2 = no softmax arithmetic, 4 = no K / V refresh (loads, split, stores), 8 = no barrier
#ifndef SIU3R_AP_DBG
#define SIU3R_AP_DBG 0
#endif

namespace siu3r_attn_pipe {

constexpr int KT = 64, D = 64, QT = 128, NT = 512;
constexpr int NKS = 2, NVS = 3;  // K / V stages (K fragments are read one iteration ahead, see the schedule in the kernel)
constexpr float NEG_BIG = -1.0e30f;
constexpr int RS = D * 2;       // V row stride in LDS
constexpr int PL = KT * D * 2;  // one bf16 plane of a K or V tile

__device__ __forceinline__ int k_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;
typedef __attribute__((ext_vector_type(4))) float f32x4v;

// KVP (bf16x3): K and V arrive PRE-SPLIT -- hi | lo bf16 planes per 32 columns, written by the projection GEMM's epilogue
// (siu3r_gemm_params.c_x3 / c_x3_col0) -- and go from global memory to the LDS stages without conversion.  The in-kernel split of fp32
// K / V uses the same definition (hi = upper 16 bits, lo = bf16(x - hi)), so both forms give the same bits.
template <bool X3, bool KVP = false>
__global__ __launch_bounds__(NT) void attn_pipe_kernel(const siu3r_attn_params p, const int q_tiles, const int has_x) {
  static_assert(!KVP || X3, "pre-split K / V: bf16x3");
  constexpr int TST = (X3 ? 2 : 1) * PL;    // one staged tile of K or V (X3: [hi | lo])
  constexpr int KLO = PL, VLO = PL;
  constexpr int VBASE = NKS * TST;          // LDS: NKS K stages, then NVS V stages
  constexpr int NR = X3 ? 2 : 1;            // 16-byte registers per thread, tile and tensor
  constexpr int KS = D / 16, DT = D / 32;
  constexpr int NM = X3 ? 3 : 1;            // MFMAs per product
  constexpr int MERGE_BYTES = (4 * 32 + 64) * (D + 2) * 4;  // end-of-kernel merge areas (key halves, side-path groups)
  constexpr int RING_BYTES = (NKS + NVS) * TST;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RING_BYTES > MERGE_BYTES ? RING_BYTES : MERGE_BYTES];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int qg = wave >> 1, kb = wave & 1;  // query group (32 queries), key half of every tile

  // linear workgroup id -> (problem, query tile): all query tiles of a (batch, head) problem on one XCD when the problems divide by 8
  int bh, q_tile;
  {
    const int L = blockIdx.x, nprob = p.B * p.H;
    if ((nprob & 7) == 0) {
      const int xcd = L & 7, idx = L >> 3;
      bh = xcd + 8 * (idx / q_tiles);
      q_tile = idx - (idx / q_tiles) * q_tiles;
    } else {
      bh = L / q_tiles;
      q_tile = L - bh * q_tiles;
    }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int nq_main = has_x ? p.Nq - 1 : p.Nq;       // queries that own MFMA columns
  const bool do_x = has_x && q_tile == q_tiles - 1;  // this workgroup also carries query Nq - 1 (uniform)
  const int q_row = q_tile * QT + qg * 32 + l31;
  const bool q_ok = q_row < nq_main;
  const float sl2 = p.scale * 1.4426950408889634f;

  // ---- Q fragments (B operand: lane = query, 8 consecutive d per k-substep)
  bf16x8 qf[KS], qfl[X3 ? KS : 1];
  if constexpr (X3) {
    const float* qp = (const float*)p.q + (int64_t)b * p.q_sb + (int64_t)(q_ok ? q_row : nq_main - 1) * p.q_sn + (int64_t)h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 a = *(const float4*)(qp + ks * 16 + 8 * lh), c = *(const float4*)(qp + ks * 16 + 8 * lh + 4);
      const float f[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      uint4 hi, lo;
      split_bf16x8(f, hi, lo);
      qf[ks] = as_bf16x8(hi);
      qfl[ks] = as_bf16x8(lo);
    }
  } else {
    const u16* qp = (const u16*)p.q + (int64_t)b * p.q_sb + (int64_t)(q_ok ? q_row : nq_main - 1) * p.q_sn + (int64_t)h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = as_bf16x8(*(const uint4*)(qp + ks * 16 + 8 * lh));
  }

  // ---- staging: thread -> (key = t >> 3, 8-element chunk t & 7) of every tile
  const int ld_key = t >> 3, ch = t & 7;
  constexpr int ESZ = X3 ? 4 : 2;
  // (planes: the thread's 8 dims are 16 bytes of the hi half of their 32-column segment, the lo half 64 bytes further)
  const int ch_off = KVP ? (ch >> 2) * 128 + (ch & 3) * 16 : ch * (8 * ESZ);
  const unsigned char* kbase = (const unsigned char*)p.k + ((int64_t)(b ^ p.kv_bxor) * p.k_sb + (int64_t)h * p.k_sh) * ESZ + ch_off;
  const unsigned char* vbase = (const unsigned char*)p.v + ((int64_t)(b ^ p.kv_bxor) * p.v_sb + (int64_t)h * p.v_sh) * ESZ + ch_off;
  auto load_rows = [&](const unsigned char* base, int64_t row_stride, int kt, u32x4v (&r)[NR]) {
    int key = kt * KT + ld_key;
    if (key > p.Nk - 1) key = p.Nk - 1;  // clamped: finite values, their scores are masked / never read
    const unsigned char* rp = base + (int64_t)key * row_stride * ESZ;
#pragma unroll
    for (int c = 0; c < NR; ++c) r[c] = *(const u32x4v*)(rp + (KVP ? 64 : 16) * c);
  };
  auto load_k = [&](int kt, u32x4v (&r)[NR]) { load_rows(kbase, p.k_sn, kt, r); };
  auto load_v = [&](int kt, u32x4v (&r)[NR]) { load_rows(vbase, p.v_sn, kt, r); };
  auto chunk_floats = [&](const u32x4v (&r)[NR], float (&f)[8]) {
    if constexpr (KVP) {  // (side path only: the value the MFMA path multiplies, hi + lo)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned wh = r[0][j], wl = r[X3 ? 1 : 0][j];
        f[2 * j] = __uint_as_float(wh << 16) + __uint_as_float(wl << 16);
        f[2 * j + 1] = __uint_as_float(wh & 0xffff0000u) + __uint_as_float(wl & 0xffff0000u);
      }
    } else if constexpr (X3) {
      const f32x4v a = __builtin_bit_cast(f32x4v, r[0]), c = __builtin_bit_cast(f32x4v, r[1]);
      f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = c[0]; f[5] = c[1]; f[6] = c[2]; f[7] = c[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned w = r[0][j];
        f[2 * j] = __builtin_bit_cast(float, w << 16);
        f[2 * j + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
      }
    }
  };
  const int k_st = k_off(ld_key, ch);
  const int v_st = ld_key * RS + ((((ch >> 2) ^ ((ld_key >> 1) & 1)) << 2) | (ch & 3)) * 16;  // 64-byte halves swapped by key bit 1
  auto store_k = [&](int stage, const u32x4v (&r)[NR]) {
    unsigned char* sK = smem + stage * TST;
    if constexpr (KVP) {
      *(u32x4v*)(sK + k_st) = r[0];
      *(u32x4v*)(sK + KLO + k_st) = r[X3 ? 1 : 0];
    } else if constexpr (X3) {
      float f[8];
      chunk_floats(r, f);
      uint4 hi, lo;
      split_trunc_bf16x8(f, hi, lo);
      *(uint4*)(sK + k_st) = hi;
      *(uint4*)(sK + KLO + k_st) = lo;
    } else {
      *(u32x4v*)(sK + k_st) = r[0];
    }
  };
  auto store_v = [&](int stage, const u32x4v (&r)[NR]) {
    unsigned char* sV = smem + VBASE + stage * TST;
    if constexpr (KVP) {
      *(u32x4v*)(sV + v_st) = r[0];
      *(u32x4v*)(sV + VLO + v_st) = r[X3 ? 1 : 0];
    } else if constexpr (X3) {
      float f[8];
      chunk_floats(r, f);
      uint4 hi, lo;
      split_trunc_bf16x8(f, hi, lo);
      *(uint4*)(sV + v_st) = hi;
      *(uint4*)(sV + VLO + v_st) = lo;
    } else {
      *(u32x4v*)(sV + v_st) = r[0];
    }
  };

  // ---- side path of query Nq - 1: this thread's 8 d of it, and the online softmax of its 8-lane group over the keys the group stages
  float qx[8], ox[8], mx = NEG_BIG, lx = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) qx[e] = ox[e] = 0.f;
  if (do_x) {
    const unsigned char* qp = (const unsigned char*)p.q + ((int64_t)b * p.q_sb + (int64_t)(p.Nq - 1) * p.q_sn + (int64_t)h * p.q_sh) * ESZ + ch * (8 * ESZ);
    u32x4v r[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) r[c] = *(const u32x4v*)(qp + 16 * c);
    if constexpr (X3) {  // (q is always fp32)
      const f32x4v a = __builtin_bit_cast(f32x4v, r[0]), c = __builtin_bit_cast(f32x4v, r[X3 ? 1 : 0]);
      qx[0] = a[0]; qx[1] = a[1]; qx[2] = a[2]; qx[3] = a[3]; qx[4] = c[0]; qx[5] = c[1]; qx[6] = c[2]; qx[7] = c[3];
    } else {
      chunk_floats(r, qx);
    }
  }
#define SIU3R_DPP_ADD(x, ctrl) ((x) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), 0xf, 0xf, true)))
  // the score of the group's key of a tile is taken while that tile's K row is in the staging registers; it meets the tile's V row
  // (staged one iteration later) in side_consume
  // (bf16x3, fp32 K / V: the side path multiplies hi + lo of the split as well, so that both operand forms give the same bits)
  auto side_floats = [&](const u32x4v (&r)[NR], float (&f)[8]) {
    chunk_floats(r, f);
    if constexpr (X3 && !KVP) {
      uint4 hi, lo;
      split_trunc_bf16x8(f, hi, lo);
      const unsigned hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(hw[j] << 16) + __uint_as_float(lw[j] << 16);
        f[2 * j + 1] = __uint_as_float(hw[j] & 0xffff0000u) + __uint_as_float(lw[j] & 0xffff0000u);
      }
    }
  };
  auto side_score = [&](int tile, const u32x4v (&rk)[NR]) {
    float kf_[8];
    side_floats(rk, kf_);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = __builtin_fmaf(kf_[e], qx[e], s);
    s = SIU3R_DPP_ADD(s, 0xB1);   // quad_perm [1,0,3,2]
    s = SIU3R_DPP_ADD(s, 0x4E);   // quad_perm [2,3,0,1]
    s = SIU3R_DPP_ADD(s, 0x141);  // row_half_mirror: the other quad of the 8-lane group
    return tile * KT + ld_key >= p.Nk ? NEG_BIG : s;
  };
  auto side_consume = [&](float s, const u32x4v (&rv)[NR]) {
    float vf_[8];
    side_floats(rv, vf_);
    const float m_new = fmaxf(mx, s);
    const float a = __builtin_amdgcn_exp2f((mx - m_new) * sl2), pw = __builtin_amdgcn_exp2f((s - m_new) * sl2);
    mx = m_new;
    lx = lx * a + pw;
#pragma unroll
    for (int e = 0; e < 8; ++e) ox[e] = __builtin_fmaf(ox[e], a, pw * vf_[e]);
  };

  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const int nkt = (p.Nk + KT - 1) / KT;

  // per-lane byte offset of the transposed V reads inside a stage (see attention.hip)
  const int i2 = (lane & 15) >> 2;
  const int v_lane = (4 * lh + i2) * RS + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int v_hswz = (i2 >> 1) & 1;
  const int k_rd = kb * 32 + l31;

  auto other_half = [&](float x) {  // value held by lane ^ 32 (v_permlane32_swap; see attention.hip for the operand aliasing note)
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, lh ? sw[0] : sw[1]);
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(SIU3R_AP_DBG & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto read_k = [&](int stage, bf16x8 (&kf)[KS], bf16x8 (&kfl)[X3 ? KS : 1]) {
    const unsigned char* sK = smem + stage * TST;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = as_bf16x8(*(const uint4*)(sK + k_off(k_rd, ks * 2 + lh)));
      if constexpr (X3) kfl[ks] = as_bf16x8(*(const uint4*)(sK + KLO + k_off(k_rd, ks * 2 + lh)));
    }
  };
  // MFMA i of a score block (X3: lo*hi, hi*lo, hi*hi per k-substep)
  auto s_mfma = [&](int i, f32x16& s, const bf16x8 (&kf)[KS], const bf16x8 (&kfl)[X3 ? KS : 1]) {
    const int ks = i / NM, w = i - ks * NM;
    if (SIU3R_AP_DBG & 1) { s[i & 15] += __builtin_bit_cast(float, (int)kf[ks][0]); return; }
    if (X3 && w == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl[ks], qf[ks], s, 0, 0, 0);
    else if (X3 && w == 1) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfl[X3 ? ks : 0], s, 0, 0, 0);
    else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
  };

  // softmax of the current score block in NV instruction groups (interleaved with the next block's MFMAs by the caller):
  // group 0: running maximum; groups 1..8: two exponentials each; 9, 10: P -> bf16 (hi, lo) of the two 16-key sub-blocks; 11: sums
  float alpha = 1.f, psum = 0.f, nm = 0.f;
  bool grew = false;
  bf16x8 ph[2], plo[X3 ? 2 : 1];
  auto softmax_group = [&](int g, f32x16& s) {
    if ((SIU3R_AP_DBG & 2) && g >= 1 && g <= 8) return;
    if (g == 0) {
      float mxv = fmaxf(s[0], s[1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mxv = fmaxf(fmaxf(mxv, s[r]), s[r + 1]);  // v_max3_f32
      mxv = fmaxf(mxv, other_half(mxv));
      const float m_new = fmaxf(m_run, mxv);
      grew = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0;
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
      m_run = m_new;
      nm = -m_new * sl2;
      psum = 0.f;
    } else if (g <= 8) {
#pragma unroll
      for (int r = 2 * (g - 1); r < 2 * g; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, nm));
        s[r] = pv;
        psum += pv;
      }
    } else if (g <= 10) {
      const int sb = g - 9;
      float pf[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = s[8 * sb + j];
      if constexpr (X3) {
        uint4 hi, lo;
        split_bf16x8(pf, hi, lo);
        ph[sb] = as_bf16x8(hi);
        plo[sb] = as_bf16x8(lo);
      } else {
        ph[sb] = as_bf16x8(pack_bf16x8(pf));
      }
    } else {
      l_run = l_run * alpha + psum;
    }
  };
  constexpr int NV = 12;
  constexpr int NSM = KS * NM;      // MFMAs of a score block (12 / 4)
  constexpr int NPV = 2 * DT * NM;  // MFMAs of a P V block (12 / 4)

  auto rescale_o = [&]() {
    if (grew) {  // wave-uniform: only when some lane's maximum grew
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
  };
  struct VFrag { s16x4 a0, a1, b0, b1; };
  auto read_v = [&](int stage, VFrag (&vf)[2][DT]) {
    const unsigned char* sV = smem + VBASE + stage * TST;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (kb * 32 + 16 * sb) * RS + v_lane + ((dt ^ v_hswz) * 64);
        vf[sb][dt].a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + off));
        vf[sb][dt].a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + off + 8 * RS));
        if constexpr (X3) {
          vf[sb][dt].b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + VLO + off));
          vf[sb][dt].b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sV + VLO + off + 8 * RS));
        }
      }
  };
  auto pv_mfma = [&](int i, const VFrag (&vf)[2][DT]) {
    const int blk = i / NM, w = i - blk * NM;
    const int sb = blk / DT, dt = blk - sb * DT;
    if (SIU3R_AP_DBG & 1) { oacc[dt][i & 15] += (float)vf[sb][dt].a0[0]; return; }
    const bf16x8 vhi = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vf[sb][dt].a0, vf[sb][dt].a1, 0, 1, 2, 3, 4, 5, 6, 7));
    if (X3 && w == 0) {
      const bf16x8 vlo = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vf[sb][dt].b0, vf[sb][dt].b1, 0, 1, 2, 3, 4, 5, 6, 7));
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph[sb], oacc[dt], 0, 0, 0);
    } else if (X3 && w == 1) {
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, plo[X3 ? sb : 0], oacc[dt], 0, 0, 0);
    } else {
      oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, ph[sb], oacc[dt], 0, 0, 0);
    }
  };

  // ---- schedule.  Iteration kt enters with S(kt) in s_cur and the K fragments of tile kt+1 in registers, and
  //   reads the V fragments of tile kt (stored two iterations ago)                       -- latency under phase 1
  //   phase 1: S(kt+1) MFMAs  ||  softmax of S(kt)
  //   reads the K fragments of tile kt+2 (stored last iteration) for the NEXT iteration  -- latency under phase 2
  //   phase 2: P V(kt) MFMAs  ||  split / store of K(kt+3) and V(kt+2), side path, global loads of K(kt+4) and V(kt+3)
  //   barrier
  // so that no LDS read sits between a barrier and the MFMAs that need it.  K ring: 2 stages (tile kt+2 readable, kt+3 being written:
  // the stage of tile kt+1, whose fragments left it before the previous barrier); V ring: 3 stages (kt, kt+1, kt+2).
  u32x4v rk[NR], rv[NR];
  float s_pend = NEG_BIG;  // side path: score of the tile whose V row is staged next
  load_k(0, rk);
  load_v(0, rv);
  store_k(0, rk);
  store_v(0, rv);
  if (do_x) side_consume(side_score(0, rk), rv);
  load_k(1, rk);
  load_v(1, rv);
  store_k(1, rk);
  store_v(1, rv);
  if (do_x) side_consume(side_score(1, rk), rv);
  load_k(2, rk);
  load_v(2, rv);
  lds_barrier();
  f32x16 s_cur;
#pragma unroll
  for (int r = 0; r < 16; ++r) s_cur[r] = 0.f;
  bf16x8 kf[KS], kfl[X3 ? KS : 1];
  read_k(0, kf, kfl);
#pragma unroll
  for (int i = 0; i < NSM; ++i) s_mfma(i, s_cur, kf, kfl);
  read_k(1, kf, kfl);
  lds_barrier();  // every wave holds its fragments of K(0) and K(1): stage 0 may take K(2)
  store_k(0, rk);
  if (do_x) s_pend = side_score(2, rk);
  load_k(3, rk);
  lds_barrier();  // K(2) is readable from iteration 0 on

  int sv_r = 0, sv_w = 2;  // V stage of tile kt / kt + 2
  for (int kt = 0; kt < nkt - 1; ++kt) {
    VFrag vf[2][DT];
    // phase 1.  Chunk i = MFMA i + its share of the softmax instructions.  The empty asm statements carry the next MFMA's A operand and
    // the softmax state: they are ordered among themselves, so neither instruction selection nor the scheduler can regroup the chunks
    // (sched_group_barrier pipelines left all MFMAs in front of the VALU work here).
    f32x16 s_nxt;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_nxt[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NSM; ++i) {
      s_mfma(i, s_nxt, kf, kfl);
      if (i == 0) read_v(sv_r, vf);  // (behind the first MFMA: in front of it, the MFMA would wait for the first of these reads)
#pragma unroll
      for (int g = i * NV / NSM; g < (i + 1) * NV / NSM; ++g) softmax_group(g, s_cur);
      if (i + 1 < NSM) {
        const int ks1 = (i + 1) / NM;
        if (X3 && (i + 1) % NM == 0) asm volatile("" : "+v"(kfl[X3 ? ks1 : 0]), "+v"(s_cur), "+v"(nm), "+v"(psum));
        else asm volatile("" : "+v"(kf[ks1]), "+v"(s_cur), "+v"(nm), "+v"(psum));
      }
    }
    // (a side-effecting use above the rescale branch: without it the exponentials sink past the branch, out of the MFMAs' shadow)
    if constexpr (X3) asm volatile("" ::"v"(ph[0]), "v"(ph[1]), "v"(plo[0]), "v"(plo[1]), "v"(l_run));
    else asm volatile("" ::"v"(ph[0]), "v"(ph[1]), "v"(l_run));
    __builtin_amdgcn_sched_barrier(0);
    rescale_o();
    // phase 2
    read_k(kt & 1, kf, kfl);  // K(kt+2)
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      pv_mfma(i, vf);
      if (!(SIU3R_AP_DBG & 4)) {
        if (i == 0) {
          store_k((kt + 1) & 1, rk);  // K(kt+3)
          if (do_x) {
            const float s_new = side_score(kt + 3, rk);
            side_consume(s_pend, rv);  // tile kt+2
            s_pend = s_new;
          }
        }
        if (i == NPV / 3) store_v(sv_w, rv);  // V(kt+2)
        if (i == NPV - 1) {
          load_k(kt + 4, rk);
          load_v(kt + 3, rv);
        }
      }
      if (i + 1 < NPV) {
        const int blk1 = (i + 1) / NM, sb1 = blk1 / DT, dt1 = blk1 - sb1 * DT;
        if (X3 && (i + 1) % NM == 0) asm volatile("" : "+v"(vf[sb1][dt1].b0)::"memory");
        else asm volatile("" : "+v"(vf[sb1][dt1].a0)::"memory");
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    s_cur = s_nxt;
    sv_r = sv_r == NVS - 1 ? 0 : sv_r + 1;
    sv_w = sv_w == NVS - 1 ? 0 : sv_w + 1;
  }
  {  // last tile: ragged key mask, softmax, P V
    const int kt = nkt - 1;
    if ((p.Nk & (KT - 1)) != 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.Nk) s_cur[r] = NEG_BIG;
    }
#pragma unroll
    for (int g = 0; g < NV; ++g) softmax_group(g, s_cur);
    rescale_o();
    VFrag vf[2][DT];
    read_v(sv_r, vf);
#pragma unroll
    for (int i = 0; i < NPV; ++i) pv_mfma(i, vf);
  }
  // (the side path has consumed tiles 0 .. nkt: every store of the loop met its V row; tiles >= nkt carry masked scores)

  // ---- merge the two key halves of every query group (odd wave parks (m, l, O), even wave folds them into its own) and the 64
  // 8-lane groups of the side path; then write O / l
  float l_tot = l_run + other_half(l_run);
  lds_barrier();  // every wave is done with the K / V stages
  float* xs = (float*)smem + qg * (32 * (D + 2));
  float* xq = (float*)smem + 4 * (32 * (D + 2));  // side path: [64 groups][D + 2]
  if (kb) {
    if (lh == 0) {
      xs[l31 * (D + 2) + D] = m_run;
      xs[l31 * (D + 2) + D + 1] = l_tot;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[l31 * (D + 2) + dt * 32 + 8 * g + 4 * lh + e] = oacc[dt][4 * g + e];
  }
  if (do_x) {
#pragma unroll
    for (int e = 0; e < 8; ++e) xq[ld_key * (D + 2) + ch * 8 + e] = ox[e];
    if (ch == 0) {
      xq[ld_key * (D + 2) + D] = mx;
      xq[ld_key * (D + 2) + D + 1] = lx;
    }
  }
  lds_barrier();
  constexpr int OESZ = X3 ? 4 : 2;
  if (kb) {
    if (do_x && wave == 1) {  // lane = d of query Nq - 1
      float M = NEG_BIG;
      for (int g = 0; g < 64; ++g) M = fmaxf(M, xq[g * (D + 2) + D]);
      float L = 0.f, o = 0.f;
      for (int g = 0; g < 64; ++g) {
        const float w = __builtin_amdgcn_exp2f((xq[g * (D + 2) + D] - M) * sl2);
        L += w * xq[g * (D + 2) + D + 1];
        o += w * xq[g * (D + 2) + lane];
      }
      unsigned char* op = (unsigned char*)p.out + (((int64_t)b * p.Nq + (p.Nq - 1)) * ((int64_t)p.H * D) + (int64_t)h * D + lane) * OESZ;
      if constexpr (X3) *(float*)op = o / L;
      else *(u16*)op = f32_to_bf16_bits(o / L);
    }
    return;
  }
  const float m1 = xs[l31 * (D + 2) + D], l1 = xs[l31 * (D + 2) + D + 1];
  const float M = fmaxf(m_run, m1);
  const float w0 = __builtin_amdgcn_exp2f((m_run - M) * sl2), w1 = __builtin_amdgcn_exp2f((m1 - M) * sl2);
  const float inv = 1.f / (l_tot * w0 + l1 * w1);
  if (!q_ok) return;
  unsigned char* op = (unsigned char*)p.out + (((int64_t)b * p.Nq + q_row) * ((int64_t)p.H * D) + (int64_t)h * D) * OESZ;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (oacc[dt][4 * g + e] * w0 + xs[l31 * (D + 2) + dt * 32 + 8 * g + 4 * lh + e] * w1) * inv;
      const int d = dt * 32 + 8 * g + 4 * lh;
      if constexpr (X3) {
        *(float4*)(op + d * 4) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        uint2 w;
        w.x = pack_bf16x2(o[0], o[1]);
        w.y = pack_bf16x2(o[2], o[3]);
        *(uint2*)(op + d * 2) = w;
      }
    }
}

}  // namespace siu3r_attn_pipe

// eligible: head_dim 64, no mask / RoPE-on-load / key-range split, at least one full key tile (every wave's first key block is
// then fully valid, so its running maximum is finite from tile 0 on)
bool siu3r_attn_pipe_ok(const siu3r_attn_params& p) {
  static const bool off = getenv("SIU3R_ATTN_NO_PIPE") != nullptr;  // A/B switch
  static const int bf16_mode = getenv("SIU3R_ATTN_PIPE_BF16") ? atoi(getenv("SIU3R_ATTN_PIPE_BF16")) : -1;  // 0 never, 1 always
  if (off || p.D != 64 || p.mask || p.rope_cos || (p.splits > 1 && p.ws) || p.Nk < 64) return false;
  if (p.dtype == SIU3R_F32) return p.split3 != 0;
  if (p.split3 || bf16_mode == 0) return false;
  // bf16 tensors: measured 35.7 -> 25.4 us on the pair shape (256 workgroups) but 18.6 -> 22.3 us on 96 workgroups, where
  // attn_fast_kernel's 64-query workgroups fill more CUs
  return bf16_mode == 1 || (int64_t)((p.Nq + siu3r_attn_pipe::QT - 1) / siu3r_attn_pipe::QT) * p.H * p.B >= 192;
}

int siu3r_attn_pipe_launch(const siu3r_attn_params& p, hipStream_t s) {
  using namespace siu3r_attn_pipe;
  const int has_x = (p.Nq > QT && (p.Nq & (QT - 1)) == 1) ? 1 : 0;
  const int q_tiles = has_x ? (p.Nq - 1) / QT : (p.Nq + QT - 1) / QT;
  const dim3 grid((unsigned)(q_tiles * p.H * p.B)), block(NT);
  if (p.dtype == SIU3R_F32 && p.kv_x3)
    hipLaunchKernelGGL((attn_pipe_kernel<true, true>), grid, block, 0, s, p, q_tiles, has_x);
  else if (p.dtype == SIU3R_F32)
    hipLaunchKernelGGL((attn_pipe_kernel<true>), grid, block, 0, s, p, q_tiles, has_x);
  else
    hipLaunchKernelGGL((attn_pipe_kernel<false>), grid, block, 0, s, p, q_tiles, has_x);
  SIU3R_LAUNCH_CHECK("siu3r_attention(pipe)");
  return 0;
}
