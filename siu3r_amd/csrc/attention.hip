// Fused (flash-style) attention for gfx950: S = (Q K^T) * scale -> online softmax -> O = P V,
// with RoPE2D applied to Q and K on load (reference croco/blocks.py:94-112,149-169 + curope kernels.cu:17-82)
// and an optional boolean key mask (Mask2Former masked cross-attention, video_seg_decoder.py:946-983).
//
// Work split: one workgroup (4 waves) per (batch, head, 128-query tile); each wave owns 32 queries.
// K/V are streamed in 64-key tiles through double-buffered LDS (register-staged, one barrier per tile).
// The score tile is computed TRANSPOSED (S^T = K Q^T, v_mfma_f32_32x32x16_bf16) so that every lane owns one
// query column: row max / row sum are in-lane reductions plus one cross-half exchange, and the softmax
// rescale factor is a per-lane scalar.  The P^T accumulator layout is fed straight back as the MFMA B operand
// of O^T = V^T P^T using a "virtual k" ordering (key(h,j) = 16*sb + (j&3) + 8*(j>>2) + 4*h); V^T is built in
// LDS with that same ordering in mind, so no cross-lane shuffle is needed between the two GEMMs.
// SPLIT = bf16x3 mode: Q,K,P,V are split hi+lo and every product uses three MFMAs (~fp32 accuracy).
#include <stdlib.h>

#include "common.h"

// attention_pipe.hip: the 8-wave software-pipelined kernel of the ViT blocks
bool siu3r_attn_pipe_ok(const siu3r_attn_params& p);
int siu3r_attn_pipe_launch(const siu3r_attn_params& p, hipStream_t s);

namespace {

constexpr int KT = 64;           // keys per tile
constexpr int VT_STRIDE = 136;   // bytes per V^T row: 64 keys * 2 B + 8 B pad (conflict-free ds_read_b64)
constexpr float NEG_BIG = -1.0e30f;

template <int D>
__device__ __forceinline__ int k_off(int row, int chunk) {
  if (D == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int D, int IN_F32, int SPLIT, int MASK>
__global__ __launch_bounds__(256) void attn_kernel(const siu3r_attn_params p) {
  constexpr int PL = SPLIT ? 2 : 1;
  constexpr int K_BYTES = KT * D * 2;
  constexpr int V_BYTES = D * VT_STRIDE;
  constexpr int STAGE = PL * (K_BYTES + V_BYTES);
  constexpr int CH = D / 32;  // chunks (8 elements) per thread per tile: 64 keys * D/8 chunks / 256 threads
  constexpr int KS = D / 16;  // k-substeps of the score GEMM
  constexpr int DT = D / 32;  // 32-row output tiles of O^T
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_row = blockIdx.x * 128 + wave * 32 + l31;
  const bool q_ok = q_row < p.Nq;
  const int esz = IN_F32 ? 4 : 2;
  const bool rope = p.rope_cos != nullptr;
  const int Q4 = D / 4;

  // ---------------- Q fragments (B operand: lane = query, 8 consecutive d per k-substep) ----------------
  bf16x8 qf[KS], qfl[KS];
  {
    float qv[KS][8];
    const unsigned char* qp = (const unsigned char*)p.q + ((int64_t)b * p.q_sb + (int64_t)q_row * p.q_sn + (int64_t)h * p.q_sh) * esz;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (q_ok) {
        f32x8 v = load8_as_f32(qp, IN_F32 ? SIU3R_F32 : SIU3R_BF16, ks * 16 + 8 * lh);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[ks][j] = v.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[ks][j] = 0.f;
      }
    }
    if constexpr (D == 64) if (rope && q_ok) {
      // head vector = [u_Y v_Y u_X v_X], quarters of D; for D=64: ks 0/1 = (u_Y, v_Y), ks 2/3 = (u_X, v_X)
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const int pos = (int)p.qpos[((int64_t)b * p.Nq + q_row) * 2 + ax];
        const float* cs = p.rope_cos + (int64_t)pos * Q4 + 8 * lh;
        const float* sn = p.rope_sin + (int64_t)pos * Q4 + 8 * lh;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float c = cs[j], s = sn[j];
          const float u = qv[2 * ax][j], v = qv[2 * ax + 1][j];
          qv[2 * ax][j] = u * c - v * s;
          qv[2 * ax + 1][j] = v * c + u * s;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (SPLIT) {
        uint4 hi, lo;
        split_bf16x8(qv[ks], hi, lo);
        qf[ks] = as_bf16x8(hi);
        qfl[ks] = as_bf16x8(lo);
      } else {
        qf[ks] = as_bf16x8(pack_bf16x8(qv[ks]));
      }
    }
  }

  // ---------------- K/V tile staging ----------------
  // D=64: thread -> (key = t>>2, chunk pair (c0, c0+2), c0 in {0,1,4,5});  D=32: (key = t>>2, chunk = t&3)
  const int ld_key = t >> 2;
  const int ld_c0 = (D == 64) ? ((t & 1) + 4 * ((t >> 1) & 1)) : (t & 3);
  // Raw staging registers: the global loads are consumed only at LDS-store time (one tile later), so no wait is
  // forced at the load site and two tiles stay in flight.  RoPE (when requested here rather than in the producing
  // GEMM's epilogue) is applied at store time with the positions fetched alongside.
  struct KVRegs {
    uint4 kb[IN_F32 ? 2 * CH : CH], vb[IN_F32 ? 2 * CH : CH];
    int pos;
    bool ok;
  };

  auto load_tile = [&](int kt, KVRegs& rg) {
    int key = kt * KT + ld_key;
    rg.ok = key < p.Nk;
    if (key > p.Nk - 1) key = p.Nk - 1;  // clamped address, zeroed at store time (branch-free loads)
    const unsigned char* kp = (const unsigned char*)p.k + ((int64_t)(b ^ p.kv_bxor) * p.k_sb + (int64_t)key * p.k_sn + (int64_t)h * p.k_sh) * esz;
    const unsigned char* vp = (const unsigned char*)p.v + ((int64_t)(b ^ p.kv_bxor) * p.v_sb + (int64_t)key * p.v_sn + (int64_t)h * p.v_sh) * esz;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = ld_c0 + 2 * c;
      if constexpr (IN_F32) {
        rg.kb[2 * c] = *(const uint4*)(kp + ch * 32);
        rg.kb[2 * c + 1] = *(const uint4*)(kp + ch * 32 + 16);
        rg.vb[2 * c] = *(const uint4*)(vp + ch * 32);
        rg.vb[2 * c + 1] = *(const uint4*)(vp + ch * 32 + 16);
      } else {
        rg.kb[c] = *(const uint4*)(kp + ch * 16);
        rg.vb[c] = *(const uint4*)(vp + ch * 16);
      }
    }
    rg.pos = 0;
    if (rope && D == 64) rg.pos = (int)p.kpos[((int64_t)b * p.Nk + key) * 2 + (ld_c0 >> 2)];
  };

  auto unpack = [&](const uint4* raw, int c, float (&f)[8], bool ok) {
    if constexpr (IN_F32) {
      const float* a = (const float*)&raw[2 * c];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = ok ? a[j] : 0.f;
    } else {
      const uint32_t* w = (const uint32_t*)&raw[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[2 * j] = ok ? bf16_bits_to_f32((u16)(w[j] & 0xffff)) : 0.f;
        f[2 * j + 1] = ok ? bf16_bits_to_f32((u16)(w[j] >> 16)) : 0.f;
      }
    }
  };

  auto store_tile = [&](int stage, const KVRegs& rg) {
    float kreg[CH][8], vreg[CH][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      unpack(rg.kb, c, kreg[c], rg.ok);
      unpack(rg.vb, c, vreg[c], rg.ok);
    }
    if (rope && D == 64) {
      const float* cs = p.rope_cos + (int64_t)rg.pos * Q4 + (ld_c0 & 1) * 8;
      const float* sn = p.rope_sin + (int64_t)rg.pos * Q4 + (ld_c0 & 1) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = cs[j], s2 = sn[j];
        const float u = kreg[0][j], v = kreg[CH - 1][j];
        kreg[0][j] = u * c - v * s2;
        kreg[CH - 1][j] = v * c + u * s2;
      }
    }
    unsigned char* sK = smem + stage * STAGE;
    unsigned char* sV = sK + K_BYTES;
    unsigned char* sKl = sK + K_BYTES + V_BYTES;
    unsigned char* sVl = sKl + K_BYTES;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = ld_c0 + 2 * c;
      const int ko = k_off<D>(ld_key, ch);
      if (SPLIT) {
        uint4 hi, lo;
        split_bf16x8(kreg[c], hi, lo);
        *(uint4*)(sK + ko) = hi;
        *(uint4*)(sKl + ko) = lo;
      } else {
        *(uint4*)(sK + ko) = pack_bf16x8(kreg[c]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = ch * 8 + j;
        const u16 hb = f32_to_bf16_bits(vreg[c][j]);
        *(u16*)(sV + d * VT_STRIDE + ld_key * 2) = hb;
        if (SPLIT) *(u16*)(sVl + d * VT_STRIDE + ld_key * 2) = f32_to_bf16_bits(vreg[c][j] - bf16_bits_to_f32(hb));
      }
    }
  };

  // ---------------- main loop ----------------
  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int nkt = (p.Nk + KT - 1) / KT;

  // key mask (Mask2Former): the 64 mask bytes of this lane's query row for one KV tile, as 4 x 16-B loads that are
  // issued one tile ahead; rows beyond Nq read a clamped row (never stored).  Requires Nk % 64 == 0.
  const uint8_t* mrow = nullptr;
  if constexpr (MASK) mrow = p.mask + ((int64_t)b * p.Nq + (q_ok ? q_row : p.Nq - 1)) * p.mask_ld;
  auto load_mask = [&](int kt, uint4 (&mk)[4]) {
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mk[i] = *(const uint4*)(mrow + (int64_t)kt * KT + 16 * i);
    }
  };

  auto process = [&](int kt, int cur, const uint4 (&mk)[4]) {
    const unsigned char* sK = smem + cur * STAGE;
    const unsigned char* sV = sK + K_BYTES;
    const unsigned char* sKl = sK + K_BYTES + V_BYTES;
    const unsigned char* sVl = sKl + K_BYTES;

    // S^T tile: rows = keys (2 blocks of 32), cols = this wave's 32 queries
    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int o = k_off<D>(kb * 32 + l31, ks * 2 + lh);
        const bf16x8 kf = as_bf16x8(*(const uint4*)(sK + o));
        if (SPLIT) {
          const bf16x8 kfl = as_bf16x8(*(const uint4*)(sKl + o));
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qf[ks], sacc[kb], 0, 0, 0);
          sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfl[ks], sacc[kb], 0, 0, 0);
        }
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[kb], 0, 0, 0);
      }
    }
    // scale, key-validity and mask; lane holds keys key(kb,r) = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*lh
    float mx = NEG_BIG;
    // this lane's keys sit in dwords {lh, 2+lh, ...} of the 16 mask dwords: select even/odd once, index statically
    uint32_t wsel[8];
    if constexpr (MASK) {
      const uint32_t* w = (const uint32_t*)&mk[0];
#pragma unroll
      for (int i = 0; i < 8; ++i) wsel[i] = lh ? w[2 * i + 1] : w[2 * i];
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bool ok = key < p.Nk;
        if constexpr (MASK) ok = ok && ((wsel[kb * 4 + (r >> 2)] >> (8 * (r & 3))) & 0xffu) == 0;
        const float s = ok ? sacc[kb][r] * sl2 : NEG_BIG;
        sacc[kb][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(sacc[kb][r] - m_new);
        sacc[kb][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

    // O^T += V^T P^T : 4 k-substeps of 16 keys; B operand = P registers as they are (virtual-k order)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        float pf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = sacc[kb][8 * sb + j];
        bf16x8 ph, plo;
        if (SPLIT) {
          uint4 hi, lo;
          split_bf16x8(pf, hi, lo);
          ph = as_bf16x8(hi);
          plo = as_bf16x8(lo);
        } else {
          ph = as_bf16x8(pack_bf16x8(pf));
        }
        const int kbase = (kb * 32 + 16 * sb + 4 * lh) * 2;  // byte offset of the lane's first 4 keys
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int vo = (dt * 32 + l31) * VT_STRIDE + kbase;
          const uint2 a0 = *(const uint2*)(sV + vo);
          const uint2 a1 = *(const uint2*)(sV + vo + 16);
          const bf16x8 vf = as_bf16x8(make_uint4(a0.x, a0.y, a1.x, a1.y));
          if (SPLIT) {
            const uint2 b0 = *(const uint2*)(sVl + vo);
            const uint2 b1 = *(const uint2*)(sVl + vo + 16);
            const bf16x8 vfl = as_bf16x8(make_uint4(b0.x, b0.y, b1.x, b1.y));
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, plo, oacc[dt], 0, 0, 0);
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph, oacc[dt], 0, 0, 0);
        }
      }
  };

  // prefetch distance 2: two register sets alternate, LDS double buffer, one barrier per KV tile
  KVRegs r0, r1;
  uint4 mA[4], mB[4];
  load_tile(0, r0);
  load_mask(0, mA);
  if (nkt > 1) load_tile(1, r1);
  // raw barriers: __syncthreads() would drain the two KV tiles in flight (hipcc puts s_waitcnt vmcnt(0) in front of it)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  store_tile(0, r0);
  lds_barrier();
  for (int kt = 0; kt < nkt; kt += 2) {
    if (kt + 2 < nkt) load_tile(kt + 2, r0);
    if (kt + 1 < nkt) load_mask(kt + 1, mB);
    process(kt, 0, mA);
    if (kt + 1 < nkt) store_tile(1, r1);
    lds_barrier();
    if (kt + 1 >= nkt) break;
    if (kt + 3 < nkt) load_tile(kt + 3, r1);
    if (kt + 2 < nkt) load_mask(kt + 2, mA);
    process(kt + 1, 1, mB);
    if (kt + 2 < nkt) store_tile(0, r0);
    lds_barrier();
  }

  // ---------------- epilogue: O[q, h*D + d] = O^T[d][q] / l ----------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  if (q_ok) {
    unsigned char* op = (unsigned char*)p.out + (((int64_t)b * p.Nq + q_row) * ((int64_t)p.H * D) + (int64_t)h * D) * esz;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        const float o0 = oacc[dt][4 * g] * inv, o1 = oacc[dt][4 * g + 1] * inv, o2 = oacc[dt][4 * g + 2] * inv,
                    o3 = oacc[dt][4 * g + 3] * inv;
        if (IN_F32) {
          *(float4*)(op + d * 4) = make_float4(o0, o1, o2, o3);
        } else {
          uint2 w;
          w.x = (uint32_t)f32_to_bf16_bits(o0) | ((uint32_t)f32_to_bf16_bits(o1) << 16);
          w.y = (uint32_t)f32_to_bf16_bits(o2) | ((uint32_t)f32_to_bf16_bits(o3) << 16);
          *(uint2*)(op + d * 2) = w;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path (no RoPE here -- it is fused into the projection GEMM's epilogue --, no bf16x3 split): the path every
// attention of the bf16 forward takes.  Same work split and S^T / virtual-k structure as attn_kernel above, but
//   * K and V tiles go global -> registers -> LDS as raw 16-byte chunks: no unpack/repack, V stays ROW-major
//     ([key][D], 64-byte halves swapped by key bit 1 for D = 64) and its transpose happens in the LDS read
//     (ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 d] block and receives it d-major);
//   * scale * log2(e) rides in the fma that feeds v_exp_f32 (the running maximum stays in raw-score units, so Q is
//     used as stored: pre-scaling Q would add a bf16 rounding to every logit); the key-validity compare runs
//     only on a ragged last tile; the O rescale is skipped while no lane's running maximum grows;
//   * P is packed with v_cvt_pk_bf16_f32; the cross-half maximum / sum use v_permlane32_swap.
// Per 64-key tile and wave: 16 MFMAs (512 cycles) against ~32 v_exp_f32 (quarter rate, 512 cycles) + ~110 VALU.
// X3 = bf16x3 mode on the same structure: q / k / v / out are fp32; K and V tiles are split into hi + lo bf16 planes when they are
// stored to LDS (two plane pairs per stage), Q and P are split in registers, and every product is three MFMAs (lo*hi + hi*lo + hi*hi).
// KH = 1: a workgroup covers 64 queries; waves 2g and 2g+1 own the SAME 32 queries and each takes one 32-key half of every 64-key tile
// (its own running maximum / sum / O), merged through LDS at the end.  Twice the workgroups with half the per-tile chain per wave:
// the ViT attention of one pair has only 288 128-query workgroups for 256 CUs, i.e. one wave per SIMD and nothing to overlap with.
template <int D, int MASK, int SPLITKV, int X3 = 0, int KH = 0>
__global__ __launch_bounds__(256) void attn_fast_kernel(const siu3r_attn_params p) {
  static_assert(!(KH && SPLITKV), "key halves and key ranges are alternatives");
  constexpr int NKB = KH ? 1 : 2;   // 32-key blocks of a tile handled by one wave
  constexpr int QT = KH ? 64 : 128;  // queries per workgroup
  constexpr int K_BYTES = KT * D * 2;
  constexpr int V_BYTES = KT * D * 2;
  constexpr int PLANE = K_BYTES + V_BYTES;          // [K | V] of one bf16 plane
  constexpr int STAGE = (X3 ? 2 : 1) * PLANE;        // X3: [K hi | V hi | K lo | V lo]
  constexpr int CH = D / 32;  // 16-byte chunks per thread per tile and tensor
  constexpr int KS = D / 16;
  constexpr int DT = D / 32;
  constexpr int RS = D * 2;   // V row stride in LDS
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;  // native vector: stays in registers (HIP's uint4 struct did not)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  // SPLITKV: blockIdx.x = query tile * splits + key-range index
  const int q_tile = SPLITKV ? (int)blockIdx.x / p.splits : (int)blockIdx.x;
  const int k_split = SPLITKV ? (int)blockIdx.x - q_tile * p.splits : 0;
  const int kb0 = KH ? (wave & 1) : 0;  // first 32-key block of this wave
  const int q_row = q_tile * QT + (KH ? (wave >> 1) : wave) * 32 + l31;
  const bool q_ok = q_row < p.Nq;
  const float sl2 = p.scale * 1.4426950408889634f;

  // Q fragments (B operand: lane = query, 8 consecutive d per k-substep), raw bf16 (X3: fp32, split once)
  bf16x8 qf[KS], qfl[X3 ? KS : 1];
  if constexpr (X3) {
    const float* qp = (const float*)p.q + (int64_t)b * p.q_sb + (int64_t)(q_ok ? q_row : p.Nq - 1) * p.q_sn + (int64_t)h * p.q_sh;
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
    const u16* qp = (const u16*)p.q + (int64_t)b * p.q_sb + (int64_t)(q_ok ? q_row : p.Nq - 1) * p.q_sn + (int64_t)h * p.q_sh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = as_bf16x8(*(const uint4*)(qp + ks * 16 + 8 * lh));
    }
  }

  // staging: thread -> (key = t>>2, chunks ld_c0 + 2c)
  const int ld_key = t >> 2;
  const int ld_c0 = (D == 64) ? ((t & 1) + 4 * ((t >> 1) & 1)) : (t & 3);
  struct KVRegs { u32x4v kb[X3 ? 2 * CH : CH], vb[X3 ? 2 * CH : CH]; };
  constexpr int ESZ = X3 ? 4 : 2;
  const unsigned char* kbase = (const unsigned char*)p.k + ((int64_t)(b ^ p.kv_bxor) * p.k_sb + (int64_t)h * p.k_sh) * ESZ;
  const unsigned char* vbase = (const unsigned char*)p.v + ((int64_t)(b ^ p.kv_bxor) * p.v_sb + (int64_t)h * p.v_sh) * ESZ;
  auto load_tile = [&](int kt, KVRegs& rg) {
    int key = kt * KT + ld_key;
    if (key > p.Nk - 1) key = p.Nk - 1;  // clamped: finite garbage, its scores are masked and its P is 0
    const unsigned char* kp = kbase + (int64_t)key * p.k_sn * ESZ;
    const unsigned char* vp = vbase + (int64_t)key * p.v_sn * ESZ;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = ld_c0 + 2 * c;  // 8-element chunk
      if constexpr (X3) {
        rg.kb[2 * c] = *(const u32x4v*)(kp + ch * 32);
        rg.kb[2 * c + 1] = *(const u32x4v*)(kp + ch * 32 + 16);
        rg.vb[2 * c] = *(const u32x4v*)(vp + ch * 32);
        rg.vb[2 * c + 1] = *(const u32x4v*)(vp + ch * 32 + 16);
      } else {
        rg.kb[c] = *(const u32x4v*)(kp + ch * 16);
        rg.vb[c] = *(const u32x4v*)(vp + ch * 16);
      }
    }
  };
  auto split_chunk = [&](const u32x4v& a, const u32x4v& c, uint4& hi, uint4& lo) {
    // (whole-vector casts: element-wise __builtin_bit_cast(float, a[i]) made hipcc treat all four elements as a[0])
    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    const f32x4v fa = __builtin_bit_cast(f32x4v, a), fc = __builtin_bit_cast(f32x4v, c);
    const float f[8] = {fa[0], fa[1], fa[2], fa[3], fc[0], fc[1], fc[2], fc[3]};
    split_bf16x8(f, hi, lo);
  };
  auto store_tile = [&](int stage, const KVRegs& rg) {
    unsigned char* sK = smem + stage * STAGE;
    unsigned char* sV = sK + K_BYTES;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = ld_c0 + 2 * c;
      const int pc = (D == 64) ? ((((ch >> 2) ^ ((ld_key >> 1) & 1)) << 2) | (ch & 3)) : ch;
      if constexpr (X3) {
        uint4 hi, lo;
        split_chunk(rg.kb[2 * c], rg.kb[2 * c + 1], hi, lo);
        *(uint4*)(sK + k_off<D>(ld_key, ch)) = hi;
        *(uint4*)(sK + PLANE + k_off<D>(ld_key, ch)) = lo;
        split_chunk(rg.vb[2 * c], rg.vb[2 * c + 1], hi, lo);
        *(uint4*)(sV + ld_key * RS + pc * 16) = hi;
        *(uint4*)(sV + PLANE + ld_key * RS + pc * 16) = lo;
      } else {
        *(u32x4v*)(sK + k_off<D>(ld_key, ch)) = rg.kb[c];
        *(u32x4v*)(sV + ld_key * RS + pc * 16) = rg.vb[c];
      }
    }
  };

  f32x16 oacc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const int nkt = (p.Nk + KT - 1) / KT;
  const bool ragged = (p.Nk & (KT - 1)) != 0;
  int kt0 = 0, kt1 = nkt;  // this workgroup's range of 64-key tiles
  if (SPLITKV) {
    const int per = (nkt + p.splits - 1) / p.splits;
    kt0 = k_split * per;
    kt1 = min(nkt, kt0 + per);
  }

  const uint8_t* mrow = nullptr;
  if constexpr (MASK) mrow = p.mask + ((int64_t)b * p.Nq + (q_ok ? q_row : p.Nq - 1)) * p.mask_ld;
  auto load_mask = [&](int kt, uint4 (&mk)[4]) {
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mk[i] = *(const uint4*)(mrow + (int64_t)kt * KT + 16 * i);
    }
  };

  // per-lane byte offset of the transposed V reads inside a stage: row (lane&15)>>2 of a 4-key group, 16-d block
  // (lane>>4)&1, 4-d piece lane&3; the 64-byte half of the row is chosen per read
  const int i2 = (lane & 15) >> 2;
  const int v_lane = (4 * lh + i2) * RS + 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  const int v_hswz = (D == 64) ? ((i2 >> 1) & 1) : 0;

  // value held by lane ^ 32.  v_permlane32_swap exchanges the upper half of its first operand with the lower half of
  // its second: with distinct registers the results are (low half, high half) broadcasts, but when the register
  // allocator gives both operands the SAME register (it does, inside this kernel) both results are the half-swapped
  // input.  Selecting r1 in the lower lanes and r0 in the upper ones is the other half's value in either case.
  auto other_half = [&](float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, lh ? sw[0] : sw[1]);
  };
  auto process = [&](int kt, int cur, const uint4 (&mk)[4]) {
    const unsigned char* sK = smem + cur * STAGE;
    const unsigned char* sV = sK + K_BYTES;
    f32x16 sacc[NKB];  // sacc[j] = key block kb0 + j
#pragma unroll
    for (int j = 0; j < NKB; ++j) {
      const int kb = kb0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = as_bf16x8(*(const uint4*)(sK + k_off<D>(kb * 32 + l31, ks * 2 + lh)));
        if constexpr (X3) {
          const bf16x8 kfl = as_bf16x8(*(const uint4*)(sK + PLANE + k_off<D>(kb * 32 + l31, ks * 2 + lh)));
          sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qf[ks], sacc[j], 0, 0, 0);
          sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfl[ks], sacc[j], 0, 0, 0);
        }
        sacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc[j], 0, 0, 0);
      }
    }
    // lane holds keys key(kb,r) = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*lh of query l31
    if (ragged && kt == nkt - 1) {
#pragma unroll
      for (int j = 0; j < NKB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * KT + (kb0 + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.Nk) sacc[j][r] = NEG_BIG;
    }
    if constexpr (MASK) {
      const uint32_t* w = (const uint32_t*)&mk[0];
#pragma unroll
      for (int j = 0; j < NKB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kb = kb0 + j;
          const uint32_t ws = lh ? w[2 * (kb * 4 + g) + 1] : w[2 * (kb * 4 + g)];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((ws >> (8 * e)) & 0xffu) sacc[j][4 * g + e] = NEG_BIG;
        }
    }
    float mx = fmaxf(sacc[0][0], sacc[NKB - 1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, sacc[0][r]), sacc[NKB - 1][r]);  // v_max3_f32
    mx = fmaxf(mx, other_half(mx));
    const float m_new = fmaxf(m_run, mx);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {  // wave-uniform: rescale only when some maximum grew
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      m_run = m_new;
    }
    float psum = 0.f;
    const float nm = -m_run * sl2;  // exp2((s - m) * sl2) as one fma + v_exp_f32 per score; m tracks the RAW maximum
#pragma unroll
    for (int j = 0; j < NKB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[j][r], sl2, nm));
        sacc[j][r] = pv;
        psum += pv;
      }
    l_run += psum;

    // O^T += V^T P^T : B operand = P registers as they are (virtual-k order), A operand = transposed LDS reads
#pragma unroll
    for (int jb = 0; jb < NKB; ++jb)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const int kb = kb0 + jb;
        float pf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = sacc[jb][8 * sb + j];
        bf16x8 ph, plo;
        if constexpr (X3) {
          uint4 hi, lo;
          split_bf16x8(pf, hi, lo);
          ph = as_bf16x8(hi);
          plo = as_bf16x8(lo);
        } else {
          ph = as_bf16x8(pack_bf16x8(pf));
        }
        const int rbase = (kb * 32 + 16 * sb) * RS + v_lane;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int ho = (D == 64) ? ((dt ^ v_hswz) * 64) : 0;
          auto p0 = (__attribute__((address_space(3))) s16x4*)(sV + rbase + ho);
          auto p1 = (__attribute__((address_space(3))) s16x4*)(sV + rbase + 8 * RS + ho);
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          if constexpr (X3) {
            auto q0 = (__attribute__((address_space(3))) s16x4*)(sV + PLANE + rbase + ho);
            auto q1 = (__attribute__((address_space(3))) s16x4*)(sV + PLANE + rbase + 8 * RS + ho);
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q0);
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q1);
            const bf16x8 vfl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph, oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, plo, oacc[dt], 0, 0, 0);
          }
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph, oacc[dt], 0, 0, 0);
        }
      }
  };

  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  KVRegs r0, r1;
  uint4 mA[4], mB[4];
  if (kt0 < kt1) {
    load_tile(kt0, r0);
    load_mask(kt0, mA);
    if (kt0 + 1 < kt1) load_tile(kt0 + 1, r1);
    store_tile(0, r0);
    lds_barrier();
    for (int kt = kt0; kt < kt1; kt += 2) {
      if (kt + 2 < kt1) load_tile(kt + 2, r0);
      if (kt + 1 < kt1) load_mask(kt + 1, mB);
      process(kt, 0, mA);
      if (kt + 1 < kt1) store_tile(1, r1);
      lds_barrier();
      if (kt + 1 >= kt1) break;
      if (kt + 3 < kt1) load_tile(kt + 3, r1);
      if (kt + 2 < kt1) load_mask(kt + 2, mA);
      process(kt + 1, 1, mB);
      if (kt + 2 < kt1) store_tile(0, r0);
      lds_barrier();
    }
  }

  // epilogue: O[q, h*D + d] = O^T[d][q] / l
  float l_tot = l_run + other_half(l_run);
  if constexpr (KH) {
    // merge the two key halves of a query group: the odd wave parks (m, l, O) in LDS, the even wave folds them into its own
    lds_barrier();  // every wave is done with the K / V stages
    float* xs = (float*)smem + (wave >> 1) * (32 * (D + 2));
    if (wave & 1) {
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
    lds_barrier();
    if (wave & 1) return;
    const float m1 = xs[l31 * (D + 2) + D], l1 = xs[l31 * (D + 2) + D + 1];
    const float M = fmaxf(m_run, m1);
    const float w0 = __builtin_amdgcn_exp2f((m_run - M) * sl2), w1 = __builtin_amdgcn_exp2f((m1 - M) * sl2);
    l_tot = l_tot * w0 + l1 * w1;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          oacc[dt][4 * g + e] = oacc[dt][4 * g + e] * w0 + xs[l31 * (D + 2) + dt * 32 + 8 * g + 4 * lh + e] * w1;
  }
  if (SPLITKV) {
    // partial result of this key range: un-normalised O (relative to m_run), m_run and l; an empty range contributes
    // (NEG_BIG, 0, 0), which the combine step weights by exp2(-inf) = 0
    const int64_t nqp = (int64_t)((p.Nq + 127) / 128) * 128;
    float* w = p.ws + ((((int64_t)b * p.H + h) * p.splits + k_split) * nqp + q_row) * (D + 4);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(w + dt * 32 + 8 * g + 4 * lh) = make_float4(oacc[dt][4 * g], oacc[dt][4 * g + 1], oacc[dt][4 * g + 2], oacc[dt][4 * g + 3]);
    if (lh == 0) {
      w[D] = m_run;
      w[D + 1] = l_tot;
    }
    return;
  }
  const float inv = 1.f / l_tot;
  if constexpr (X3) {
    if (q_ok) {
      float* op = (float*)p.out + ((int64_t)b * p.Nq + q_row) * ((int64_t)p.H * D) + (int64_t)h * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(op + dt * 32 + 8 * g + 4 * lh) = make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
    }
    return;
  }
  if (q_ok) {
    u16* op = (u16*)p.out + ((int64_t)b * p.Nq + q_row) * ((int64_t)p.H * D) + (int64_t)h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        uint2 w;
        w.x = pack_bf16x2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
        w.y = pack_bf16x2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        *(uint2*)(op + d) = w;
      }
  }
}

// combine the key-range partials of attn_fast_kernel<.., SPLITKV=1>: one thread per (b, h, query, 4 d)
template <int D>
__global__ void attn_combine_kernel(const siu3r_attn_params p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int d4 = (int)(i % (D / 4));
  const int64_t r = i / (D / 4);
  const int q = (int)(r % p.Nq);
  const int64_t bh = r / p.Nq;
  if (bh >= (int64_t)p.B * p.H) return;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int64_t nqp = (int64_t)((p.Nq + 127) / 128) * 128;
  const float* w = p.ws + (bh * p.splits * nqp + q) * (D + 4);
  float M = NEG_BIG;
  for (int s_ = 0; s_ < p.splits; ++s_) M = fmaxf(M, w[(int64_t)s_ * nqp * (D + 4) + D]);
  float L = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
  for (int s_ = 0; s_ < p.splits; ++s_) {
    const float* ws = w + (int64_t)s_ * nqp * (D + 4);
    const float wt = __builtin_amdgcn_exp2f((ws[D] - M) * sl2);
    const float4 o = *(const float4*)(ws + 4 * d4);
    L += wt * ws[D + 1];
    o0 += wt * o.x; o1 += wt * o.y; o2 += wt * o.z; o3 += wt * o.w;
  }
  const float inv = 1.f / L;
  const int b = (int)(bh / p.H), h = (int)(bh % p.H);
  if (p.dtype == SIU3R_F32) {
    *(float4*)((float*)p.out + ((int64_t)b * p.Nq + q) * ((int64_t)p.H * D) + (int64_t)h * D + 4 * d4) = make_float4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
    return;
  }
  u16* op = (u16*)p.out + ((int64_t)b * p.Nq + q) * ((int64_t)p.H * D) + (int64_t)h * D + 4 * d4;
  uint2 pk;
  pk.x = pack_bf16x2(o0 * inv, o1 * inv);
  pk.y = pack_bf16x2(o2 * inv, o3 * inv);
  *(uint2*)op = pk;
}

// measured (tools/mb_attn.py): bf16 108 workgroups 22.9 -> 18.4 us, 288 workgroups 36.0 -> 39.0 us; bf16x3 288 workgroups 78 -> 70 us
static const int64_t g_attn_kh_max = getenv("SIU3R_ATTN_KH_MAX") ? atoll(getenv("SIU3R_ATTN_KH_MAX")) : -1;

template <int D, int X3>
int launch_fast(const siu3r_attn_params& p, hipStream_t s) {
  dim3 block(256);
  const int qtiles = (p.Nq + 127) / 128;
  if (p.splits > 1 && p.ws) {
    dim3 grid(qtiles * p.splits, p.H, p.B);
    if (p.mask)
      hipLaunchKernelGGL((attn_fast_kernel<D, 1, 1, X3>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((attn_fast_kernel<D, 0, 1, X3>), grid, block, 0, s, p);
    const int64_t n = (int64_t)p.B * p.H * p.Nq * (D / 4);
    hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((unsigned)((n + 255) / 256)), block, 0, s, p);
  } else if (!p.mask && p.Nk >= 64 && (int64_t)qtiles * p.H * p.B < (g_attn_kh_max >= 0 ? g_attn_kh_max : (X3 ? 512 : 256))) {
    // too few 128-query workgroups to give every SIMD two waves: 64-query workgroups whose wave pairs split each key tile
    dim3 grid((p.Nq + 63) / 64, p.H, p.B);
    hipLaunchKernelGGL((attn_fast_kernel<D, 0, 0, X3, 1>), grid, block, 0, s, p);
  } else {
    dim3 grid(qtiles, p.H, p.B);
    if (p.mask)
      hipLaunchKernelGGL((attn_fast_kernel<D, 1, 0, X3>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((attn_fast_kernel<D, 0, 0, X3>), grid, block, 0, s, p);
  }
  SIU3R_LAUNCH_CHECK("siu3r_attention(fast)");
  return 0;
}

template <int D, int IN_F32, int SPLIT>
int launch(const siu3r_attn_params& p, hipStream_t s) {
  dim3 grid((p.Nq + 127) / 128, p.H, p.B), block(256);
  if (p.mask)
    hipLaunchKernelGGL((attn_kernel<D, IN_F32, SPLIT, 1>), grid, block, 0, s, p);
  else
    hipLaunchKernelGGL((attn_kernel<D, IN_F32, SPLIT, 0>), grid, block, 0, s, p);
  SIU3R_LAUNCH_CHECK("siu3r_attention");
  return 0;
}

}  // namespace

static bool kv_x3_ok(const siu3r_attn_params& p) {
  static const bool no_fast = getenv("SIU3R_ATTN_NO_FAST") != nullptr;
  // whole segments: a head's 64 dims start on a 32-column boundary of the projection output, rows and batch items likewise
  const bool seg = p.k_sh % 32 == 0 && p.k_sn % 32 == 0 && p.k_sb % 32 == 0 && p.v_sh % 32 == 0 && p.v_sn % 32 == 0 && p.v_sb % 32 == 0 &&
                   ((uintptr_t)p.k % 128) == 0 && ((uintptr_t)p.v % 128) == 0;
  return !no_fast && p.dtype == SIU3R_F32 && p.split3 && seg && siu3r_attn_pipe_ok(p);
}

extern "C" int siu3r_attention_kv_x3_ok(const siu3r_attn_params* pp) { return pp && kv_x3_ok(*pp) ? 1 : 0; }

extern "C" int siu3r_attention(const siu3r_attn_params* pp, void* stream) {
  const siu3r_attn_params& p = *pp;
  SIU3R_CHECK(p.q && p.k && p.v && p.out, "siu3r_attention: null pointer");
  SIU3R_CHECK(p.D == 64 || p.D == 32, "siu3r_attention: head_dim %d unsupported (32 or 64)", p.D);
  SIU3R_CHECK(p.B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "siu3r_attention: empty problem");
  SIU3R_CHECK(p.dtype == SIU3R_BF16 || p.dtype == SIU3R_F32, "siu3r_attention: bad dtype %d", p.dtype);
  SIU3R_CHECK(!(p.split3 && p.dtype != SIU3R_F32), "siu3r_attention: bf16x3 mode needs fp32 tensors");
  SIU3R_CHECK(!(p.mask && (p.mask_ld % 64 != 0 || p.mask_ld < p.Nk || ((uintptr_t)p.mask & 15) != 0)),
              "siu3r_attention: the key mask needs a 16-byte aligned buffer with row stride mask_ld %% 64 == 0 and >= Nk (mask_ld=%ld Nk=%d)", (long)p.mask_ld, p.Nk);
  SIU3R_CHECK(p.kv_bxor >= 0 && (p.kv_bxor & (p.kv_bxor + 1)) == 0 && p.B % (p.kv_bxor + 1) == 0 && !(p.kv_bxor && p.rope_cos),
              "siu3r_attention: kv_bxor=%d must be 2^n - 1 with B a multiple of 2^n, and excludes RoPE on load", p.kv_bxor);
  SIU3R_CHECK(!(p.rope_cos && p.D != 64), "siu3r_attention: RoPE2D path is specialised for head_dim 64");
  SIU3R_CHECK(!(p.rope_cos && !(p.rope_sin && p.qpos && p.kpos)), "siu3r_attention: rope tables/positions missing");
  const int64_t al = p.dtype == SIU3R_F32 ? 4 : 8;  // 16-byte vector loads
  SIU3R_CHECK(p.q_sn % al == 0 && p.q_sh % al == 0 && p.q_sb % al == 0 && p.k_sn % al == 0 && p.k_sh % al == 0 &&
                  p.k_sb % al == 0 && p.v_sn % al == 0 && p.v_sh % al == 0 && p.v_sb % al == 0,
              "siu3r_attention: q/k/v strides must keep 16-byte alignment");
  SIU3R_CHECK(!p.kv_x3 || kv_x3_ok(p), "siu3r_attention: kv_x3 (pre-split K / V planes) needs the pipelined bf16x3 kernel (head_dim 64, no mask / RoPE on load / "
              "key split, >= 64 keys) and segment-aligned K / V: query siu3r_attention_kv_x3_ok first");
  hipStream_t s = (hipStream_t)stream;
  static const bool no_fast = getenv("SIU3R_ATTN_NO_FAST") != nullptr;  // A/B switch
  if (!no_fast && siu3r_attn_pipe_ok(p)) return siu3r_attn_pipe_launch(p, s);
  if (p.dtype == SIU3R_BF16 && !p.split3 && !p.rope_cos && !no_fast) return p.D == 64 ? launch_fast<64, 0>(p, s) : launch_fast<32, 0>(p, s);
  if (p.dtype == SIU3R_F32 && p.split3 && !p.rope_cos && !no_fast) return p.D == 64 ? launch_fast<64, 1>(p, s) : launch_fast<32, 1>(p, s);
  if (p.D == 64) {
    if (p.split3) return launch<64, 1, 1>(p, s);
    if (p.dtype == SIU3R_F32) return launch<64, 1, 0>(p, s);
    return launch<64, 0, 0>(p, s);
  } else {
    if (p.split3) return launch<32, 1, 1>(p, s);
    if (p.dtype == SIU3R_F32) return launch<32, 1, 0>(p, s);
    return launch<32, 0, 0>(p, s);
  }
}
