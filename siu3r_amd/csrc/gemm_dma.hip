// bf16 fast-path GEMM / implicit-GEMM conv for gfx950: operands stream global -> LDS with the LDS-DMA
// (global_load_lds_dwordx4, 16 B/lane, 1 KiB per wave-instruction) into a 3-stage ring; MFMA reads stage kt while
// the DMAs of stages kt+1 and kt+2 are in flight.  The waits are hand-counted (s_waitcnt vmcnt(N), raw s_barrier):
// hipcc neither tracks LDS-DMA completion nor inserts waits for it, and -- unlike the register-staged kernel in
// gemm.hip, whose loads hipcc drains with vmcnt(0) before every LDS write -- nothing is drained early.
// The LDS image keeps gemm.hip's XOR swizzle: the DMA destination is lane-linear, so the permutation is applied
// to the per-lane SOURCE address (cdna_hip_programming.md rule 21).  Zero padding (conv borders, K tail) points the
// lane at a 16-byte zero page instead of branching.  Same tiles, MFMA layout and epilogue as gemm.hip.
#include "common.h"
#include "gemm_epilogue.h"

namespace siu3r_gemm_dma {

constexpr int BM = 128, BK = 64, STAGES = 3;
constexpr int A_TILE_BYTES = BM * BK * 2;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint4 relu_bf16x8(uint4 v) {
  uint32_t* w = (uint32_t*)&v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t neg = (w[j] >> 15) & 0x00010001u;  // sign bits -> bit 0 of each half
    w[j] &= ~(neg * 0xffffu);
  }
  return v;
}

template <int NI, int CONV>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const siu3r_gemm_params p) {
  constexpr int BN = 64 * NI;
  constexpr int B_TILE_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int A_DMA = 4;        // 1-KiB DMA pieces per wave per K-tile for A (16 pieces / 4 waves)
  constexpr int W_DMA = 2 * NI;   // ... for W (BN/8 pieces / 4 waves)
  constexpr int LP = A_DMA + W_DMA;
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * STAGE_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_m = (p.m + BM - 1) / BM, tiles_n = (p.n + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
  const int ry = xcd / p.map_gx, rx = xcd - ry * p.map_gx;
  const int lm = li / p.map_rn, ln = li - lm * p.map_rn;
  const int tile_m = ry * p.map_rm + lm, tile_n = rx * p.map_rn + ln;
  if (lm >= p.map_rm || tile_m >= tiles_m || tile_n >= tiles_n) return;
  const int z = blockIdx.z;

  const u16* Ab = (const u16*)p.a + (int64_t)z * p.sa;
  const u16* Wb = (const u16*)p.w_hi + (int64_t)z * p.sw;
  const u16* zero = (const u16*)g_zero_page;

  // ---- DMA geometry.  Piece g (8 rows x 128 B) of a tile; lane -> (row = 8 g + lane/8, physical chunk = lane%8);
  // the lane fetches the LOGICAL chunk that the swizzle stores there.
  const int prow = lane >> 3, pchunk = lane & 7;
  int a_c[A_DMA];                 // logical 8-element k-chunk fetched by this lane for piece i
  const u16* a_ptr[A_DMA];        // dense: row base pointer (+ chunk offset); conv: image base of the row's batch item
  int a_iy0[A_DMA], a_ix0[A_DMA];
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int r = (wave * A_DMA + i) * 8 + prow;
    a_c[i] = pchunk ^ ((r >> 1) & 7);
    int m = tile_m * BM + r;
    if (m > p.m - 1) m = p.m - 1;  // rows beyond M: clamped, never stored
    a_iy0[i] = a_ix0[i] = 0;
    if (CONV) {
      const int ohw = p.oh * p.ow;
      const int b = m / ohw, rr = m - b * ohw;
      const int oy = rr / p.ow, ox = rr - oy * p.ow;
      a_ptr[i] = Ab + (int64_t)b * p.ih * p.iw * p.cin;
      a_iy0[i] = oy * p.stride - p.pad;
      a_ix0[i] = ox * p.stride - p.pad;
    } else {
      a_ptr[i] = Ab + (int64_t)m * p.lda + a_c[i] * 8;
    }
  }
  const u16* w_ptr[W_DMA];
#pragma unroll
  for (int i = 0; i < W_DMA; ++i) {
    const int r = (wave * W_DMA + i) * 8 + prow;
    const int c = pchunk ^ ((r >> 1) & 7);
    int n = tile_n * BN + r;
    if (n > p.n - 1) n = p.n - 1;
    w_ptr[i] = Wb + (int64_t)n * p.kpad + c * 8;
  }

  // one A piece (index i) and/or W pieces of K-tile kt into ring stage `stage`
  auto issue_a = [&](int kt, int stage, int i) {
    unsigned char* sA = smem + stage * STAGE_BYTES;
    const u16* src;
    const int k0 = kt * BK + a_c[i] * 8;
    if (CONV) {
      const int tap = k0 / p.cin, c0 = k0 - tap * p.cin;
      const int ky = tap / p.kw, kx = tap - ky * p.kw;
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = k0 < p.k && iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw;
      src = ok ? a_ptr[i] + ((int64_t)iy * p.iw + ix) * p.cin + c0 : zero;
    } else {
      src = k0 < p.k ? a_ptr[i] + kt * BK : zero;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(sA + (wave * A_DMA + i) * 1024), 16, 0, 0);
  };
  auto issue_w = [&](int kt, int stage, int i) {
    unsigned char* sB = smem + stage * STAGE_BYTES + A_TILE_BYTES;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[i] + kt * BK),
                                     (__attribute__((address_space(3))) void*)(sB + (wave * W_DMA + i) * 1024), 16, 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) issue_a(kt, stage, i);
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) issue_w(kt, stage, i);
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragment reads are inline asm: hipcc treats every LDS-DMA as a pending LDS store and would put
  // s_waitcnt vmcnt(0) in front of any ds_read it can see, draining the ring each K-step.  The reads are issued in
  // two halves of (2 + NI) * 2 so that the second half's latency hides under the first half's MFMAs; each wait
  // statement names its fragments "+v" so no MFMA can be scheduled above it (cdna_hip_programming.md 5.7 (ii)).
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned int lds_base = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned int offA[2], offB[NI];  // per-lane row offsets; the k-substep enters through the XOR term
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = (wm * 64 + i * 32 + l31) * 128;
#pragma unroll
  for (int j = 0; j < NI; ++j) offB[j] = (wn * (32 * NI) + j * 32 + l31) * 128;
  const int swzA0 = ((wm * 64 + l31) >> 1) & 7;          // rows i*32 apart share (row>>1)&7
  const int swzB0 = ((wn * (32 * NI) + l31) >> 1) & 7;
  const unsigned int relu_mask = p.relu_in ? 0xffffffffu : 0u;

  // K-tile body: fragment reads of k-substep ks+1 are issued before the MFMAs of ks, and the DMA pieces of the
  // K-tile two ahead are spread over the four MFMA groups so that their issue cost hides under the matrix pipe.
  // Only lgkmcnt(0) is used for the reads (a counted wait could be satisfied by an unrelated scalar load).
  auto read_frags = [&](unsigned int sA, unsigned int sB, int ks, u32x4 (&fa)[2], u32x4 (&fb)[NI]) {
    const int c = ks * 2 + lh;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned int ad = sA + offA[i] + ((c ^ swzA0) << 4);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[i]) : "v"(ad) : "memory");
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned int ad = sB + offB[j] + ((c ^ swzB0) << 4);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[j]) : "v"(ad) : "memory");
    }
  };
  auto compute = [&](int stage, int kt_next, int stage_next) {
    const unsigned int sA = lds_base + stage * STAGE_BYTES;
    const unsigned int sB = sA + A_TILE_BYTES;
    u32x4 fa[4][2], fb[4][NI];
    read_frags(sA, sB, 0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (NI == 2)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks][0]), "+v"(fa[ks][1]), "+v"(fb[ks][0]), "+v"(fb[ks][NI - 1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[ks][0]), "+v"(fa[ks][1]), "+v"(fb[ks][0])::"memory");
      if (ks < 3) read_frags(sA, sB, ks + 1, fa[ks + 1], fb[ks + 1]);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a[2], bq[NI];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 v = fa[ks][i];
        // fused input ReLU (ResidualConvUnit): clear negative bf16 halves, branch-free
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int neg = (v[e] >> 15) & 0x00010001u;
          v[e] &= ~((neg * 0xffffu) & relu_mask);
        }
        union { u32x4 u; bf16x8 h; } cv;
        cv.u = v;
        a[i] = cv.h;
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        union { u32x4 u; bf16x8 h; } cv;
        cv.u = fb[ks][j];
        bq[j] = cv.h;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
      if (kt_next >= 0) {  // wave-uniform
        issue_a(kt_next, stage_next, ks);
        if (NI == 2) issue_w(kt_next, stage_next, ks);
        else if (ks < 2) issue_w(kt_next, stage_next, ks);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- 3-stage ring, prefetch distance 2, one raw barrier per K-tile, counted vmcnt
  const int nkt = p.kpad / BK;
  issue(0, 0);
  if (nkt > 1) issue(1, 1);
  int st = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt must have landed; only tile kt+1 (the LP newest DMAs of this wave) may still be in flight
    if (kt + 1 < nkt) {
      if (LP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile kt are in LDS; stage (kt+2)%3 is no longer read
    asm volatile("" ::: "memory");
    int s2 = st + 2;
    if (s2 >= STAGES) s2 -= STAGES;
    compute(st, kt + 2 < nkt ? kt + 2 : -1, s2);
    st = (st + 1 == STAGES) ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: LDS-staged, row-wise vectorised (gemm_epilogue.h)
  siu3r_epi::run<NI>(p, acc, smem, tile_m, tile_n, z, t);
}

}  // namespace siu3r_gemm_dma

// called by siu3r_gemm() (gemm.hip) for bf16 activations without the bf16x3 split, dense or conv gather
int siu3r_gemm_dma_launch(const siu3r_gemm_params& p, int ni, void* stream) {
  using namespace siu3r_gemm_dma;
  const int BN = 64 * ni;
  const int tiles = 8 * p.map_rm * p.map_rn;
  (void)BN;
  dim3 grid(tiles, 1, p.batch > 0 ? p.batch : 1), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (ni == 1) {
    if (p.a_mode == 1)
      hipLaunchKernelGGL((gemm_dma_kernel<1, 1>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((gemm_dma_kernel<1, 0>), grid, block, 0, s, p);
  } else {
    if (p.a_mode == 1)
      hipLaunchKernelGGL((gemm_dma_kernel<2, 1>), grid, block, 0, s, p);
    else
      hipLaunchKernelGGL((gemm_dma_kernel<2, 0>), grid, block, 0, s, p);
  }
  SIU3R_LAUNCH_CHECK("siu3r_gemm(dma)");
  return 0;
}
