// bf16 fast-path GEMM / implicit-GEMM conv for gfx950: operands stream global -> LDS with the LDS-DMA
// (buffer_load_dwordx4 ... lds, 16 B/lane, 1 KiB per wave-instruction) into a 3-stage ring; MFMA reads stage kt while
// the DMAs of stages kt+1 and kt+2 are in flight.  The waits are hand-counted (s_waitcnt vmcnt(N), raw s_barrier):
// hipcc neither tracks LDS-DMA completion nor inserts waits for it, and -- unlike the register-staged kernel in
// gemm.hip, whose loads hipcc drains with vmcnt(0) before every LDS write -- nothing is drained early.
//
// The K loop is budgeted in VALU slots: 16 MFMAs of a 128x128x64 step keep the matrix pipe busy for 512 cycles, i.e.
// 128 wave64 VALU issues; everything else has to fit under that.  Hence
//   * buffer addressing: per-lane byte offsets are loop-invariant VGPRs, the K advance travels in the scalar offset,
//     and padding (conv borders, K tail) is an out-of-range offset that the hardware turns into zeros -- no pointer
//     arithmetic, selects or divisions in the loop;
//   * MODE 1 (3x3/1x1 convs with cin % 64 == 0): one filter tap per K tile, tracked by a scalar cursor, border validity
//     as one bit test against a per-row tap mask;
//   * the fused input ReLU is one v_pk_max_i16 per fragment dword and compiled in only where it is used.
// MODE 2 (any cin % 8 == 0, e.g. the 7x7 stem): per-lane tap state for the lane's two chunk columns, advanced without divisions.
// The LDS image keeps gemm.hip's XOR swizzle: the DMA destination is lane-linear, so the permutation is applied
// to the per-lane SOURCE address (cdna_hip_programming.md rule 21).  Same tiles, MFMA layout and epilogue as gemm.hip.
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

namespace siu3r_gemm_dma {

constexpr int BM = 128, BK = 64, STAGES = 3;
#ifndef SIU3R_GEMM_PD_WIDE
#define SIU3R_GEMM_PD_WIDE 2  // prefetch distance (K tiles in flight) of the 128 x 128 kernels; 3 and 4 (= 5 ring stages of 32 KiB, the whole LDS) measured no faster
#endif
constexpr int A_TILE_BYTES = BM * BK * 2;
constexpr unsigned OOB = 0xffffff00u;  // >= num_records of every resource below: the DMA writes zeros
constexpr int RSRC_FLAGS = 0x00020000;

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) short i16x2;

// KSPL = 2 (NI = 1 only): eight waves per workgroup; waves 4..7 work on the SAME 64x32 sub-tiles as waves 0..3 but on the second
// half of every K tile (k-substeps 2,3), and the two partial accumulators are summed in the epilogue.  A single wave can issue
// one ds_read_b128 per ~31 cycles and one LDS-DMA piece per ~55 whatever the rest of the CU does (tools/probes/lds_bw_probe.hip),
// so a K step costs a wave 12 x 31 + 6 x 55 cycles of issue against 256 cycles of MFMA; splitting the K tile over twice the
// waves halves that per wave and lets the CU's LDS (~220 B/clk from 8+ waves) rather than one wave's issue rate set the pace.
// (second launch bound = waves per SIMD: the 128 x 64 kernels must keep two workgroups per CU, i.e. <= 128 VGPRs)
template <int NI, int MODE, bool RELU, int KSPL, bool LNF = false>
__global__ __launch_bounds__(256 * KSPL, NI == 1 ? 2 * KSPL : 1) void gemm_dma_kernel(const siu3r_gemm_params p) {
#if __HIP_DEVICE_COMPILE__  // the buffer-resource type has no host representation; the host pass only needs the stub
  constexpr int BN = 64 * NI;
  constexpr int B_TILE_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int A_DMA = 4 / KSPL;        // 1-KiB DMA pieces per wave per K-tile for A (16 pieces / 4 KSPL waves)
  constexpr int W_DMA = 2 * NI / KSPL;   // ... for W (BN/8 pieces)
  constexpr int LP = A_DMA + W_DMA;
  constexpr int KSW = 4 / KSPL;          // k-substeps (of 16) of a K tile handled by one wave
  // ring depth: the L2 -> LDS stream of a CU is latency-bound (~1.2 us issue -> landed), its rate is set by the bytes in flight.  Two
  // 128 x 64 workgroups per CU hold 2 x 2 tiles of 24 KiB in flight (3 stages each); the single 128 x 128 workgroup gets 5 stages
  // and keeps 4 tiles of 32 KiB in flight
  constexpr int PD = (NI == 2) ? SIU3R_GEMM_PD_WIDE : 2;
  constexpr int STAGES = PD + 1;
  static_assert(STAGES * STAGE_BYTES >= siu3r_epi::staging_bytes<NI, KSPL>(), "epilogue staging must fit the ring");
  static_assert(STAGES * STAGE_BYTES <= 160 * 1024, "ring exceeds the LDS");
  __shared__ __attribute__((aligned(128))) unsigned char smem[STAGES * STAGE_BYTES];
  // s_waitcnt vmcnt(tiles * LP): at most `tiles` younger K tiles' pieces of this wave still in flight
  auto wait_tiles_in_flight = [&](int tiles) {
    static_assert(PD <= 4 && LP * 3 <= 63, "vmcnt immediates");
    const int n = tiles * LP;
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define SIU3R_VM_CASE(V) else if (n == V) asm volatile("s_waitcnt vmcnt(" #V ")" ::: "memory");
    SIU3R_VM_CASE(3) SIU3R_VM_CASE(4) SIU3R_VM_CASE(6) SIU3R_VM_CASE(8) SIU3R_VM_CASE(9) SIU3R_VM_CASE(12) SIU3R_VM_CASE(16) SIU3R_VM_CASE(18) SIU3R_VM_CASE(24)
#undef SIU3R_VM_CASE
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // 0 .. 4 KSPL - 1
  const int khalf = wave >> 2, wq = wave & 3;                // K half of this wave, position in the 2x2 sub-tile grid
  const int wm = wq >> 1, wn = wq & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int M = p.m, N = p.n, K = p.k, kpad = p.kpad;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int xcd = (blockIdx.x + blockIdx.z) & 7, li = blockIdx.x >> 3;  // (regions rotate with the batch item: few tiles per item x many items still load all XCDs)
#ifndef SIU3R_GEMM_DBG
#define SIU3R_GEMM_DBG 0  // tuning builds only (tools/ab_build.sh): 1 no epilogue, 2 no in-loop DMA, 4 no MFMA, 8 no barrier, 16 no reads
#endif
  constexpr int dbg = SIU3R_GEMM_DBG;
  const int map_gx = p.map_gx;
  const int ry = xcd / map_gx, rx = xcd - ry * map_gx;
  const int lm = li / p.map_rn, ln = li - lm * p.map_rn;
  const int tile_m = ry * p.map_rm + lm, tile_n = rx * p.map_rn + ln;
  if (lm >= p.map_rm || tile_m >= tiles_m || tile_n >= tiles_n) return;
  const int z = blockIdx.z;
  // split-K: this workgroup multiplies K tiles [kbase, kbase + nkt) of the kpad / BK tiles
  const int nkt_all = kpad / BK;
  const int kbase = p.splitk > 1 ? (int)((int64_t)blockIdx.y * nkt_all / p.splitk) : 0;
  const int nkt = p.splitk > 1 ? (int)((int64_t)(blockIdx.y + 1) * nkt_all / p.splitk) - kbase : nkt_all;
  uint64_t* trace = p.trace ? p.trace + (size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 : nullptr;
  auto stamp = [&](int slot) {
    if (trace && t == 0) trace[slot] = __builtin_readcyclecounter();
  };
  stamp(0);

  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const u16* Ab = (const u16*)p.a + zof.a;
  const u16* Wb = (const u16*)p.w_hi + zof.w;

  // ---- DMA geometry.  Piece g (8 rows x 128 B) of a tile; lane -> (row = 8 g + lane/8, physical chunk = lane%8);
  // the lane fetches the LOGICAL chunk that the swizzle stores there.
  const int prow = lane >> 3, pchunk = lane & 7;
  const int cin = p.cin, iw = p.iw, ih = p.ih, kw = p.kw, kh = p.kh;
  const int pad_bias = (MODE != 0) ? (p.pad * iw + p.pad) * cin * 2 : 0;  // keeps every per-row offset non-negative
  __amdgpu_buffer_rsrc_t rA, rW;
  if (MODE == 0)
    rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, (short)0, (int)(((int64_t)(M - 1) * p.lda + K) * 2), RSRC_FLAGS);
  else
    rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ab - pad_bias), (short)0, (int)OOB, RSRC_FLAGS);
  rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, (short)0, (int)((int64_t)N * kpad * 2), RSRC_FLAGS);

  unsigned a_voff[A_DMA];          // byte offset of the lane's chunk at K-tile 0 / of the row's (iy0, ix0) pixel (conv, biased)
  unsigned a_mask[A_DMA];          // MODE 1: bit (ky*kw+kx) set <=> that tap is inside the image for this row
  int a_c[A_DMA];                  // logical 8-element k-chunk fetched by this lane for piece i (a_c[i] == a_c[i & 1])
  int a_iy0[A_DMA], a_ix0[A_DMA];  // MODE 2: top-left input pixel of the row
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int r = (wave * A_DMA + i) * 8 + prow;
    a_c[i] = pchunk ^ ((r >> 1) & 7);
    int m = tile_m * BM + r;
    // rows beyond M are never stored: they fetch row M - 1 again (one cached line per K tile).  Out-of-range offsets ("zeros, no
    // traffic") looked cheaper and are not: a piece whose lanes are out of range lands far later than a real one, and M = 2 x 1025
    // tokens leaves a 2-row seventeenth tile row in every encoder GEMM (bf16x3 2050 x 1024 x 4096: 249 us with out-of-range rows,
    // 107 us clamped; bf16 2050 x 3072 x 1024: 31.9 -> 27.7 us)
    const bool row_ok = m < M;
    if (!row_ok) m = M - 1;
    a_voff[i] = a_mask[i] = 0;
    a_iy0[i] = a_ix0[i] = 0;
    if (MODE == 0) {
      a_voff[i] = (unsigned)(((int64_t)m * p.lda + a_c[i] * 8) * 2);
    } else {
      const int ohw = p.oh * p.ow;
      const int b = m / ohw, rr = m - b * ohw;
      const int oy = rr / p.ow, ox = rr - oy * p.ow;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      if (MODE == 1) {
        a_voff[i] = (unsigned)((((b * ih + iy0) * iw + ix0) * cin + a_c[i] * 8) * 2 + pad_bias);
        unsigned mk = 0;
        for (int ky = 0; ky < kh; ++ky)
          for (int kx = 0; kx < kw; ++kx)
            if (iy0 + ky >= 0 && iy0 + ky < ih && ix0 + kx >= 0 && ix0 + kx < iw) mk |= 1u << (ky * kw + kx);
        a_mask[i] = row_ok ? mk : 0u;
      } else {
        a_voff[i] = (unsigned)((((b * ih + iy0) * iw + ix0) * cin) * 2 + pad_bias);
        a_iy0[i] = row_ok ? iy0 : -(1 << 20);  // fails every bounds test
        a_ix0[i] = ix0;
      }
    }
  }
  unsigned w_voff[W_DMA];
#pragma unroll
  for (int i = 0; i < W_DMA; ++i) {
    const int r = (wave * W_DMA + i) * 8 + prow;
    const int c = pchunk ^ ((r >> 1) & 7);
    int n = tile_n * BN + r;
    if (n > N - 1) n = N - 1;
    w_voff[i] = (unsigned)(((int64_t)n * kpad + c * 8) * 2);
  }
  const bool ktail = (K & (BK - 1)) != 0;

  // MODE 1 cursor of the next K tile to be issued (tiles are issued strictly in order): wave-uniform scalars
  int cur_c0 = 0, cur_kx = 0, cur_toff = 0, cur_row = 0;  // cur_row = byte offset of tap row ky
  unsigned cur_bit = 1u;
  auto cursor_advance = [&]() {
    cur_c0 += BK;
    cur_toff += BK * 2;
    if (cur_c0 == cin) {
      cur_c0 = 0;
      cur_bit <<= 1;
      if (++cur_kx == kw) {
        cur_kx = 0;
        cur_row += iw * cin * 2;
      }
      cur_toff = cur_row + cur_kx * cin * 2;
    }
  };

  // MODE 2 (any cin % 8 == 0, e.g. the 7x7 stem with cin = 8): a lane fetches only two chunk columns of the K tile (a_c[0] for
  // its even pieces, a_c[1] for the odd ones); for each it tracks (channel offset, kx, ky) of the NEXT tile to issue and advances
  // them by 64 K elements with two conditional wraps -- no per-piece divisions (they cost ~300 VALU per K tile and wave)
  int s_c0[2] = {0, 0}, s_kx[2] = {0, 0}, s_ky[2] = {0, 0};
  int q64_r = 0, q64_qx = 0, q64_qy = 0;
  if (MODE == 2) {
    const int q64 = BK / cin;
    q64_r = BK - q64 * cin;
    q64_qy = q64 / kw;
    q64_qx = q64 - q64_qy * kw;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int k0 = a_c[s_] * 8;
      const int tap = k0 / cin;
      s_c0[s_] = k0 - tap * cin;
      s_ky[s_] = tap / kw;
      s_kx[s_] = tap - s_ky[s_] * kw;
    }
  }
  auto state_advance = [&]() {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      s_c0[s_] += q64_r;
      const int carry = s_c0[s_] >= cin ? 1 : 0;
      s_c0[s_] -= carry * cin;
      s_kx[s_] += q64_qx + carry;
      const int wrap = s_kx[s_] >= kw ? 1 : 0;
      s_kx[s_] -= wrap * kw;
      s_ky[s_] += q64_qy + wrap;
    }
  };

  // one A piece (index i) and/or W pieces of K-tile kt into ring stage `stage`
  auto issue_a = [&](int kt, int stage, int i) {
    unsigned char* dst = smem + stage * STAGE_BYTES + (wave * A_DMA + i) * 1024;
    if (MODE == 0) {
      unsigned voff = a_voff[i];
      if (ktail && kbase + kt == nkt_all - 1) voff = ((kbase + kt) * BK + a_c[i] * 8 < K) ? voff : OOB;  // (an out-of-range row stays out of range)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, (kbase + kt) * (BK * 2), 0, 0);
    } else if (MODE == 1) {
      const unsigned voff = (a_mask[i] & cur_bit) ? a_voff[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, cur_toff, 0, 0);
    } else {
      const int s_ = i & 1;
      const int ky = s_ky[s_], kx = s_kx[s_];
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      // ky >= kh <=> the chunk lies in the zero-padded K tail
      const bool ok = ky < kh && (unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw;
      const unsigned voff = ok ? a_voff[i] + (unsigned)(((ky * iw + kx) * cin + s_c0[s_]) * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, 0, 0, 0);
    }
  };
  auto issue_w = [&](int kt, int stage, int i) {
    unsigned char* dst = smem + stage * STAGE_BYTES + A_TILE_BYTES + (wave * W_DMA + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr)dst, 16, w_voff[i], (kbase + kt) * (BK * 2), 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) issue_a(kt, stage, i);
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) issue_w(kt, stage, i);
    if (MODE == 1) cursor_advance();
    if (MODE == 2) state_advance();
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragment reads are inline asm: hipcc treats every LDS-DMA as a pending LDS store and would put
  // s_waitcnt vmcnt(0) in front of any ds_read it can see, draining the ring each K-step.  Each wait statement names
  // its fragments "+v" so no MFMA can be scheduled above it (cdna_hip_programming.md 5.7 (ii)).
  // Address of (row, k-substep ks): row*128 + (((2 ks + lh) ^ swz) << 4) = (row*128 | ((lh ^ swz) << 4)) ^ (ks << 5).
  const unsigned int lds_base = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned int offA[2], offB[NI];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + l31;
    offA[i] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = wn * (32 * NI) + j * 32 + l31;
    offB[j] = A_TILE_BYTES + row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
  }

  auto read_frags = [&](const unsigned int (&adA)[2], const unsigned int (&adB)[NI], int ks, u32x4 (&fa)[2], u32x4 (&fb)[NI]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned int ad = adA[i] ^ (ks << 5);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[i]) : "v"(ad) : "memory");
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const unsigned int ad = adB[j] ^ (ks << 5);
      asm volatile("ds_read_b128 %0, %1" : "=v"(fb[j]) : "v"(ad) : "memory");
    }
  };
  // K-tile body.  All 4 x (2 + NI) fragment reads of a K tile are issued back to back, one MFMA group BEFORE the
  // tile is consumed: inside the last k-substep of the previous tile, right after the barrier that publishes it.
  // A ds_read_b128 round trip is ~180 cycles with four waves reading; a k-substep's MFMAs last 64 (NI=1) / 128 cycles,
  // so reading one substep ahead left most of that latency exposed four times per K tile.  The waits are counted
  // (LDS returns in order; an interleaved scalar load can only make a counted wait more conservative).
  auto issue_tile_reads = [&](int stage, u32x4 (&fa)[4][2], u32x4 (&fb)[4][NI]) {
    const unsigned int sbase = lds_base + stage * STAGE_BYTES;  // 128-byte aligned: the XOR stays inside the row
    unsigned int adA[2], adB[NI];
#pragma unroll
    for (int i = 0; i < 2; ++i) adA[i] = sbase + offA[i];
#pragma unroll
    for (int j = 0; j < NI; ++j) adB[j] = sbase + offB[j];
#ifdef SIU3R_GEMM_RDLINEAR  // tuning build: conflict-free lane-linear addresses (wrong data) = the LDS read floor
#pragma unroll
    for (int i = 0; i < 2; ++i) adA[i] = sbase + (wave * 2 + i) * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < NI; ++j) adB[j] = sbase + (wave * NI + j) * 4096 + lane * 16 + 128;
#endif
    // lgkmcnt is a 4-bit counter: never more than 15 reads in flight.  NI = 2 (16 reads per tile) issues its last
    // k-substep from tile_body, after the first wait
#pragma unroll
    for (int j = 0; j < ((NI == 2 && KSPL == 1) ? 3 : KSW); ++j) read_frags(adA, adB, khalf * KSW + j, fa[j], fb[j]);
  };
  auto issue_last_reads = [&](int stage, u32x4 (&fa)[4][2], u32x4 (&fb)[4][NI]) {
    const unsigned int sbase = lds_base + stage * STAGE_BYTES;
    unsigned int adA[2], adB[NI];
#pragma unroll
    for (int i = 0; i < 2; ++i) adA[i] = sbase + offA[i];
#pragma unroll
    for (int j = 0; j < NI; ++j) adB[j] = sbase + offB[j];
    read_frags(adA, adB, 3, fa[3], fb[3]);
  };
#define SIU3R_WAIT_FRAGS(CNT)                                                                                              \
  do {                                                                                                                     \
    if (NI == 2)                                                                                                           \
      asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fb[j][0]), "+v"(fb[j][NI - 1])::"memory"); \
    else                                                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(fa[j][0]), "+v"(fa[j][1]), "+v"(fb[j][0])::"memory");          \
  } while (0)
  auto tile_body = [&](int kt, int st, u32x4 (&fa)[4][2], u32x4 (&fb)[4][NI], u32x4 (&na)[4][2], u32x4 (&nb)[4][NI]) {
    const int kt_next = (kt + PD < nkt && !(dbg & 2)) ? kt + PD : -1;  // wave-uniform
    int stage_next = st + PD;
    if (stage_next >= STAGES) stage_next -= STAGES;
    int stage_read = st + 1;
    if (stage_read >= STAGES) stage_read -= STAGES;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      // fragments of this wave's j-th k-substep: (KSW - 1 - j) * (2 + NI) younger reads may still be in flight
      if (KSPL == 2) {
        if (j == 1) SIU3R_WAIT_FRAGS(0);
        else if (NI == 2) SIU3R_WAIT_FRAGS(4);
        else SIU3R_WAIT_FRAGS(3);
      } else if (j == 0) {
        if (NI == 2) {
          SIU3R_WAIT_FRAGS(8);
          issue_last_reads(st, fa, fb);
        } else {
          SIU3R_WAIT_FRAGS(9);
        }
      }
      else if (j == 1) { if (NI == 2) SIU3R_WAIT_FRAGS(8); else SIU3R_WAIT_FRAGS(6); }
      else if (j == 2) { if (NI == 2) SIU3R_WAIT_FRAGS(4); else SIU3R_WAIT_FRAGS(3); }
      else SIU3R_WAIT_FRAGS(0);
      if (j == KSW - 1 && kt + 1 < nkt) {
        // this wave has finished reading tile kt.  Tile kt+1 must have landed: only the pieces of tiles kt+2 .. kt+PD (the last one
        // issued in this tile's first group) may remain in flight
        {
          int younger = nkt - 2 - kt;  // issued tiles behind kt+1
          if (dbg & 2) younger = 0;
          wait_tiles_in_flight(younger < PD - 1 ? younger : PD - 1);
        }
        if (!(dbg & 8)) __builtin_amdgcn_s_barrier();  // tile kt+1 is visible to every wave; stage st is free for tile kt+PD+1
        asm volatile("" ::: "memory");
        if (!(dbg & 16)) issue_tile_reads(stage_read, na, nb);
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a[2], bq[NI];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        u32x4 v = fa[j][i];
        if (RELU) {  // fused input ReLU (ResidualConvUnit): a negative bf16 is a negative int16
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            union { unsigned int u; i16x2 h; } x;
            x.u = v[e];
            x.h = __builtin_elementwise_max(x.h, (i16x2){0, 0});
            v[e] = x.u;
          }
        }
        union { u32x4 u; bf16x8 h; } cv;
        cv.u = v;
        a[i] = cv.h;
      }
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) {
        union { u32x4 u; bf16x8 h; } cv;
        cv.u = fb[j][jj];
        bq[jj] = cv.h;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) if (!(dbg & 4)) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[jj], acc[i][jj], 0, 0, 0);
      // all pieces of tile kt+2 go out behind the first MFMA group: they then have almost two K tiles to land (spreading them
      // over the groups measured 2-5 % slower: the late pieces had barely one)
      if (kt_next >= 0 && j == 0) {
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) issue_a(kt_next, stage_next, i);
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) issue_w(kt_next, stage_next, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 && kt_next >= 0) cursor_advance();
    if (MODE == 2 && kt_next >= 0) state_advance();
  };

  // ---- 3-stage ring, prefetch distance 2, one raw barrier per K-tile, counted vmcnt
  stamp(1);
  for (int q = 0; q < kbase; ++q) {  // split-K: tap cursor / per-lane tap state of this slice's first K tile
    if (MODE == 1) cursor_advance();
    if (MODE == 2) state_advance();
  }
#pragma unroll
  for (int q = 0; q < PD; ++q)
    if (q < nkt) issue(q, q);
  stamp(2);
  wait_tiles_in_flight((nkt < PD ? nkt : PD) - 1);
  __builtin_amdgcn_s_barrier();  // tile 0 is in LDS
  asm volatile("" ::: "memory");
  stamp(3);
  u32x4 fA0[4][2], fB0[4][NI], fA1[4][2], fB1[4][NI];
  issue_tile_reads(0, fA0, fB0);
  int st = 0;
  for (int kt = 0; kt < nkt; kt += 2) {
    tile_body(kt, st, fA0, fB0, fA1, fB1);
    st = (st + 1 == STAGES) ? 0 : st + 1;
    if (kt + 1 < nkt) {
      tile_body(kt + 1, st, fA1, fB1, fA0, fB0);
      st = (st + 1 == STAGES) ? 0 : st + 1;
    }
  }
#undef SIU3R_WAIT_FRAGS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(4);

  // ---- epilogue: LDS-staged, row-wise vectorised (gemm_epilogue.h)
  if (dbg & 1) { if (acc[0][0][0] == 123.456f) ((float*)p.c)[0] = 1.f; return; }
  siu3r_epi::run<NI, KSPL, LNF>(p, acc, smem, tile_m, tile_n, z, t);
  if (trace && t == 0) {
    trace[6] = __builtin_readcyclecounter();  // last store issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace[7] = __builtin_readcyclecounter();  // stores acknowledged
  }
#endif
}


// ---- bf16x3 on the LDS-DMA path -------------------------------------------------------------------------------------------------
// The 1e-3 parity mode multiplies fp32 activations: C = A_hi W_hi + A_lo W_hi + A_hi W_lo with A = A_hi + A_lo split into two bf16
// halves (16 mantissa bits survive).  Here A travels as fp32 straight from HBM into LDS (same LDS-DMA ring, same 128-byte rows: a K
// tile is 32 fp32 = 128 B) and is split in REGISTERS after the fragment read (hi = the upper 16 bits, lo = bf16(a - hi): two ands,
// two subtractions, one byte permute and one v_cvt_pk_bf16_f32 per pair -- ~24 VALU per 8-element fragment, issued under the
// 3 MFMAs it feeds); W is pre-split and pre-INTERLEAVED at load time as [n][k / 32][hi 32 | lo 32], so that one 128-byte row of the
// W tile carries both planes of a K tile and the tile / piece / swizzle geometry is byte for byte that of the bf16 kernel.
// Per K tile and workgroup the ring moves the same 24 KiB as the bf16 kernel does per 64-deep tile, for 3/2 the MFMA work at half
// the depth: the mode costs about two bf16 launches instead of the 3.2 of the register-staged kernel (gemm.hip), and activations
// need no second representation in HBM.  NI = 1 (128 x 64 tiles); MODE 0 (dense), MODE 1 (conv, cin % 32 == 0) and MODE 2 (conv, small cin % 4 == 0).
template <int MODE, bool RELU, int KSPL, bool LNF = false>
__global__ __launch_bounds__(256 * KSPL, 2 * KSPL) void gemm_dma_x3_kernel(const siu3r_gemm_params p) {
#if __HIP_DEVICE_COMPILE__
  constexpr int BN = 64, BKE = 32;
  constexpr int B_TILE_BYTES = BN * 128;
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int A_DMA = 4 / KSPL, W_DMA = 2 / KSPL, LP = A_DMA + W_DMA;
  constexpr int KSW = 2 / KSPL;  // k-substeps (of 16) of a K tile handled by one wave
  static_assert(STAGES * STAGE_BYTES >= siu3r_epi::staging_bytes<1, KSPL>(), "epilogue staging must fit the ring");
  __shared__ __attribute__((aligned(128))) unsigned char smem[STAGES * STAGE_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int khalf = wave >> 2, wq = wave & 3;
  const int wm = wq >> 1, wn = wq & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int M = p.m, N = p.n, K = p.k, kpad = p.kpad;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int xcd = (blockIdx.x + blockIdx.z) & 7, li = blockIdx.x >> 3;  // (regions rotate with the batch item: few tiles per item x many items still load all XCDs)
  const int map_gx = p.map_gx;
  const int ry = xcd / map_gx, rx = xcd - ry * map_gx;
  const int lm = li / p.map_rn, ln = li - lm * p.map_rn;
  const int tile_m = ry * p.map_rm + lm, tile_n = rx * p.map_rn + ln;
  if (lm >= p.map_rm || tile_m >= tiles_m || tile_n >= tiles_n) return;
  const int z = blockIdx.z;
  const int nkt_all = kpad / BKE;
  const int kbase = p.splitk > 1 ? (int)((int64_t)blockIdx.y * nkt_all / p.splitk) : 0;
  const int nkt = p.splitk > 1 ? (int)((int64_t)(blockIdx.y + 1) * nkt_all / p.splitk) - kbase : nkt_all;

  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const float* Ab = (const float*)p.a + zof.a;
  const u16* Wb = (const u16*)p.w_x3 + zof.w * 2;  // interleaved planes: 2 * kpad bf16 per row

  const int prow = lane >> 3, pchunk = lane & 7;
  const int cin = p.cin, iw = p.iw, ih = p.ih, kw = p.kw, kh = p.kh;
  const int pad_bias = (MODE != 0) ? (p.pad * iw + p.pad) * cin * 4 : 0;
  __amdgpu_buffer_rsrc_t rA, rW;
  if (MODE == 0)
    rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, (short)0, (int)(((int64_t)(M - 1) * p.lda + K) * 4), RSRC_FLAGS);
  else
    rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ab - pad_bias), (short)0, (int)OOB, RSRC_FLAGS);
  rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, (short)0, (int)((int64_t)N * kpad * 4), RSRC_FLAGS);

  unsigned a_voff[A_DMA], a_mask[A_DMA];
  int a_c[A_DMA];
  int a_iy0[A_DMA], a_ix0[A_DMA];  // MODE 2: top-left input pixel of the row
#pragma unroll
  for (int i = 0; i < A_DMA; ++i) {
    const int r = (wave * A_DMA + i) * 8 + prow;
    a_c[i] = pchunk ^ ((r >> 1) & 7);  // logical 16-byte chunk (4 fp32) of the row's K-tile slice
    int m = tile_m * BM + r;
    const bool row_ok = m < M;
    if (!row_ok) m = M - 1;
    a_voff[i] = a_mask[i] = 0;
    a_iy0[i] = a_ix0[i] = 0;
    if (MODE == 0) {
      a_voff[i] = (unsigned)(((int64_t)m * p.lda) * 4 + a_c[i] * 16);  // (row clamped to M - 1, see the bf16 kernel)
    } else {
      const int ohw = p.oh * p.ow;
      const int b = m / ohw, rr = m - b * ohw;
      const int oy = rr / p.ow, ox = rr - oy * p.ow;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      if (MODE == 1) {
        a_voff[i] = (unsigned)((((b * ih + iy0) * iw + ix0) * cin) * 4 + a_c[i] * 16 + pad_bias);
        unsigned mk = 0;  // (a zero-padded K tail lands on tap kh*kw, whose bit is never set)
        for (int ky = 0; ky < kh; ++ky)
          for (int kx = 0; kx < kw; ++kx)
            if (iy0 + ky >= 0 && iy0 + ky < ih && ix0 + kx >= 0 && ix0 + kx < iw) mk |= 1u << (ky * kw + kx);
        a_mask[i] = row_ok ? mk : 0u;
      } else {
        a_voff[i] = (unsigned)((((b * ih + iy0) * iw + ix0) * cin) * 4 + pad_bias);
        a_iy0[i] = row_ok ? iy0 : -(1 << 20);  // fails every bounds test
        a_ix0[i] = ix0;
      }
    }
  }
  unsigned w_voff[W_DMA];
#pragma unroll
  for (int i = 0; i < W_DMA; ++i) {
    const int r = (wave * W_DMA + i) * 8 + prow;
    const int c = pchunk ^ ((r >> 1) & 7);
    int n = tile_n * BN + r;
    if (n > N - 1) n = N - 1;
    w_voff[i] = (unsigned)(((int64_t)n * kpad) * 4 + c * 16);
  }
  const bool ktail = (K & (BKE - 1)) != 0;

  int cur_c0 = 0, cur_kx = 0, cur_toff = 0, cur_row = 0;
  unsigned cur_bit = 1u;
  auto cursor_advance = [&]() {
    cur_c0 += BKE;
    cur_toff += BKE * 4;
    if (cur_c0 == cin) {
      cur_c0 = 0;
      cur_bit <<= 1;
      if (++cur_kx == kw) {
        cur_kx = 0;
        cur_row += iw * cin * 4;
      }
      cur_toff = cur_row + cur_kx * cin * 4;
    }
  };
  // MODE 2 (any cin % 4 == 0, e.g. the 7x7 stem with cin = 8): as in the bf16 kernel, a lane fetches two chunk columns of the K tile
  // (a_c[0] for its even pieces, a_c[1] for the odd ones) and tracks (channel offset, kx, ky) of the next tile to issue for each,
  // advanced by 32 K elements with two conditional wraps
  int s_c0[2] = {0, 0}, s_kx[2] = {0, 0}, s_ky[2] = {0, 0};
  int q32_r = 0, q32_qx = 0, q32_qy = 0;
  if (MODE == 2) {
    const int q32 = BKE / cin;
    q32_r = BKE - q32 * cin;
    q32_qy = q32 / kw;
    q32_qx = q32 - q32_qy * kw;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const int k0 = a_c[s_] * 4;
      const int tap = k0 / cin;
      s_c0[s_] = k0 - tap * cin;
      s_ky[s_] = tap / kw;
      s_kx[s_] = tap - s_ky[s_] * kw;
    }
  }
  auto state_advance = [&]() {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      s_c0[s_] += q32_r;
      const int carry = s_c0[s_] >= cin ? 1 : 0;
      s_c0[s_] -= carry * cin;
      s_kx[s_] += q32_qx + carry;
      const int wrap = s_kx[s_] >= kw ? 1 : 0;
      s_kx[s_] -= wrap * kw;
      s_ky[s_] += q32_qy + wrap;
    }
  };
  auto issue_a = [&](int kt, int stage, int i) {
    unsigned char* dst = smem + stage * STAGE_BYTES + (wave * A_DMA + i) * 1024;
    if (MODE == 2) {
      const int s_ = i & 1;
      const int ky = s_ky[s_], kx = s_kx[s_];
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = ky < kh && (unsigned)iy < (unsigned)ih && (unsigned)ix < (unsigned)iw;  // ky >= kh: zero-padded K tail
      const unsigned voff = ok ? a_voff[i] + (unsigned)(((ky * iw + kx) * cin + s_c0[s_]) * 4) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, 0, 0, 0);
    } else if (MODE == 0) {
      unsigned voff = a_voff[i];
      if (ktail && kbase + kt >= nkt_all - 2) voff = ((kbase + kt) * BKE + a_c[i] * 4 < K) ? voff : OOB;  // (kpad is a multiple of 64: the tail spans two tiles)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, (kbase + kt) * (BKE * 4), 0, 0);
    } else {
      const unsigned voff = (a_mask[i] & cur_bit) ? a_voff[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, voff, cur_toff, 0, 0);
    }
  };
  auto issue_w = [&](int kt, int stage, int i) {
    unsigned char* dst = smem + stage * STAGE_BYTES + A_TILE_BYTES + (wave * W_DMA + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr)dst, 16, w_voff[i], (kbase + kt) * 128, 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < A_DMA; ++i) issue_a(kt, stage, i);
#pragma unroll
    for (int i = 0; i < W_DMA; ++i) issue_w(kt, stage, i);
    if (MODE == 1) cursor_advance();
    if (MODE == 2) state_advance();
  };

  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  // A row r, k-substep ks (16 k = 64 B), lane half lh: fp32 chunks c = 4 ks + 2 lh and c + 1  ->  (r*128 | (((2 lh) ^ swz) << 4)) ^ (ks << 6)
  // and that ^ 16;  W row n: hi chunk 2 ks + lh, lo chunk 4 + 2 ks + lh  ->  (n*128 | ((lh ^ swz) << 4)) ^ (ks << 5) and that ^ 64
  const unsigned int lds_base = (unsigned int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned int offA[2], offB;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + l31;
    offA[i] = row * 128 + ((((lh << 1)) ^ ((row >> 1) & 7)) << 4);
  }
  {
    const int row = wn * 32 + l31;
    offB = A_TILE_BYTES + row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
  }
  // per k-substep: A fragments of the two 32-row blocks (2 x 2 reads) and the W hi / lo fragments (2 reads)
  struct Frag {
    u32x4 a[2][2], bh, bl;
  };
  auto read_sub = [&](unsigned int sbase, int ks, Frag& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned int ad = (sbase + offA[i]) ^ (ks << 6);
      asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[i][0]) : "v"(ad) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(f.a[i][1]) : "v"(ad ^ 16u) : "memory");
    }
    const unsigned int adb = (sbase + offB) ^ (ks << 5);
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.bh) : "v"(adb) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.bl) : "v"(adb ^ 64u) : "memory");
  };
  auto issue_tile_reads = [&](int stage, Frag (&f)[KSW]) {
    const unsigned int sbase = lds_base + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < KSW; ++j) read_sub(sbase, khalf * KSW + j, f[j]);
  };
  // hi = upper 16 bits (truncation), lo = bf16_rne(a - hi): a = hi + lo up to 2^-17 |a|
  auto split8 = [&](const u32x4& v0, const u32x4& v1, bf16x8& hi, bf16x8& lo) {
    unsigned int w[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    union { u32x4 u; bf16x8 h; } H, L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned int x0 = w[2 * q], x1 = w[2 * q + 1];
      if (RELU) {
        x0 = (x0 & 0x80000000u) ? 0u : x0;
        x1 = (x1 & 0x80000000u) ? 0u : x1;
      }
      const unsigned int h0 = x0 & 0xffff0000u, h1 = x1 & 0xffff0000u;
      const float l0 = __uint_as_float(x0) - __uint_as_float(h0), l1 = __uint_as_float(x1) - __uint_as_float(h1);
      H.u[q] = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
      L.u[q] = pack_bf16x2(l0, l1);
    }
    hi = H.h;
    lo = L.h;
  };
#define SIU3R_X3_WAIT(CNT) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(f[j].a[0][0]), "+v"(f[j].a[0][1]), "+v"(f[j].a[1][0]), "+v"(f[j].a[1][1]), "+v"(f[j].bh), "+v"(f[j].bl)::"memory")
  auto tile_body = [&](int kt, int st, Frag (&f)[KSW], Frag (&nf)[KSW]) {
    const int kt_next = (kt + 2 < nkt) ? kt + 2 : -1;
    int stage_next = st + 2;
    if (stage_next >= STAGES) stage_next -= STAGES;
    int stage_read = st + 1;
    if (stage_read >= STAGES) stage_read -= STAGES;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      if (KSW == 2 && j == 0) SIU3R_X3_WAIT(6);
      else SIU3R_X3_WAIT(0);
      if (KSW == 1 && kt_next >= 0) {
        // one k-substep per wave: the pieces of tile kt+2 go out BEFORE the counted wait below, so that "LP pieces may remain
        // in flight" means tile kt+2's and tile kt+1 has landed (their stage was last read before the previous barrier)
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) issue_a(kt_next, stage_next, i);
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) issue_w(kt_next, stage_next, i);
      }
      if (j == KSW - 1 && kt + 1 < nkt) {
        if (kt_next >= 0) {
          if (LP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_tile_reads(stage_read, nf);
      }
      __builtin_amdgcn_sched_barrier(0);
      union { u32x4 u; bf16x8 h; } bh, bl;
      bh.u = f[j].bh;
      bl.u = f[j].bl;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 ah, al;
        split8(f[j].a[i][0], f[j].a[i][1], ah, al);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh.h, acc[i][0], 0, 0, 0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl.h, acc[i][0], 0, 0, 0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh.h, acc[i][0], 0, 0, 0);
      }
      if (KSW == 2 && kt_next >= 0 && j == 0) {
#pragma unroll
        for (int i = 0; i < A_DMA; ++i) issue_a(kt_next, stage_next, i);
#pragma unroll
        for (int i = 0; i < W_DMA; ++i) issue_w(kt_next, stage_next, i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 && kt_next >= 0) cursor_advance();
    if (MODE == 2 && kt_next >= 0) state_advance();
  };

  for (int q = 0; q < kbase; ++q) {
    if (MODE == 1) cursor_advance();
    if (MODE == 2) state_advance();
  }
  issue(0, 0);
  if (nkt > 1) issue(1, 1);
  if (nkt > 1) {
    if (LP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  Frag f0[KSW], f1[KSW];
  issue_tile_reads(0, f0);
  int st = 0;
  for (int kt = 0; kt < nkt; kt += 2) {
    tile_body(kt, st, f0, f1);
    st = (st + 1 == STAGES) ? 0 : st + 1;
    if (kt + 1 < nkt) {
      tile_body(kt + 1, st, f1, f0);
      st = (st + 1 == STAGES) ? 0 : st + 1;
    }
  }
#undef SIU3R_X3_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  siu3r_epi::run<1, KSPL, LNF>(p, acc, smem, tile_m, tile_n, z, t);
#endif
}

}  // namespace siu3r_gemm_dma

// A-mode the bf16 LDS-DMA kernels run this problem in, or -1 (outside their range: the register-staged kernel of gemm.hip takes it)
int siu3r_gemm_dma_mode(const siu3r_gemm_params& p) {
  const int64_t lim = 0xfffff000ll;
  if (p.w_lo || p.a_dtype != SIU3R_BF16 || p.a_mode == 2) return -1;
  if ((int64_t)p.n * p.kpad * 2 >= lim) return -1;
  if (p.a_mode == 0) return (((int64_t)(p.m - 1) * p.lda + p.k) * 2 >= lim || p.relu_in) ? -1 : 0;
  const int64_t img = (int64_t)(p.m / (p.oh * p.ow)) * p.ih * p.iw * p.cin * 2 + (int64_t)(p.pad * p.iw + p.pad) * p.cin * 2;
  if (img >= lim) return -1;
  const int mode = (p.cin % 64 == 0 && p.kh * p.kw <= 32 && p.kpad == p.k) ? 1 : 2;
  return (mode == 2 && p.relu_in) ? -1 : mode;
}
// ... and of the bf16x3 LDS-DMA kernel (fp32 activations, p.w_x3)
int siu3r_gemm_dma_x3_mode(const siu3r_gemm_params& p) {
  const int64_t lim = 0xfffff000ll;
  if (!p.w_x3 || p.a_dtype != SIU3R_F32 || (int64_t)p.n * p.kpad * 4 >= lim) return -1;
  if (p.a_mode == 0) return (((int64_t)(p.m - 1) * p.lda + p.k) * 4 >= lim || p.relu_in) ? -1 : 0;
  if (p.a_mode != 1) return -1;
  const int64_t img = (int64_t)(p.m / (p.oh * p.ow)) * p.ih * p.iw * p.cin * 4 + (int64_t)(p.pad * p.iw + p.pad) * p.cin * 4;
  if (img >= lim || p.cin % 4 != 0) return -1;
  if (p.cin % 32 == 0 && p.kh * p.kw <= 31) return 1;
  return p.relu_in ? -1 : 2;
}

// Called by siu3r_gemm() (gemm.hip) for bf16 activations without the bf16x3 split, dense or conv gather.
// Returns 1 when the problem is outside what the buffer-addressed kernels cover (the caller then uses the
// register-staged kernel): operands of 4 GiB or more, or a fused input ReLU outside MODE 1.
int siu3r_gemm_dma_launch(const siu3r_gemm_params& p, int ni, void* stream) {
  using namespace siu3r_gemm_dma;
  const int64_t lim = 0xfffff000ll;
  int mode;
  if ((int64_t)p.n * p.kpad * 2 >= lim) return 1;
  if (p.a_mode == 0) {
    mode = 0;
    if (((int64_t)(p.m - 1) * p.lda + p.k) * 2 >= lim || p.relu_in) return 1;
  } else {
    const int64_t img = (int64_t)(p.m / (p.oh * p.ow)) * p.ih * p.iw * p.cin * 2 + (int64_t)(p.pad * p.iw + p.pad) * p.cin * 2;
    if (img >= lim) return 1;
    mode = (p.cin % 64 == 0 && p.kh * p.kw <= 32 && p.kpad == p.k) ? 1 : 2;
    if (mode == 2 && p.relu_in) return 1;
  }
  const int tiles = 8 * p.map_rm * p.map_rn;
  dim3 grid(tiles, p.splitk > 1 ? p.splitk : 1, p.batch > 0 ? p.batch : 1);
  hipStream_t s = (hipStream_t)stream;
  static const bool no_ksplit = getenv("SIU3R_GEMM_NO_KSPLIT") != nullptr;  // A/B switch: 4-wave workgroups for the 128x64 tile
  const int kspl = no_ksplit ? 1 : 2;
  dim3 block(256 * kspl);
#define SIU3R_DMA_LAUNCH(NI_, MODE_, RELU_, KS_) hipLaunchKernelGGL((gemm_dma_kernel<NI_, MODE_, RELU_, KS_>), grid, block, 0, s, p)
#define SIU3R_DMA_MODES(NI_, KS_)                                      \
  do {                                                                 \
    if (mode == 0 && p.ln_stats) hipLaunchKernelGGL((gemm_dma_kernel<NI_, 0, false, KS_, true>), grid, block, 0, s, p); \
    else if (mode == 0) SIU3R_DMA_LAUNCH(NI_, 0, false, KS_);          \
    else if (mode == 1 && p.relu_in) SIU3R_DMA_LAUNCH(NI_, 1, true, KS_); \
    else if (mode == 1) SIU3R_DMA_LAUNCH(NI_, 1, false, KS_);          \
    else SIU3R_DMA_LAUNCH(NI_, 2, false, KS_);                         \
  } while (0)
  if (ni == 2 && kspl == 2) SIU3R_DMA_MODES(2, 2);
  else if (ni == 2) SIU3R_DMA_MODES(2, 1);
  else if (kspl == 2) SIU3R_DMA_MODES(1, 2);
  else SIU3R_DMA_MODES(1, 1);
#undef SIU3R_DMA_MODES
#undef SIU3R_DMA_LAUNCH
  SIU3R_LAUNCH_CHECK("siu3r_gemm(dma)");
  return 0;
}

// bf16x3 with fp32 activations on the LDS-DMA path (p.w_x3 = interleaved hi/lo planes).  Returns 1 when the problem is outside
// what the kernel covers (the caller falls back to the register-staged bf16x3 kernel of gemm.hip).
int siu3r_gemm_dma_x3_launch(const siu3r_gemm_params& p, void* stream) {
  using namespace siu3r_gemm_dma;
  const int64_t lim = 0xfffff000ll;
  if (!p.w_x3 || p.a_dtype != SIU3R_F32 || (int64_t)p.n * p.kpad * 4 >= lim) return 1;
  int mode;
  if (p.a_mode == 0) {
    mode = 0;
    if (((int64_t)(p.m - 1) * p.lda + p.k) * 4 >= lim || p.relu_in) return 1;
  } else if (p.a_mode == 1) {
    const int64_t img = (int64_t)(p.m / (p.oh * p.ow)) * p.ih * p.iw * p.cin * 4 + (int64_t)(p.pad * p.iw + p.pad) * p.cin * 4;
    if (img >= lim || p.cin % 4 != 0) return 1;
    if (p.cin % 32 == 0 && p.kh * p.kw <= 31) mode = 1;  // (kpad > k: the padded tiles sit on tap kh*kw, masked for every row)
    else if (!p.relu_in) mode = 2;
    else return 1;
  } else {
    return 1;
  }
  const int tiles = 8 * p.map_rm * p.map_rn;
  dim3 grid(tiles, p.splitk > 1 ? p.splitk : 1, p.batch > 0 ? p.batch : 1);
  hipStream_t s = (hipStream_t)stream;
  static const bool no_ksplit = getenv("SIU3R_GEMM_NO_KSPLIT") != nullptr;
  if (mode == 0 && p.ln_stats) {
    if (no_ksplit) hipLaunchKernelGGL((gemm_dma_x3_kernel<0, false, 1, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_dma_x3_kernel<0, false, 2, true>), grid, dim3(512), 0, s, p);
  } else if (no_ksplit) {
    if (mode == 0) hipLaunchKernelGGL((gemm_dma_x3_kernel<0, false, 1>), grid, dim3(256), 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL((gemm_dma_x3_kernel<2, false, 1>), grid, dim3(256), 0, s, p);
    else if (p.relu_in) hipLaunchKernelGGL((gemm_dma_x3_kernel<1, true, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_dma_x3_kernel<1, false, 1>), grid, dim3(256), 0, s, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((gemm_dma_x3_kernel<0, false, 2>), grid, dim3(512), 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL((gemm_dma_x3_kernel<2, false, 2>), grid, dim3(512), 0, s, p);
    else if (p.relu_in) hipLaunchKernelGGL((gemm_dma_x3_kernel<1, true, 2>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((gemm_dma_x3_kernel<1, false, 2>), grid, dim3(512), 0, s, p);
  }
  SIU3R_LAUNCH_CHECK("siu3r_gemm(dma x3)");
  return 0;
}
