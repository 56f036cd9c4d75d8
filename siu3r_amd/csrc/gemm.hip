// MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// C[M,N] = epilogue( A[M,K] x W[N,K]^T ), bf16 MFMA operands, fp32 accumulate.
// Tile 128 x (64*NI) x 64, 256 threads = 4 waves (2x2), each wave 64 x 32*NI = 2 x NI
// v_mfma_f32_32x32x16_bf16 tiles.  Operands are register-staged (global -> VGPR -> LDS) with a prefetch
// distance of TWO K-tiles (two register sets, loads for tile k+2 are in flight while tile k is multiplied and
// tile k+1 is written to LDS) and a double-buffered, XOR-swizzled LDS image (one barrier per K-step).
// At batch 1 the SIU3R GEMMs have 50-550 tiles for 256 CUs, i.e. <= 1-2 workgroups per CU: the K-loop is
// latency-bound, which is what the deep prefetch and the narrow NI=1 tile (2x the workgroups) address.
// bf16x3 mode (SPLIT): A is fp32, split on the fly into hi+lo bf16; W carries a pre-split lo plane;
// three MFMAs per product recover ~fp32 accuracy.
//
// A addressing modes: dense rows, NHWC implicit conv gather (any KHxKW/stride/pad, Cin % 8 == 0),
// NCHW fp32 16x16 patchify (coalesced patch-embed im2col, reference croco/patch_embed.py:19-29).
#include <stdlib.h>

#include "common.h"
#include "gemm_epilogue.h"

int siu3r_gemm_dma_launch(const siu3r_gemm_params& p, int ni, void* stream);  // gemm_dma.hip (bf16 LDS-DMA fast path)
int siu3r_gemm_dma_x3_launch(const siu3r_gemm_params& p, void* stream);       // gemm_dma.hip (bf16x3 LDS-DMA path, fp32 A)
int siu3r_gemm_dma_mode(const siu3r_gemm_params& p);                          // -1: outside the bf16 LDS-DMA kernels' range
int siu3r_gemm_dma_x3_mode(const siu3r_gemm_params& p);
int siu3r_gemm_pp_mode(const siu3r_gemm_params& p);                           // gemm_pp.hip (8-wave ping-pong kernels); -1: outside their range
void siu3r_gemm_pp_name(const siu3r_gemm_params& p, int cfg, char* buf, int n);
int siu3r_gemm_pp_launch(const siu3r_gemm_params& p, int cfg, void* stream);
int siu3r_gemm_skinny_launch(const siu3r_gemm_params& p, void* stream);
bool siu3r_gemm_pp_folds_skinny(const siu3r_gemm_params& p);
bool siu3r_gemm_pp_a_x3_ok(const siu3r_gemm_params& p);          // pre-split A planes readable / output planes writable by a ping-pong launch
bool siu3r_gemm_pp_c_x3_ok(const siu3r_gemm_params& p, int cfg);
static const bool g_disable_dma = getenv("SIU3R_GEMM_NO_DMA") != nullptr;     // debugging / A-B switch
// 128x64 tiles (two workgroups per CU) beat 128x128 (one per CU: 96 KiB ring) at every size measured on gfx950 --
// finer wave quantisation and a second workgroup to overlap prologue/epilogue; 128x128 stays reachable for A/B runs
static const int g_narrow_max = getenv("SIU3R_GEMM_NARROW_MAX") ? atoi(getenv("SIU3R_GEMM_NARROW_MAX")) : 0x7fffffff;

namespace {

constexpr int BM = 128, BK = 64;
constexpr int A_TILE_BYTES = BM * BK * 2;  // 16 KiB

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // row stride 128 B, 16-B chunks XOR-swizzled so that ds_read_b128 lane groups hit 16 distinct slots
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int A_F32, int SPLIT, int NI>
struct Regs {
  float4 ra0[A_F32 ? 4 : 1], ra1[A_F32 ? 4 : 1];  // fp32 A: 8 floats per row-chunk
  uint4 rab[A_F32 ? 1 : 4];                       // bf16 A
  uint4 rwh[2 * NI], rwl[SPLIT ? 2 * NI : 1];
};

template <int A_F32, int SPLIT, int NI, bool LNF = false>
__global__ __launch_bounds__(256) void gemm_kernel(const siu3r_gemm_params p) {
  constexpr int BN = 64 * NI;
  constexpr int B_TILE_BYTES = BN * BK * 2;
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int STAGE_BYTES = PLANES * (A_TILE_BYTES + B_TILE_BYTES);
  constexpr int WROWS = 2 * NI;  // weight rows per thread per K-tile
  // stage s: [A_hi | B_hi | (A_lo | B_lo)]
  constexpr int SMEM_BYTES = 2 * STAGE_BYTES > siu3r_epi::staging_bytes<NI>() ? 2 * STAGE_BYTES : siu3r_epi::staging_bytes<NI>();
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware block -> tile map.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only):
  // the tile grid is cut into gy x gx = 8 rectangular regions, one per XCD, and an XCD's workgroups walk its
  // region row-major, so that the ~32 workgroups co-resident on an XCD form a compact patch that shares A-row
  // and W-column panels in that XCD's private 4 MiB L2 instead of re-fetching them over the fabric.
  const int tiles_m = (p.m + BM - 1) / BM, tiles_n = (p.n + BN - 1) / BN;
  const int xcd = (blockIdx.x + blockIdx.z) & 7, li = blockIdx.x >> 3;  // (regions rotate with the batch item: few tiles per item x many items still load all XCDs)
  const int ry = xcd / p.map_gx, rx = xcd - ry * p.map_gx;
  const int lm = li / p.map_rn, ln = li - lm * p.map_rn;
  const int tile_m = ry * p.map_rm + lm, tile_n = rx * p.map_rn + ln;
  if (lm >= p.map_rm || tile_m >= tiles_m || tile_n >= tiles_n) return;
  const int z = blockIdx.z;

  const int esz_a = A_F32 ? 4 : 2;
  const siu3r_zoff zof = siu3r_batch_offsets(p, z);
  const unsigned char* Ab = (const unsigned char*)p.a + zof.a * esz_a;
  const u16* Wh = (const u16*)p.w_hi + zof.w;
  const u16* Wl = SPLIT ? (const u16*)p.w_lo + zof.w : nullptr;

  // ---- per-thread load geometry: chunk (8 k-elements) x 4 A rows, x WROWS weight rows
  const int chunk = t & 7, row0 = t >> 3;
  int64_t a_base[4];
  int a_iy0[4], a_ix0[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = tile_m * BM + row0 + 32 * i;
    a_ok[i] = true;  // rows beyond M are clamped to M-1: their products are never stored
    if (m > p.m - 1) m = p.m - 1;
    a_iy0[i] = a_ix0[i] = 0;
    a_base[i] = 0;
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
  int64_t w_base[WROWS];
  bool w_ok[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    int n = tile_n * BN + row0 + 32 * i;
    w_ok[i] = true;  // columns beyond N are clamped: never stored
    if (n > p.n - 1) n = p.n - 1;
    w_base[i] = (int64_t)n * p.kpad + chunk * 8;
  }

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  using R = Regs<A_F32, SPLIT, NI>;

  auto load_tile = [&](int kt, R& rg) {
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
      // branch-free: out-of-range taps / K tail read a clamped (valid) address and are zeroed by a select,
      // so the loads stay in one basic block and the compiler can keep them in flight with counted vmcnt
      bool ok = k_ok;
      int64_t off;
      if (p.a_mode == 0) {
        off = a_base[i] + (k_ok ? k0 : 0);
      } else if (p.a_mode == 1) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        ok = ok && iy >= 0 && iy < p.ih && ix >= 0 && ix < p.iw;
        iy = min(max(iy, 0), p.ih - 1);
        ix = min(max(ix, 0), p.iw - 1);
        off = a_base[i] + ((int64_t)iy * p.iw + ix) * p.cin + (k_ok ? c0 : 0);
      } else {
        off = a_base[i] + (k_ok ? koff : 0);
      }
      if constexpr (A_F32) {
        const float4* q = (const float4*)(Ab + off * 4);
        const float4 v0 = q[0], v1 = q[1];
        const float4 zz = make_float4(0, 0, 0, 0);
        rg.ra0[i] = ok ? v0 : zz;
        rg.ra1[i] = ok ? v1 : zz;
      } else {
        const uint4 v = *(const uint4*)(Ab + off * 2);
        rg.rab[i] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      rg.rwh[i] = *(const uint4*)(Wh + w_base[i] + (int64_t)kt * BK);
      if constexpr (SPLIT) rg.rwl[i] = *(const uint4*)(Wl + w_base[i] + (int64_t)kt * BK);
    }
  };

  auto store_tile = [&](int stage, const R& rg) {
    unsigned char* sA = smem + stage * STAGE_BYTES;
    unsigned char* sB = sA + A_TILE_BYTES;
    unsigned char* sAl = sB + B_TILE_BYTES;
    unsigned char* sBl = sAl + A_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = lds_off(row0 + 32 * i, chunk);
      if constexpr (A_F32) {
        float f[8] = {rg.ra0[i].x, rg.ra0[i].y, rg.ra0[i].z, rg.ra0[i].w, rg.ra1[i].x, rg.ra1[i].y, rg.ra1[i].z, rg.ra1[i].w};
        if (p.relu_in) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        if constexpr (SPLIT) {
          uint4 hi, lo;
          split_bf16x8(f, hi, lo);
          *(uint4*)(sA + o) = hi;
          *(uint4*)(sAl + o) = lo;
        } else {
          *(uint4*)(sA + o) = pack_bf16x8(f);
        }
      } else {
        uint4 v = rg.rab[i];
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
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int o = lds_off(row0 + 32 * i, chunk);
      *(uint4*)(sB + o) = rg.rwh[i];
      if constexpr (SPLIT) *(uint4*)(sBl + o) = rg.rwl[i];
    }
  };

  auto compute = [&](int stage) {
    const unsigned char* sA = smem + stage * STAGE_BYTES;
    const unsigned char* sB = sA + A_TILE_BYTES;
    const unsigned char* sAl = sB + B_TILE_BYTES;
    const unsigned char* sBl = sAl + A_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks * 2 + lh;
      bf16x8 fa[2], fb[NI], fal[2], fbl[NI];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31;
        fa[i] = as_bf16x8(*(const uint4*)(sA + lds_off(ra, c)));
        if constexpr (SPLIT) fal[i] = as_bf16x8(*(const uint4*)(sAl + lds_off(ra, c)));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int rb = wn * (32 * NI) + j * 32 + l31;
        fb[j] = as_bf16x8(*(const uint4*)(sB + lds_off(rb, c)));
        if constexpr (SPLIT) fbl[j] = as_bf16x8(*(const uint4*)(sBl + lds_off(rb, c)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (SPLIT) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fb[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fbl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // ---- main loop: prefetch distance 2 (register sets r0/r1 alternate), LDS double buffer
  const int nkt = p.kpad / BK;
  // raw barriers: __syncthreads() would also drain the global loads of the tiles in flight (hipcc puts s_waitcnt vmcnt(0) in
  // front of it), i.e. undo the prefetch; LDS visibility only needs lgkmcnt(0)
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  R r0, r1;
  load_tile(0, r0);
  if (nkt > 1) load_tile(1, r1);
  store_tile(0, r0);
  lds_barrier();
  for (int kt = 0; kt < nkt; kt += 2) {
    // even step: tile kt is in LDS stage 0, tile kt+1 in r1; fetch tile kt+2 into r0
    if (kt + 2 < nkt) load_tile(kt + 2, r0);
    compute(0);
    if (kt + 1 < nkt) store_tile(1, r1);
    lds_barrier();
    if (kt + 1 >= nkt) break;
    // odd step: tile kt+1 is in LDS stage 1, tile kt+2 in r0; fetch tile kt+3 into r1
    if (kt + 3 < nkt) load_tile(kt + 3, r1);
    compute(1);
    if (kt + 2 < nkt) store_tile(0, r0);
    lds_barrier();
  }

  // ---- epilogue: LDS-staged, row-wise vectorised (gemm_epilogue.h)
  siu3r_epi::run<NI, 1, LNF>(p, acc, smem, tile_m, tile_n, z, t);
}

template <int NI>
int launch(const siu3r_gemm_params& pin, hipStream_t s) {
  constexpr int BN = 64 * NI;
  siu3r_gemm_params p = pin;
  const int tm = (p.m + BM - 1) / BM, tn = (p.n + BN - 1) / BN;
  // choose the 8-region factorisation (gy x gx) with the smallest per-XCD operand footprint
  // 8 regions (gy x gx), one per XCD: first the factorisation whose largest region holds the fewest tiles (an XCD with more workgroups
  // than CUs runs a second round while the others idle: 5 row tiles cut 2 x 4 put 18 tiles per batch item on four XCDs and 12 on the
  // rest), then the smallest per-XCD operand footprint
  int best = -1;
  long best_cost = 0;
  for (int gx = 1; gx <= 8; gx *= 2) {
    const int gy = 8 / gx;
    const int rm = (tm + gy - 1) / gy, rn = (tn + gx - 1) / gx;
    const long cost = (long)rm * rn * 1000000 + (long)rm * BM + (long)rn * BN;
    if (best < 0 || cost < best_cost) {
      best = gx;
      best_cost = cost;
    }
  }
  p.map_gx = best;
  p.map_rm = (tm + (8 / best) - 1) / (8 / best);
  p.map_rn = (tn + best - 1) / best;
  const int tiles = 8 * p.map_rm * p.map_rn;
  dim3 grid(tiles, 1, p.batch > 0 ? p.batch : 1), block(256);
  if (p.splitk <= 1 || NI != 1) p.splitk = 0;  // (slabs are sized for 128 x 64 tiles)
  if (!p.w_lo && p.a_dtype == SIU3R_BF16 && p.a_mode != 2 && !g_disable_dma) {
    const int rc = siu3r_gemm_dma_launch(p, NI, s);
    if (rc <= 0) return rc;  // 1: outside the buffer-addressed kernels' range -> register-staged kernel below
  }
  if (p.w_x3 && NI == 1 && !g_disable_dma) {
    const int rc = siu3r_gemm_dma_x3_launch(p, s);
    if (rc <= 0) return rc;  // 1: outside the kernel's range -> register-staged bf16x3 below
  }
  p.splitk = 0;  // the register-staged kernels multiply the whole K
  if (p.ln_stats) {  // (rare: a folded LayerNorm outside the LDS-DMA kernels' range)
    if (p.w_lo) hipLaunchKernelGGL((gemm_kernel<1, 1, NI, true>), grid, block, 0, s, p);
    else if (p.a_dtype == SIU3R_F32) hipLaunchKernelGGL((gemm_kernel<1, 0, NI, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<0, 0, NI, true>), grid, block, 0, s, p);
  } else if (p.w_lo) {
    hipLaunchKernelGGL((gemm_kernel<1, 1, NI>), grid, block, 0, s, p);
  } else if (p.a_dtype == SIU3R_F32) {
    hipLaunchKernelGGL((gemm_kernel<1, 0, NI>), grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_kernel<0, 0, NI>), grid, block, 0, s, p);
  }
  SIU3R_LAUNCH_CHECK("siu3r_gemm");
  return 0;
}

}  // namespace


// ---- what to launch.  One decision function for siu3r_gemm and siu3r_gemm_plan: kernel family, tile, split-K slices, skinny rows.
static int g_tile_default = getenv("SIU3R_GEMM_PP") ? atoi(getenv("SIU3R_GEMM_PP")) : 0;
static int g_no_skinny = getenv("SIU3R_GEMM_NO_SKINNY") ? 1 : 0;
static int g_no_splitk = getenv("SIU3R_NO_SPLITK") ? atoi(getenv("SIU3R_NO_SPLITK")) : 0;
static int g_no_tuned = getenv("SIU3R_GEMM_NO_TUNED") ? 1 : 0;
static int g_force_skinny = getenv("SIU3R_GEMM_FORCE_SKINNY") ? 1 : 0;  // tests: remainder rows go to the skinny launch whenever it is applicable
static float g_bf16_pp_hurdle = getenv("SIU3R_GEMM_BF16_PP_HURDLE") ? (float)atof(getenv("SIU3R_GEMM_BF16_PP_HURDLE")) : 1.3f;
extern "C" int siu3r_gemm_tune(int key, int value) {
  if (key == 0) g_tile_default = value;
  else if (key == 1) g_no_skinny = value;
  else if (key == 2) g_no_splitk = value;
  else if (key == 3) g_no_tuned = value;
  else if (key == 4) g_force_skinny = value;
  else return 1;
  return 0;
}

#include "gemm_tuned.h"

namespace {
// measured (tile, split-K) choice of this exact problem, nullptr if it is not in the table
const siu3r_tuned_entry* tuned_entry(const siu3r_gemm_params& p, bool x3) {
  const int Z = p.batch > 0 ? p.batch : 1, ln = p.ln_stats ? 1 : 0;
  for (const siu3r_tuned_entry* e = kTuned; e->m; ++e)
    if (e->m == p.m && e->n == p.n && e->k == p.k && e->batch == Z && e->a_mode == p.a_mode && e->out_mode == p.out_mode && e->kh == p.kh &&
        e->stride == p.stride && e->x3 == (x3 ? 1 : 0) && e->ln == ln)
      return e;
  return nullptr;
}

// Cost model (microseconds), fitted to graph-timed replays of the 495 GEMM launches of a real 2 x 512^2 step under every tile
// configuration (tools/gemm_replay.py; mean error ~15 %): a launch runs ceil(workgroups / slots) rounds; a round costs t0 (prologue
// latency plus the epilogue, which the one-workgroup-per-CU ping-pong kernels expose in full every round while the two co-resident
// workgroups of the 128 x 64 kernels hide each other's) plus its K steps (a step = 64 bytes of every operand row: 16 fp32 or 32 bf16);
// split-K adds a fixed ~2.5 us and its slab traffic (~5 TB/s: written and read once).  Slots: 256 for the 8-wave ping-pong kernels
// (one workgroup per CU), 512 for the 128 x 64 kernels.  Separate constants for the bf16 kernels (32 bf16 per step).
struct Cand { int cfg, bm, bn, slots; float t0_x3, t_step_x3, t0_bf, t_step_bf; };
// (ping-pong rows: one-round K sweeps of tools/mb_one.py after the fast row pass of gemm_epilogue_pp.h -- t0 was 27.5 / 16.8 / 7.7 us in
// bf16x3 and 36.9 / 12.1 / 7.8 in bf16 with the general pass)
const Cand kCands[] = {{SIU3R_TILE_PP_256x256, 256, 256, 256, 16.3f, 1.18f, 12.1f, 0.80f},
                       {SIU3R_TILE_PP_256x128, 256, 128, 256, 11.7f, 0.65f, 8.5f, 0.47f},
                       {SIU3R_TILE_PP_128x128, 128, 128, 256, 7.8f, 0.40f, 6.5f, 0.30f},
                       {SIU3R_TILE_128x64, 128, 64, 512, 6.2f, 0.46f, 6.8f, 0.31f}};

void plan(const siu3r_gemm_params& p, siu3r_gemm_plan_t& pl) {
  memset(&pl, 0, sizeof(pl));
  const bool x3 = p.w_x3 != nullptr && p.a_dtype == SIU3R_F32;
  const int pp_mode = (g_disable_dma || p.a_mode == 2) ? -1 : siu3r_gemm_pp_mode(p);
  const int dma_mode = g_disable_dma ? -1 : (x3 ? siu3r_gemm_dma_x3_mode(p) : siu3r_gemm_dma_mode(p));
  int force = p.tile_cfg ? p.tile_cfg : g_tile_default;
  const bool may_split = !g_no_splitk && p.splitk != 1 && p.sk_ws && p.sk_cnt;
  int want_s = p.splitk;  // 0 auto, 1 never, > 1 exactly
  if (force == 0 && !g_no_tuned) {
    if (const siu3r_tuned_entry* e = tuned_entry(p, x3)) {
      force = e->tile_cfg;
      if (e->splitk > 0 && p.splitk == 0) want_s = (e->splitk > 1 && !may_split) ? 1 : e->splitk;
    }
  }
  const int64_t Z = p.batch > 0 ? p.batch : 1;
  const int ksteps = p.kpad / (x3 ? 16 : 32);  // steps of 64 operand-row bytes
  float best_t = 0.f;
  int best = -1, best_s = 1, best_skinny = 0;
  for (int ci = 0; ci < 4; ++ci) {
    const Cand& c = kCands[ci];
    const bool is_pp = c.cfg > 0;
    if (is_pp && pp_mode < 0) continue;
    if (force != 0 && force != c.cfg && !(force > 0 && !is_pp && pp_mode < 0)) continue;  // a forced ping-pong tile falls back to 128 x 64 outside its range
    // the last <= 32 rows of a dense problem may go to the skinny kernel instead of opening a row of tiles of their own: worth it when
    // that row of tiles would cost another round of workgroups (both options are priced)
    const int rem = p.m % c.bm;
    const bool skinny_ok = is_pp && !g_no_skinny && p.a_mode == 0 && rem != 0 && rem <= 32 && p.n >= 64;
    for (int sk = (skinny_ok && g_force_skinny) ? 1 : 0; sk <= (skinny_ok ? 1 : 0); ++sk) {
      const int skinny = sk ? rem : 0;
      const int mrows = p.m - skinny;
      const int64_t tiles = (int64_t)((mrows + c.bm - 1) / c.bm) * ((p.n + c.bn - 1) / c.bn) * Z;
      const bool can_split = may_split && (is_pp || (dma_mode >= 0 && c.bn == 64));
      const bool fixed_s = want_s >= 1 && (want_s == 1 || can_split);
      const int smax = fixed_s ? want_s : (can_split ? 8 : 1);
      for (int S = (fixed_s ? want_s : 1); S <= smax; S *= 2) {
        if (S > 1) {
          if (ksteps / S < (is_pp ? 8 : 16) && !fixed_s) break;  // every slice keeps a few steps (and whole phases)
          if (tiles * S * c.bm * c.bn > p.sk_ws_floats || tiles > p.sk_cnt_n) break;
          if (tiles >= 2 * c.slots && !fixed_s) break;
        }
        const int64_t wgs = tiles * S;
        const int64_t rounds = (wgs + c.slots - 1) / c.slots;
        float t = rounds * ((x3 ? c.t0_x3 : c.t0_bf) + (float)((ksteps + S - 1) / S) * (x3 ? c.t_step_x3 : c.t_step_bf));
        if (mrows == 0) t = 0.f;
        // skinny launch: a kernel boundary plus its K loop (its fragment-shaped loads are address-bound: ~9 us per 1024 k in bf16x3)
        // (<= 4 rows and 2048 <= kpad <= 4096: the matrix-vector kernel of gemm_pp.hip, 26 us at K = 4096 in bf16x3)
        if (skinny && mrows > 0 && p.kpad <= 2048 && !getenv("SIU3R_GEMM_NO_FOLD")) t += 1.0f + (x3 ? 5.0f : 3.0f) * (float)p.kpad / 1024.f;  // carried by the tile launch (extra workgroups at its tail)
        else if (skinny) t += (skinny <= 4 && p.kpad >= 2048 && p.kpad <= 4096) ? 5.0f + (x3 ? 5.0f : 3.0f) * (float)p.kpad / 1024.f : 3.5f + (x3 ? 9.3f : 5.0f) * (float)p.kpad / 1024.f;
        if (S > 1) t += 2.5f + (float)(2.0 * S * tiles * c.bm * c.bn * 4.0 / 5.0e6);
        // bf16: the 128 x 64 LDS-DMA kernel is already within a few per cent of the best tile on almost every shape of the network and
        // shares a CU with the other chains' kernels; end to end the ping-pong tiles LOSE 7-11 % there (same-box A/B of bench.py at
        // B = 1 and 8), so they must promise a clear gain
        if (!x3 && is_pp && force == 0) t *= g_bf16_pp_hurdle;
        if (best < 0 || t < best_t) {
          best = ci;
          best_t = t;
          best_s = S;
          best_skinny = skinny;
        }
      }
    }
  }
  if (best < 0) best = 3;
  const Cand& c = kCands[best];
  pl.tile_cfg = c.cfg;
  pl.bm = c.bm;
  pl.bn = c.bn;
  pl.splitk = best_s;
  pl.skinny_rows = best_skinny;
  if (c.cfg < 0) {
    // the 128 x 64 family: 128 x 128 tiles only for A/B runs (SIU3R_GEMM_NARROW_MAX); register-staged kernels never split
    const int64_t tiles128 = (int64_t)((p.m + BM - 1) / BM) * ((p.n + 127) / 128) * Z;
    const bool narrow = (p.n <= 64) || (tiles128 < g_narrow_max && p.n > 64) || p.w_x3 != nullptr;
    pl.bn = narrow ? 64 : 128;
    if (!narrow || dma_mode < 0) pl.splitk = 1;
  }
  if (pl.splitk > 1) {
    const int64_t tiles = (int64_t)((p.m - pl.skinny_rows + pl.bm - 1) / pl.bm) * ((p.n + pl.bn - 1) / pl.bn) * Z;
    pl.ws_floats = tiles * pl.splitk * pl.bm * pl.bn;
    pl.counters = (int32_t)tiles;
  }
  // the kernel's name as rocprofv3 prints it (bench.py's roofline leg keys its HIP-event averages by it)
  auto tf = [](bool b) { return b ? "true" : "false"; };
  if (c.cfg > 0) {
    siu3r_gemm_pp_name(p, c.cfg, pl.kernel, sizeof(pl.kernel));
  } else if (x3 && dma_mode >= 0 && pl.bn == 64) {
    snprintf(pl.kernel, sizeof(pl.kernel), "siu3r_gemm_dma::gemm_dma_x3_kernel<%d, %s, 2, %s>", dma_mode, tf(dma_mode == 1 && p.relu_in), tf(dma_mode == 0 && p.ln_stats));
  } else if (!x3 && !p.w_lo && dma_mode >= 0) {
    snprintf(pl.kernel, sizeof(pl.kernel), "siu3r_gemm_dma::gemm_dma_kernel<%d, %d, %s, 2, %s>", pl.bn / 64, dma_mode, tf(dma_mode == 1 && p.relu_in), tf(dma_mode == 0 && p.ln_stats));
  } else {
    snprintf(pl.kernel, sizeof(pl.kernel), "gemm_kernel<%d, %d, %d%s>", p.a_dtype == SIU3R_F32 ? 1 : 0, p.w_lo ? 1 : 0, pl.bn / 64, p.ln_stats ? ", true" : "");
  }
}
}  // namespace

// parameter validation shared by siu3r_gemm and siu3r_gemm_plan (the plan divides by the conv geometry: a malformed block must come
// back as an error string from both, not as SIGFPE from the query)
static int validate_gemm(const siu3r_gemm_params& p) {
  SIU3R_CHECK(p.a && p.w_hi && (p.c || p.c_x3), "siu3r_gemm: null operand pointer");
  SIU3R_CHECK(p.m > 0 && p.n > 0 && p.k > 0, "siu3r_gemm: empty problem (m=%d n=%d k=%d)", p.m, p.n, p.k);
  SIU3R_CHECK(p.kpad % BK == 0 && p.kpad >= p.k, "siu3r_gemm: kpad=%d must be a multiple of 64 and >= k=%d", p.kpad, p.k);
  SIU3R_CHECK(p.a_dtype == SIU3R_BF16 || p.a_dtype == SIU3R_F32, "siu3r_gemm: bad a_dtype %d", p.a_dtype);
  SIU3R_CHECK(p.c_dtype == SIU3R_BF16 || p.c_dtype == SIU3R_F32, "siu3r_gemm: bad c_dtype %d", p.c_dtype);
  SIU3R_CHECK(!(p.w_lo && p.a_dtype != SIU3R_F32), "siu3r_gemm: bf16x3 mode needs fp32 activations");
  SIU3R_CHECK(p.a_mode >= 0 && p.a_mode <= 2, "siu3r_gemm: bad a_mode %d", p.a_mode);
  if (p.a_mode != 0) SIU3R_CHECK(p.oh > 0 && p.ow > 0 && p.ih > 0 && p.iw > 0 && (p.a_mode == 2 || (p.cin > 0 && p.kh > 0 && p.kw > 0 && p.stride > 0)),
                                 "siu3r_gemm: conv geometry must be positive (ih=%d iw=%d cin=%d kh=%d kw=%d stride=%d oh=%d ow=%d)", p.ih, p.iw, p.cin, p.kh, p.kw, p.stride, p.oh, p.ow);
  if (p.a_mode == 0) {
    SIU3R_CHECK(p.k % 8 == 0 && p.lda % 8 == 0, "siu3r_gemm: dense A needs k %% 8 == 0 and lda %% 8 == 0 (k=%d lda=%ld)", p.k, (long)p.lda);
  } else if (p.a_mode == 1) {
    // (cin % 4: fp32 activations on the bf16x3 LDS-DMA gather only -- 16-byte chunks of 4 channels)
    SIU3R_CHECK(p.cin % 8 == 0 || (p.cin % 4 == 0 && p.w_x3 && p.a_dtype == SIU3R_F32 && !p.relu_in && !g_disable_dma),
                "siu3r_gemm: conv gather needs cin %% 8 == 0 (cin %% 4 == 0 for fp32 activations with w_x3) (cin=%d)", p.cin);
    SIU3R_CHECK(p.k == p.kh * p.kw * p.cin, "siu3r_gemm: conv k=%d != kh*kw*cin", p.k);
    SIU3R_CHECK(p.m % (p.oh * p.ow) == 0, "siu3r_gemm: conv m=%d not a multiple of oh*ow", p.m);
  } else {
    SIU3R_CHECK(p.a_dtype == SIU3R_F32 && p.k == 768 && p.ih == p.oh * 16 && p.iw == p.ow * 16,
                "siu3r_gemm: patchify mode needs fp32 NCHW input, k=768, H,W multiples of 16");
  }
  if (p.out_mode == 1) SIU3R_CHECK(p.n == p.up * p.up * p.cout && p.m % (p.ih * p.iw) == 0 && p.cout % 8 == 0, "siu3r_gemm: bad conv-transpose geometry (cout %% 8 == 0 required)");
  if (p.rope_cos) SIU3R_CHECK(p.rope_sin && p.rope_pos && p.rope_ncols % 64 == 0 && p.rope_ncols <= p.n && p.act == 0 && p.out_mode == 0,
                              "siu3r_gemm: bad RoPE epilogue arguments");
  if (p.up_src) SIU3R_CHECK(p.a_mode == 1 && p.out_mode == 0 && p.oh % 2 == 0 && p.ow % 2 == 0, "siu3r_gemm: up_src needs conv mode with even output size");
  if (p.ln_stats)
    SIU3R_CHECK(p.ln_c1 && p.ln_c2 && !p.bias && p.a_mode == 0 && p.ln_tiles == (p.k + 63) / 64 && p.ln_tiles <= 16 && !p.relu_in,
                "siu3r_gemm: folded LayerNorm needs c1/c2, no bias, dense A and ln_tiles == ceil(k/64) <= 16 (k=%d, ln_tiles=%d)", p.k, p.ln_tiles);
  if (p.stats_out) SIU3R_CHECK(p.out_mode == 0 && !p.up_src, "siu3r_gemm: stats_out needs a plain row-major output");
  if (p.c_aux) SIU3R_CHECK(p.out_mode == 0, "siu3r_gemm: c_aux needs a plain row-major output");
  if (p.splitk > 1)
    SIU3R_CHECK(p.sk_ws && p.sk_cnt && p.splitk <= 64 && p.splitk <= p.kpad / 64, "siu3r_gemm: split-K needs a workspace, zeroed counters and splitk <= kpad / 64 (splitk=%d)", p.splitk);
  SIU3R_CHECK(p.tile_cfg >= -1 && p.tile_cfg <= 3, "siu3r_gemm: bad tile_cfg %d", p.tile_cfg);
  if (p.bmod > 0) SIU3R_CHECK(p.batch > 0 && p.batch % p.bmod == 0, "siu3r_gemm: batch %d is not a multiple of bmod %d", p.batch, p.bmod);
  return 0;
}

extern "C" int siu3r_gemm_plan(const siu3r_gemm_params* pp, siu3r_gemm_plan_t* out) {
  SIU3R_CHECK(pp && out, "siu3r_gemm_plan: null argument");
  if (int rc = validate_gemm(*pp)) return rc;
  plan(*pp, *out);
  out->a_x3_ok = out->tile_cfg > 0 && siu3r_gemm_pp_a_x3_ok(*pp);
  out->c_x3_ok = out->tile_cfg > 0 && siu3r_gemm_pp_c_x3_ok(*pp, out->tile_cfg);
  return 0;
}

extern "C" int siu3r_gemm(const siu3r_gemm_params* pp, void* stream) {
  const siu3r_gemm_params& p = *pp;
  if (int rc = validate_gemm(p)) return rc;
  siu3r_gemm_plan_t pl;
  plan(p, pl);
  hipStream_t s = (hipStream_t)stream;
  // pre-split planes exist on the ping-pong path only: anything else would read / leave garbage
  SIU3R_CHECK(!p.a_x3 || (pl.tile_cfg > 0 && siu3r_gemm_pp_a_x3_ok(p)), "siu3r_gemm: a_x3 (pre-split A planes) needs a ping-pong plan, bf16x3 weights, dense A, "
              "k == kpad and lda / batch strides %% 32 == 0 (this block runs %s): query siu3r_gemm_plan_t.a_x3_ok first", pl.kernel);
  SIU3R_CHECK(!p.c_x3 || (pl.tile_cfg > 0 && siu3r_gemm_pp_c_x3_ok(p, pl.tile_cfg)), "siu3r_gemm: c_x3 (output planes) needs a ping-pong plan and the plain fp32 "
              "row-major output form with n %% 64 == 0, ldc / batch strides %% 32 == 0 (this block runs %s): query siu3r_gemm_plan_t.c_x3_ok first", pl.kernel);
  SIU3R_CHECK(p.c || p.c_x3, "siu3r_gemm: null output");
  siu3r_gemm_params q = p;
  q.splitk = pl.splitk;
  if (pl.splitk > 1) SIU3R_CHECK(p.sk_ws && p.sk_cnt && p.sk_ws_floats >= pl.ws_floats && p.sk_cnt_n >= pl.counters,
                                 "siu3r_gemm: split-K workspace too small (%ld floats / %d counters needed)", (long)pl.ws_floats, pl.counters);
  // "splitk > 1 = exactly that many slices" (include/siu3r_hip.h): a kernel that cannot split is an error, not a silent single slice
  SIU3R_CHECK(!(p.splitk > 1 && pl.splitk != p.splitk), "siu3r_gemm: splitk=%d was requested but %s cannot split K for this problem (it would run %d slice(s)); "
              "request 0 / 1 or another tile_cfg", p.splitk, pl.kernel, pl.splitk);
  if (pl.tile_cfg > 0) {
    q.m_main = pl.skinny_rows > 0 ? p.m - pl.skinny_rows : 0;
    if (!(pl.skinny_rows > 0 && q.m_main == 0)) {
      const int rc = siu3r_gemm_pp_launch(q, pl.tile_cfg, stream);
      if (rc) return rc;
    }
    if (pl.skinny_rows > 0 && !(q.m_main > 0 && siu3r_gemm_pp_folds_skinny(q))) {
      q.splitk = 0;
      return siu3r_gemm_skinny_launch(q, stream);
    }
    return 0;
  }
  return pl.bn == 64 ? launch<1>(q, s) : launch<2>(q, s);
}
