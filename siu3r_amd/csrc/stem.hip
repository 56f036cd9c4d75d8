// The Gaussian heads' stem as ONE dedicated bf16x3 kernel: x = ReLU(conv7x7(image) + bias) + up_x2(path_1)
// (reference src/models/heads/dpt_gs_head.py:71-77 input_merger, :158-162 feat_up + add; the bilinear x2 upsample is
// Interpolate(scale_factor=2, mode="bilinear", align_corners=True), dpt_block.py:230-235).
//
// Why not the implicit-GEMM kernels: M = 512^2 pixels per view, N = 256, K = 7 * 7 * 4 = 196 is a SHORT-K product whose 268 MB output
// dominates; on the ping-pong tiles a round costs launch + ring fill + a 262 KB epilogue for 13 K steps of work (218-230 us per view,
// 151 without the upsample-add: tools/mb_stem2.py), and the 49 taps of 16 bytes arrive as 16-byte LDS-DMA pieces.  Here:
//   * a workgroup (8 waves) owns a 16 x 8 pixel tile and ALL 256 channels; wave w owns channels [32 w, 32 w + 32) of the 128 pixels
//     (64 accumulator + 112 weight registers: two waves per SIMD without spills; a 16 x 16 tile spilled 525 registers; two 4-wave
//     workgroups per CU, each half the channels, measured slower);
//   * the 14 x 23 pixel input patch is loaded ONCE per tile (16-byte pixels, coalesced rows), split into hi / lo bf16 planes when it is
//     stored to LDS (hi = the upper 16 bits, lo = bf16(a - hi): the split of every bf16x3 product of the library) and read back as MFMA A
//     fragments: with K ordered (ky, kx in 0..7, c in 0..3) -- kx = 7 and c = 3 carry zero weights -- the 8 consecutive k of a lane are
//     two neighbouring patch pixels = 16 contiguous bytes;
//   * the wave's W fragments (14 K steps x hi / lo: 112 registers) are loaded once per workgroup and stay in registers: persistent
//     workgroups walk the tiles of their head, no weight traffic after the first tile, the next tile's patch is in flight during the MFMAs;
//   * the 6 x 10 source pixels of path_1 a tile's upsample can touch (60 KB) travel global -> LDS by LDS-DMA (one 1 KiB instruction per
//     source pixel, no registers) while the MFMAs run; read per output pixel from global memory (four 128-byte corners behind each other,
//     no registers left to run ahead) the upsample-add cost as much as the convolution;
//   * the epilogue transposes a 32-pixel x 32-channel block through wave-private LDS so that a lane holds 4 consecutive channels of a pixel:
//     bias, ReLU, the four bilinear corners of path_1 as 16-byte loads (8 lanes = one 128-byte line), one 16-byte store (fp32 output) or
//     two 8-byte stores (pre-split planes [hi 32 | lo 32] per 32 channels: the A operand form of the ping-pong GEMM that reads this map).
// Arithmetic: C = A_hi W_hi + A_lo W_hi + A_hi W_lo in fp32 MFMA accumulators, like every bf16x3 GEMM here; the K order (hence the fp32
// summation order) differs from the implicit-GEMM kernels', so results agree to fp32 rounding, not bit for bit.
#include "common.h"

namespace {

constexpr int TS = 16;           // tile width (pixels) and the granularity of H, W
constexpr int MB = 4;            // 32-pixel blocks (two tile rows each) per tile: a tile is 16 x 8 pixels, 64 accumulator registers per wave
constexpr int TH = 2 * MB;       // tile height
constexpr int PW = 24, PH = TH + 6;  // patch: TH + 6 rows, 16 + 7 columns (the zero-weight tap kx = 7 reads one column further) padded to 24
constexpr int NPP = PW * PH;     // 336 patch pixels
constexpr int KS = 14;           // K steps of 16: k = ky * 32 + kx * 4 + c
constexpr int LDS_ST = 36;       // floats per staged pixel row (32 channels + 4: 16-byte aligned rows)
constexpr int UR = 6, UC = 10;   // rows / columns of the x2 source map a 16 x 8 tile can touch (align_corners: scale < 1/2)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct StemParams {
  const float* img;    // [Z, H, W, 4] fp32 (RGB + one zero channel)
  const uint4* wfrag;  // [G][8 waves][14 steps][2 planes][64 lanes] x 8 bf16: MFMA B fragments (ops.pack_stem7)
  const float* bias;   // [G, 256] or null
  const float* up;     // [Z, H/2, W/2, 256] fp32 or null
  float* out;          // [Z, H, W, 256]: fp32 values (planes == 0) or pre-split planes of the same size and strides (planes == 1)
  int B, G, H, W, planes;
};

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const uint32_t u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
  hi.x = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  hi.y = __builtin_amdgcn_perm(u3, u2, 0x07060302u);
  lo.x = pack_bf16x2(v.x - __uint_as_float(u0 & 0xffff0000u), v.y - __uint_as_float(u1 & 0xffff0000u));
  lo.y = pack_bf16x2(v.z - __uint_as_float(u2 & 0xffff0000u), v.w - __uint_as_float(u3 & 0xffff0000u));
}

// LDS-only workgroup barrier: __syncthreads() also waits for the global stores in flight (a workgroup-scope release), which would put the
// epilogue's stores back in front of the next tile's MFMAs
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512) void stem7_x3_kernel(const StemParams p) {
#if __HIP_DEVICE_COMPILE__
  __shared__ uint2 s_hi[NPP], s_lo[NPP];
  __shared__ __attribute__((aligned(16))) float4 s_raw[384];  // the NEXT tile's patch as it arrives (fp32 pixels, LDS-DMA)
  __shared__ float s_stage[8][32 * LDS_ST];
  __shared__ __attribute__((aligned(16))) float s_up[UR * UC][256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.y;
  const int tx_n = p.W / TS, ty_n = p.H / TH, per_img = tx_n * ty_n, ntiles = p.B * per_img;

  // ---- this wave's W fragments: 14 steps x (hi, lo), resident for the whole kernel
  bf16x8 bh[KS], bl[KS];
  {
    const uint4* wf = p.wfrag + ((size_t)(g * 8 + wave) * KS * 2) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bh[s] = as_bf16x8(wf[(2 * s) * 64]);
      bl[s] = as_bf16x8(wf[(2 * s + 1) * 64]);
    }
  }
  const float* bias = p.bias ? p.bias + g * 256 + wave * 32 : nullptr;

  // Patch of a tile: global -> s_raw by LDS-DMA, one 16-byte pixel per lane, waves 0..5 (a pixel outside the image = an out-of-range
  // offset = zeros).  Every VMEM wait of the loop sits in front of the epilogue, where the only recent VMEM operations are the DMA pieces
  // issued a whole MFMA phase earlier -- the epilogue's stores are never waited for behind their own tile (a register prefetch of the
  // next patch was: gfx9 counts loads and stores in one vmcnt, and MFMAs + stores ran back to back, 11.7 us per tile for 5 + 6.7).
  auto fetch_patch = [&](int tile) {
    if (wave * 64 >= NPP) return;
    const int b = tile / per_img, r = tile - b * per_img, ty = r / tx_n, tx = r - ty * tx_n;
    const int z = b * p.G + g;
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)(p.img + (int64_t)z * p.H * p.W * 4), (short)0, p.H * p.W * 16, 0x00020000);
    const int i = wave * 64 + lane, py = i / PW, pc = i - py * PW;
    const int gy = ty * TH - 3 + py, gx = tx * TS - 3 + pc;
    const bool in = i < NPP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ri, (lds_ptr_t)&s_raw[wave * 64], 16, in ? (unsigned)((gy * p.W + gx) * 16) : 0x80000000u, 0, 0, 0);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch_patch(tile);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int l31 = lane & 31, lh = lane >> 5;
  const int a_off = ((l31 >> 4) * PW + (l31 & 15) + 2 * lh);  // patch pixel of (m-block 0, ky 0, kx0 = 2 * chunk)
  const int sh = p.H >> 1, sw = p.W >> 1;
  const float ry = p.H > 1 ? (float)(sh - 1) / (float)(p.H - 1) : 0.f, rx = p.W > 1 ? (float)(sw - 1) / (float)(p.W - 1) : 0.f;

  for (; tile < ntiles; tile += gridDim.x) {
    lds_barrier();  // the patch in s_raw is complete (its DMA was waited for in front of the previous epilogue); planes and s_up are free
    if (t < NPP) {    // raw pixels -> hi / lo planes
      uint2 h, l;
      split4(s_raw[t], h, l);
      s_hi[t] = h;
      s_lo[t] = l;
    }
    lds_barrier();
    const int nxt = tile + gridDim.x;
    if (nxt < ntiles) fetch_patch(nxt);  // the next tile's pixels travel during the MFMAs
    const int b = tile / per_img, rr = tile - b * per_img, ty = rr / tx_n, tx = rr - ty * tx_n;
    const int z = b * p.G + g;
    const int yb = (int)(ry * (ty * TH)), xb = (int)(rx * (tx * TS));  // first source row / column of the tile
    if (p.up) {  // so do the source pixels of its upsample: one 1 KiB LDS-DMA instruction each
      const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(p.up + (int64_t)z * sh * sw * 256), (short)0, sh * sw * 1024, 0x00020000);
      for (int j = wave; j < UR * UC; j += 8) {
        const int jy = j / UC, jx = j - jy * UC;
        const int ys = min(yb + jy, sh - 1), xs = min(xb + jx, sw - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)&s_up[j][0], 16, (unsigned)((ys * sw + xs) * 1024 + lane * 16), 0, 0, 0);
      }
    }

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int so = (s >> 1) * PW + 4 * (s & 1);  // tap row ky = s / 2, first tap kx0 = 4 * (s & 1) + 2 * chunk
      __builtin_amdgcn_sched_barrier(0);  // a step's fragment reads stay in the step (hoisted together they spill)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int o = a_off + so + 2 * mb * PW;
        const uint2 h0 = s_hi[o], h1 = s_hi[o + 1], l0 = s_lo[o], l1 = s_lo[o + 1];
        const bf16x8 ah = as_bf16x8(make_uint4(h0.x, h0.y, h1.x, h1.y)), al = as_bf16x8(make_uint4(l0.x, l0.y, l1.x, l1.y));
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s], acc[mb], 0, 0, 0);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s], acc[mb], 0, 0, 0);
      }
    }

    // ---- epilogue: per 32-pixel block, transpose through the wave's own LDS, then a lane = (pixel, 4 channels)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (next patch, upsample source) have landed ...
    lds_barrier();                                    // ... and everybody else's
    float* st = s_stage[wave];
    const int prow = lane >> 3, c4 = 4 * (lane & 7);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *(const f32x4*)(bias + c4);
    // a lane's pixels of a block: row (k >> 1) of the block's two tile rows, column 8 (k & 1) + prow: the column terms of the upsample
    // (two per lane) and the lane's part of the addresses are computed once per tile, the row terms are wave-uniform
    int xo[2][2];
    float lxv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ox = tx * TS + 8 * h + prow;
      const float fx = rx * ox;
      const int x0 = (int)fx, x1 = x0 + (x0 < sw - 1 ? 1 : 0);
      lxv[h] = fx - x0;
      xo[h][0] = (x0 - xb) * 256;
      xo[h][1] = (x1 - xb) * 256;
    }
    const float* ub = &s_up[0][wave * 32 + c4];
    float* ob = p.out + (((int64_t)z * p.H + ty * TH) * p.W + tx * TS + prow) * 256 + wave * 32 + c4;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * lh) * LDS_ST + l31] = acc[mb][r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int m = 8 * k + prow;  // pixel of the block: tile row 2 mb + (k >> 1), column 8 (k & 1) + prow
        f32x4 v = *(const f32x4*)(st + m * LDS_ST + c4);
        const int trow = 2 * mb + (k >> 1), h = k & 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bv[e], 0.f);
        if (p.up) {
          const int oy = ty * TH + trow;  // (wave-uniform)
          const float fy = ry * oy;
          const int y0 = (int)fy, y1 = y0 + (y0 < sh - 1 ? 1 : 0);
          const float ly = fy - y0, lx = lxv[h];
          const float *u0 = ub + (y0 - yb) * (UC * 256), *u1 = ub + (y1 - yb) * (UC * 256);
          const f32x4 a_ = *(const f32x4*)(u0 + xo[h][0]), b_ = *(const f32x4*)(u0 + xo[h][1]);
          const f32x4 c_ = *(const f32x4*)(u1 + xo[h][0]), d_ = *(const f32x4*)(u1 + xo[h][1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (1.f - ly) * ((1.f - lx) * a_[e] + lx * b_[e]) + ly * ((1.f - lx) * c_[e] + lx * d_[e]);
        }
        float* op = ob + ((int64_t)trow * p.W + 8 * h) * 256;
        if (p.planes) {  // one 128-byte line per pixel and 32 channels: [hi 32 | lo 32] bf16 (op points c4 floats into the line)
          uint2 hh, ll;
          split4(make_float4(v[0], v[1], v[2], v[3]), hh, ll);
          *(uint2*)((unsigned char*)op - 2 * c4) = hh;
          *(uint2*)((unsigned char*)op + 64 - 2 * c4) = ll;
        } else {
          *(f32x4*)op = v;
        }
      }
    }
  }
#endif
}

}  // namespace

extern "C" int siu3r_stem7x7_x3(const float* img, const void* wfrag, const float* bias, const float* up_src, float* out, int B, int G, int H, int W,
                                int planes_out, void* stream) {
  SIU3R_CHECK(img && wfrag && out, "stem7x7_x3: null pointer");
  SIU3R_CHECK(B > 0 && G > 0 && H > 0 && W > 0 && H % TS == 0 && W % TS == 0, "stem7x7_x3: H and W must be multiples of 16 (H=%d W=%d)", H, W);
  SIU3R_CHECK((int64_t)H * W * 1024 / 4 < 0x7fffffffll, "stem7x7_x3: image too large for 32-bit buffer offsets");
  SIU3R_CHECK(up_src == nullptr || (H % 2 == 0 && W % 2 == 0), "stem7x7_x3: the x2 upsample source needs even H, W");
  StemParams p{img, (const uint4*)wfrag, bias, up_src, out, B, G, H, W, planes_out ? 1 : 0};
  const int ntiles = B * (H / TH) * (W / TS);
  int nwg = 256 / G;  // one workgroup per CU (8 waves x ~250 registers); the heads share the chip
  if (nwg < 1) nwg = 1;
  if (nwg > ntiles) nwg = ntiles;
  hipLaunchKernelGGL(stem7_x3_kernel, dim3(nwg, G), dim3(512), 0, (hipStream_t)stream, p);
  SIU3R_LAUNCH_CHECK("siu3r_stem7x7_x3");
  return 0;
}
