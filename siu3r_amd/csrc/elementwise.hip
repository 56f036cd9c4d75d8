// HBM-bound kernels of the SIU3R path: RoPE2D (seam 1), LayerNorm, GroupNorm, bilinear resize,
// deformable-attention sampling, depth-wise conv, Gaussian adapter, ...   All channel-last, vectorised
// 16 B per lane where the layout allows, one pass over the data per kernel.
#include "common.h"

namespace {

// ---- 4-element vector access of runtime dtype --------------------------------------------------
struct f32x4v {
  float v[4];
};
__device__ __forceinline__ f32x4v load4(const void* p, int dtype, int64_t i) {
  f32x4v r;
  if (dtype == SIU3R_F32) {
    float4 a = *(const float4*)((const float*)p + i);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  } else {
    uint2 a = *(const uint2*)((const u16*)p + i);
    r.v[0] = bf16_bits_to_f32((u16)(a.x & 0xffff));
    r.v[1] = bf16_bits_to_f32((u16)(a.x >> 16));
    r.v[2] = bf16_bits_to_f32((u16)(a.y & 0xffff));
    r.v[3] = bf16_bits_to_f32((u16)(a.y >> 16));
  }
  return r;
}
__device__ __forceinline__ void store4(void* p, int dtype, int64_t i, const f32x4v& r) {
  if (dtype == SIU3R_F32) {
    *(float4*)((float*)p + i) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
    uint2 w;
    w.x = (uint32_t)f32_to_bf16_bits(r.v[0]) | ((uint32_t)f32_to_bf16_bits(r.v[1]) << 16);
    w.y = (uint32_t)f32_to_bf16_bits(r.v[2]) | ((uint32_t)f32_to_bf16_bits(r.v[3]) << 16);
    *(uint2*)((u16*)p + i) = w;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ================================ RoPE2D (seam 1) ================================================
// One thread per (token, rotation pair); loops over heads re-using cos/sin, like the reference
// kernel (kernels.cu:17-82): inv_freq = fwd / powf(base, q/Q), angle = pos * inv_freq.
// DT: SIU3R_BF16 / SIU3R_F32 / SIU3R_F16 / SIU3R_F64 -- the reference dispatches float, double and half (kernels.cu:101); the rotation
// runs in fp32 (double tokens: in double, with the fp32 cos / sin of the reference's __sincosf-free path)
template <int DT>
__global__ void rope2d_kernel(void* tokens, const int64_t* pos, int B, int N, int H, int D, int64_t sb, int64_t sn,
                              int64_t sh, float base, float fwd) {
  const int Q = D / 4, half = D / 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * N * half;
  if (idx >= total) return;
  const int pr = (int)(idx % half);
  const int64_t tok = idx / half;
  const int n = (int)(tok % N), b = (int)(tok / N);
  const int axis = pr / Q, q = pr - axis * Q;
  const float inv_freq = fwd / powf(base, q / float(Q));
  const float freq = pos[tok * 2 + axis] * inv_freq;
  const float c = cosf(freq), s = sinf(freq);
  const int64_t e0 = b * sb + n * sn + axis * 2 * Q + q;
  for (int h = 0; h < H; ++h) {
    const int64_t e = e0 + h * sh;
    if (DT == SIU3R_F32) {
      float* p = (float*)tokens + e;
      const float u = p[0], v = p[Q];
      p[0] = u * c - v * s;
      p[Q] = v * c + u * s;
    } else if (DT == SIU3R_F64) {
      double* p = (double*)tokens + e;
      const double u = p[0], v = p[Q];
      p[0] = u * (double)c - v * (double)s;
      p[Q] = v * (double)c + u * (double)s;
    } else if (DT == SIU3R_F16) {
      _Float16* p = (_Float16*)tokens + e;
      const float u = (float)p[0], v = (float)p[Q];
      p[0] = (_Float16)(u * c - v * s);
      p[Q] = (_Float16)(v * c + u * s);
    } else {
      u16* p = (u16*)tokens + e;
      const float u = bf16_bits_to_f32(p[0]), v = bf16_bits_to_f32(p[Q]);
      p[0] = f32_to_bf16_bits(u * c - v * s);
      p[Q] = f32_to_bf16_bits(v * c + u * s);
    }
  }
}

// ================================ LayerNorm ======================================================
// One wave per row; the row lives in registers (C <= 2048); two-pass mean / variance.
// y2 (optional): a second, bf16 copy of the result -- the fp32 one stays the residual stream, the bf16 one feeds the next
// GEMM through the LDS-DMA path (which takes bf16 A) instead of the slower fp32-A kernel.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, void* y, int y_dtype, void* y2, const float* gamma,
                                                        const float* beta, int64_t rows, int C, int64_t ldx,
                                                        int64_t ldy, int64_t ldy2, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float4 v[8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
      v[i] = *(const float4*)(xr + c);
      sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < C) {
      const float4 g = *(const float4*)(gamma + c), bb = *(const float4*)(beta + c);
      f32x4v o;
      o.v[0] = (v[i].x - mean) * rstd * g.x + bb.x;
      o.v[1] = (v[i].y - mean) * rstd * g.y + bb.y;
      o.v[2] = (v[i].z - mean) * rstd * g.z + bb.z;
      o.v[3] = (v[i].w - mean) * rstd * g.w + bb.w;
      store4(y, y_dtype, row * ldy + c, o);
      if (y2) store4(y2, SIU3R_BF16, row * ldy2 + c, o);
    }
  }
}

// ================================ add (row broadcast) ===========================================
__global__ void add_kernel(const float* a, const float* b, float* y, int64_t rows, int64_t b_rows, int C4) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C4) return;
  const int64_t r = idx / C4;
  const int c = (int)(idx - r * C4);
  const float4 x = ((const float4*)a)[idx];
  const float4 z = ((const float4*)b)[(r % b_rows) * C4 + c];
  ((float4*)y)[idx] = make_float4(x.x + z.x, x.y + z.y, x.z + z.z, x.w + z.w);
}

// ================================ image pack NCHW(3) -> NHWC(8) =================================
__global__ void pack_image_kernel(const float* img, void* out, int out_dtype, int N, int H, int W, int cpad) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t hw = (int64_t)H * W;
  if (idx >= (int64_t)N * hw) return;
  const int64_t n = idx / hw, r = idx - n * hw;
  if (cpad == 4) {  // fp32 only: one 16-byte pixel = one chunk of the bf16x3 GEMM's small-cin gather
    float4 o = {img[(n * 3 + 0) * hw + r], img[(n * 3 + 1) * hw + r], img[(n * 3 + 2) * hw + r], 0.f};
    ((float4*)out)[idx] = o;
    return;
  }
  f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v.v[j] = 0.f;
  for (int c = 0; c < 3; ++c) v.v[c] = img[(n * 3 + c) * hw + r];
  store8_from_f32(out, out_dtype, idx * 8, v);
}

// ================================ bilinear resize (NHWC) ========================================
__device__ __forceinline__ void src_index(int o, int in, int out, int align, int& i0, int& i1, float& l1) {
  float src;
  if (align) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = scale * o;
  } else {
    const float scale = (float)in / (float)out;
    src = scale * (o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
}

__global__ void resize_kernel(const void* x, int x_dtype, void* y, int y_dtype, const void* addend, int add_dtype,
                              const float* ch_scale, const float* ch_shift, int N, int IH, int IW, int OH, int OW,
                              int C, int align, int64_t x_bs, int64_t add_bs) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)N * OH * OW * C4;
  if (idx >= total) return;
  const int c = (int)(idx % C4) * 4;
  int64_t r = idx / C4;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const int n = (int)(r / OH);
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(oy, IH, OH, align, y0, y1, ly);
  src_index(ox, IW, OW, align, x0, x1, lx);
  const int64_t base = (int64_t)n * x_bs + c;  // x_bs / add_bs: elements between batch items (a dense map: IH * IW * C)
  const f32x4v v00 = load4(x, x_dtype, base + ((int64_t)y0 * IW + x0) * C);
  const f32x4v v01 = load4(x, x_dtype, base + ((int64_t)y0 * IW + x1) * C);
  const f32x4v v10 = load4(x, x_dtype, base + ((int64_t)y1 * IW + x0) * C);
  const f32x4v v11 = load4(x, x_dtype, base + ((int64_t)y1 * IW + x1) * C);
  const int64_t oidx = (((int64_t)n * OH + oy) * OW + ox) * C + c;
  f32x4v o;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o.v[j] = (1.f - ly) * ((1.f - lx) * v00.v[j] + lx * v01.v[j]) + ly * ((1.f - lx) * v10.v[j] + lx * v11.v[j]);
  if (addend) {
    const f32x4v a = load4(addend, add_dtype, (int64_t)n * add_bs + ((int64_t)oy * OW + ox) * C + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] += a.v[j];
  }
  if (ch_scale) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = o.v[j] * ch_scale[c + j] + ch_shift[c + j];
  }
  store4(y, y_dtype, oidx, o);
}

// rows_pb: rows per batch item; x_bs / add_bs: elements between batch items of x / addend (dense: rows_pb * C); y is dense
__global__ void affine_add_kernel(const void* x, int x_dtype, const void* addend, int add_dtype, void* y, int y_dtype,
                                  const float* ch_scale, const float* ch_shift, int64_t rows, int C, int64_t rows_pb, int64_t x_bs, int64_t add_bs) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C4) return;
  const int c = (int)(idx % C4) * 4;
  const int64_t row = idx / C4, n = row / rows_pb, r = row - n * rows_pb;
  f32x4v o = load4(x, x_dtype, n * x_bs + r * C + c);
  if (addend) {
    const f32x4v a = load4(addend, add_dtype, n * add_bs + r * C + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] += a.v[j];
  }
  if (ch_scale) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = o.v[j] * ch_scale[c + j] + ch_shift[c + j];
  }
  store4(y, y_dtype, idx * 4, o);
}

// ================================ max-pool 3x3 s2 p1 (NHWC) =====================================
__global__ void maxpool_kernel(const void* x, void* y, int dtype, int N, int IH, int IW, int OH, int OW, int C) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * OH * OW * C4) return;
  const int c = (int)(idx % C4) * 4;
  int64_t r = idx / C4;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const int n = (int)(r / OH);
  f32x4v m;
#pragma unroll
  for (int j = 0; j < 4; ++j) m.v[j] = -INFINITY;
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = oy * 2 - 1 + dy;
    if (iy < 0 || iy >= IH) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = ox * 2 - 1 + dx;
      if (ix < 0 || ix >= IW) continue;
      const f32x4v v = load4(x, dtype, (((int64_t)n * IH + iy) * IW + ix) * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) m.v[j] = fmaxf(m.v[j], v.v[j]);
    }
  }
  store4(y, dtype, idx * 4, m);
}

// ================================ max-pool 2x2 s2 (NHWC, floor) ==================================
// VGG16's pooling (the LPIPS feature stack, torchmetrics functional/image/lpips.py: torchvision vgg16().features indices 4, 9, 16, 23)
__global__ void maxpool2x2_kernel(const void* x, void* y, int dtype, int N, int IH, int IW, int OH, int OW, int C) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * OH * OW * C4) return;
  const int c = (int)(idx % C4) * 4;
  int64_t r = idx / C4;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const int n = (int)(r / OH);
  const int64_t base = (((int64_t)n * IH + 2 * oy) * IW + 2 * ox) * C + c;
  const f32x4v a = load4(x, dtype, base), b = load4(x, dtype, base + C), d = load4(x, dtype, base + (int64_t)IW * C),
               e = load4(x, dtype, base + (int64_t)IW * C + C);
  f32x4v m;
#pragma unroll
  for (int j = 0; j < 4; ++j) m.v[j] = fmaxf(fmaxf(a.v[j], b.v[j]), fmaxf(d.v[j], e.v[j]));
  store4(y, dtype, idx * 4, m);
}

// ================================ LPIPS layer distance ===========================================
// One feature tap of the perceptual distance (torchmetrics _LPIPS.forward): per pixel, both feature vectors are scaled to unit length
// (x / sqrt(eps + sum_c x^2)), the squared difference is weighted by the learned non-negative 1x1 "lin" weights and summed over the
// channels.  f0, f1 [npix, C] fp32 channel-last, w [C]; dist [npix] (the caller averages over the pixels and adds the five taps).
// One wave per pixel, two passes over its 2 x C values (the second hits L1 / L2).
__global__ __launch_bounds__(256) void lpips_layer_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ w,
                                                          float* __restrict__ dist, int64_t npix, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const float* a = f0 + pix * C;
  const float* b = f1 + pix * C;
  float s0 = 0.f, s1 = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 x = *(const float4*)(a + c), y = *(const float4*)(b + c);
    s0 += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    s1 += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
  }
  const float n0 = sqrtf(eps + s0), n1 = sqrtf(eps + s1);
  float acc = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 x = *(const float4*)(a + c), y = *(const float4*)(b + c), ww = *(const float4*)(w + c);
    const float d0 = x.x / n0 - y.x / n1, d1 = x.y / n0 - y.y / n1, d2 = x.z / n0 - y.z / n1, d3 = x.w / n0 - y.w / n1;
    acc += ww.x * (d0 * d0) + ww.y * (d1 * d1) + ww.z * (d2 * d2) + ww.w * (d3 * d3);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) dist[pix] = acc;
}

// ================================ depth-wise 3x3 + bias + GELU over 3 token scales =============
// tokens [B, 21n, C]: [0,16n) is a (2H x 2W) map, [16n,20n) (H x W), [20n,21n) (H/2 x W/2)
__global__ void dwconv_gelu_kernel(const void* x, void* y, int dtype, const float* w9c, const float* bias, int B,
                                   int H, int W, int C) {
  const int C4 = C >> 2;
  const int n = (H * W) / 4;
  const int ntok = 21 * n;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * ntok * C4) return;
  const int c = (int)(idx % C4) * 4;
  int64_t r = idx / C4;
  const int tok = (int)(r % ntok);
  const int b = (int)(r / ntok);
  int start, hh, ww;
  if (tok < 16 * n) {
    start = 0; hh = 2 * H; ww = 2 * W;
  } else if (tok < 20 * n) {
    start = 16 * n; hh = H; ww = W;
  } else {
    start = 20 * n; hh = H / 2; ww = W / 2;
  }
  const int loc = tok - start;
  const int py = loc / ww, px = loc - py * ww;
  f32x4v acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc.v[j] = bias[c + j];
  for (int dy = 0; dy < 3; ++dy) {
    const int iy = py - 1 + dy;
    if (iy < 0 || iy >= hh) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int ix = px - 1 + dx;
      if (ix < 0 || ix >= ww) continue;
      const f32x4v v = load4(x, dtype, ((int64_t)b * ntok + start + iy * ww + ix) * C + c);
      const float4 wv = *(const float4*)(w9c + (dy * 3 + dx) * C + c);
      acc.v[0] += v.v[0] * wv.x;
      acc.v[1] += v.v[1] * wv.y;
      acc.v[2] += v.v[2] * wv.z;
      acc.v[3] += v.v[3] * wv.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc.v[j] = gelu_erf(acc.v[j]);
  store4(y, dtype, idx * 4, acc);
}

// ================================ MS-deformable attention sampling ==============================
struct MsdShapes {
  int h[4], w[4], start[4];
};
// LT, PT > 0: levels / points known at compile time -- the softmax weights are computed once (the generic form evaluated every expf
// twice), all loops unroll, the corner tests become weights (an out-of-range corner reads a clamped in-range texel with weight 0: same
// sums, no divergent skips) and the compiler issues a point's four 16-byte gathers together.  LT = 0: any L <= 4, any P.
template <int LT, int PT>
__global__ __launch_bounds__(256, 4) void msdeform_kernel(const void* value, int v_dtype, const float* offs_aw, const float* ref, MsdShapes sh,
                                                       void* out, int out_dtype, int B, int S, int Q, int heads, int d, int L, int P, int qb) {
  const int D4 = d >> 2;
  int dc, hd, q, b;
  if (qb > 0) {
    // a workgroup = qb = 256 / D4 CONSECUTIVE queries of ONE head: neighbouring queries sample neighbouring texels of the same 4 d-byte
    // head slice (L1 hits: with the (query, head)-major order below a wave's eight rows were eight heads of one query, no two of which
    // share a line), and with the head as the fastest block index a head's workgroups all land on XCD (head % 8): its L2 holds one
    // head's slice of `value` instead of all of them
    const int ql = (int)threadIdx.x / D4;
    dc = ((int)threadIdx.x - ql * D4) * 4;
    int64_t blk = blockIdx.x;
    hd = (int)(blk % heads);
    blk /= heads;
    const int nqb = (Q + qb - 1) / qb;
    q = (int)(blk % nqb) * qb + ql;
    b = (int)(blk / nqb);
    if (q >= Q || b >= B) return;
  } else {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * Q * heads * D4) return;
    dc = (int)(idx % D4) * 4;
    int64_t r = idx / D4;
    hd = (int)(r % heads);
    r /= heads;
    q = (int)(r % Q);
    b = (int)(r / Q);
  }
  if constexpr (LT > 0) {
    constexpr int LP = LT * PT;
    static_assert(PT == 4, "one 16-byte load per level");
    const float* row = offs_aw + ((int64_t)b * Q + q) * (heads * LP * 3);
    const float* offs = row + hd * LP * 2;
    const float* logit = row + heads * LP * 2 + hd * LP;
    float mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const float4 t = *(const float4*)(logit + 4 * l);
      mx = fmaxf(fmaxf(mx, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
    }
    float den = 0.f;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const float4 t = *(const float4*)(logit + 4 * l);
      den += expf(t.x - mx);
      den += expf(t.y - mx);
      den += expf(t.z - mx);
      den += expf(t.w - mx);
    }
    f32x4v acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[j] = 0.f;
    const int64_t vb = (int64_t)b * S;
#pragma unroll 1
    for (int l = 0; l < LT; ++l) {  // (a level at a time: fully unrolled, the scheduler precomputes every point's addresses and spills)
      const int hh = sh.h[l], ww = sh.w[l], st0 = sh.start[l];
      const float rx = ref[((int64_t)q * LT + l) * 2 + 0], ry = ref[((int64_t)q * LT + l) * 2 + 1];
      const float4 lg = *(const float4*)(logit + 4 * l);
      const float4 oa = *(const float4*)(offs + 8 * l), ob = *(const float4*)(offs + 8 * l + 4);
      const float e4[4] = {lg.x, lg.y, lg.z, lg.w};
      const float o8[8] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
      f32x4v v[4][4];
      float wk[4][4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const float aw = expf(e4[pt] - mx) / den;
        const float locx = rx + o8[pt * 2 + 0] / (float)ww;
        const float locy = ry + o8[pt * 2 + 1] / (float)hh;
        const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
        const float fx = ((gx + 1.f) * ww - 1.f) * 0.5f, fy = ((gy + 1.f) * hh - 1.f) * 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float lx = fx - x0f, ly = fy - y0f;
        const float wgt[4] = {(1.f - lx) * (1.f - ly), lx * (1.f - ly), (1.f - lx) * ly, lx * ly};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
          const bool ok = xx >= 0 && xx < ww && yy >= 0 && yy < hh;
          const int xc = min(max(xx, 0), ww - 1), yc = min(max(yy, 0), hh - 1);
          const int s_ = st0 + yc * ww + xc;
          v[pt][k] = load4(value, v_dtype, ((vb + s_) * heads + hd) * d + dc);
          wk[pt][k] = ok ? wgt[k] * aw : 0.f;
        }
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc.v[j] += v[pt][k].v[j] * wk[pt][k];
    }
    store4(out, out_dtype, (((int64_t)b * Q + q) * heads + hd) * d + dc, acc);
  } else {
    const int LP = L * P;
    const float* row = offs_aw + ((int64_t)b * Q + q) * (heads * LP * 3);
    const float* offs = row + hd * LP * 2;
    const float* logit = row + heads * LP * 2 + hd * LP;
    float mx = -INFINITY;
    for (int i = 0; i < LP; ++i) mx = fmaxf(mx, logit[i]);
    float den = 0.f;
    for (int i = 0; i < LP; ++i) den += expf(logit[i] - mx);
    f32x4v acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[j] = 0.f;
    for (int l = 0; l < L; ++l) {
      const int hh = sh.h[l], ww = sh.w[l];
      const float rx = ref[((int64_t)q * L + l) * 2 + 0], ry = ref[((int64_t)q * L + l) * 2 + 1];
      for (int pt = 0; pt < P; ++pt) {
        const int i = l * P + pt;
        const float aw = expf(logit[i] - mx) / den;
        const float locx = rx + offs[i * 2 + 0] / (float)ww;
        const float locy = ry + offs[i * 2 + 1] / (float)hh;
        // grid_sample(align_corners=False): pixel = ((2*loc-1 + 1) * size - 1) / 2
        const float gx = 2.f * locx - 1.f, gy = 2.f * locy - 1.f;
        const float fx = ((gx + 1.f) * ww - 1.f) * 0.5f, fy = ((gy + 1.f) * hh - 1.f) * 0.5f;
        const float x0f = floorf(fx), y0f = floorf(fy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float lx = fx - x0f, ly = fy - y0f;
        const float wgt[4] = {(1.f - lx) * (1.f - ly), lx * (1.f - ly), (1.f - lx) * ly, lx * ly};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
          if (xx < 0 || xx >= ww || yy < 0 || yy >= hh) continue;
          const int64_t s = sh.start[l] + (int64_t)yy * ww + xx;
          const f32x4v v = load4(value, v_dtype, (((int64_t)b * S + s) * heads + hd) * d + dc);
          const float wk = wgt[k] * aw;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc.v[j] += v.v[j] * wk;
        }
      }
    }
    store4(out, out_dtype, (((int64_t)b * Q + q) * heads + hd) * d + dc, acc);
  }
}

// ================================ GroupNorm (NHWC) ==============================================
// stats: one block per (n, group); cg = C/groups channels (multiple of 4).  1024 threads, two 16-byte loads in flight per thread: the 64
// blocks of a pair (2 items x 32 groups) each stream 2 MB of 32-byte pieces at a 1 KB stride -- with 256 threads and one load in
// flight a block moved 33 GB/s and the kernel took 64 us for 134 MB (round 6)
constexpr int GN_T = 1024;
__global__ __launch_bounds__(GN_T) void gn_stats_kernel(const void* x, int x_dtype, float* stats, int HW, int C,
                                                        int groups, float eps) {
  // workgroups go to the XCDs round robin (linear id % 8) and every XCD has its own L2: neighbouring groups share 128-byte lines (a group
  // of the network is 8 channels = 32 bytes of a pixel's 1 KB row), so the groups of one line are given to ONE XCD -- with consecutive
  // groups on consecutive XCDs every L2 fetched every line for a quarter of it
  const int n = blockIdx.y, bx = blockIdx.x;
  const int g = groups % 8 == 0 ? (bx % 8) * (groups / 8) + bx / 8 : bx;
  const int cg = C / groups, cg4 = cg >> 2;
  const int64_t total = (int64_t)HW * cg4;
  double s = 0.0, ss = 0.0;
  auto acc = [&](const f32x4v& v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += v.v[j];
      ss += (double)v.v[j] * v.v[j];
    }
  };
  int64_t i = threadIdx.x;
  if (GN_T % cg4 == 0) {
    // the block's threads cover whole pixels (the network's cg4 = 2): a thread keeps its quad and walks pixels at a constant stride
    const int64_t step = (int64_t)(GN_T / cg4) * C;
    int64_t off = (int64_t)n * HW * C + g * cg + (int)(threadIdx.x % cg4) * 4 + (int64_t)(threadIdx.x / cg4) * C;
    for (; i + GN_T < total; i += 2 * GN_T, off += 2 * step) {
      const f32x4v v = load4(x, x_dtype, off), w = load4(x, x_dtype, off + step);
      acc(v);
      acc(w);
    }
    for (; i < total; i += GN_T, off += step) acc(load4(x, x_dtype, off));
  }
  for (; i < total; i += GN_T) {  // (general case: a division per load)
    const int64_t px = i / cg4;
    acc(load4(x, x_dtype, ((int64_t)n * HW + px) * C + g * cg + (int)(i - px * cg4) * 4));
  }
  __shared__ double sh_s[GN_T], sh_ss[GN_T];
  sh_s[threadIdx.x] = s;
  sh_ss[threadIdx.x] = ss;
  __syncthreads();
  for (int o = GN_T / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + o];
      sh_ss[threadIdx.x] += sh_ss[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = (double)HW * cg;
    const double mean = sh_s[0] / cnt;
    double var = sh_ss[0] / cnt - mean * mean;
    if (var < 0) var = 0;
    stats[((int64_t)n * groups + g) * 2 + 0] = (float)mean;
    stats[((int64_t)n * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__global__ void gn_apply_kernel(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma,
                                const float* beta, const float* stats, const void* addend, int add_dtype, int relu,
                                int N, int HW, int C, int groups) {
  const int C4 = C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * HW * C4) return;
  const int c = (int)(idx % C4) * 4;
  const int n = (int)(idx / ((int64_t)HW * C4));
  const int g = c / (C / groups);
  const float mean = stats[((int64_t)n * groups + g) * 2], rstd = stats[((int64_t)n * groups + g) * 2 + 1];
  f32x4v v = load4(x, x_dtype, idx * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float o = (v.v[j] - mean) * rstd * gamma[c + j] + beta[c + j];
    if (relu) o = fmaxf(o, 0.f);
    v.v[j] = o;
  }
  if (addend) {
    const f32x4v a = load4(addend, add_dtype, idx * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v.v[j] += a.v[j];
  }
  store4(y, y_dtype, idx * 4, v);
}

// ================================ pts3d 'exp' post-process ======================================
__global__ void pts3d_kernel(float* xyz, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[i * 3], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
  const float d = sqrtf(x * x + y * y + z * z);
  const float dc = fmaxf(d, 1e-8f);
  const float e = expm1f(d);
  xyz[i * 3] = x / dc * e;
  xyz[i * 3 + 1] = y / dc * e;
  xyz[i * 3 + 2] = z / dc * e;
}

// ================================ Gaussian adapter ==============================================
// 128 Gaussians per block; raw rows (83 floats) are staged through LDS so that global traffic is
// fully coalesced in both directions (row stride 83 words is odd -> conflict-free LDS access).
constexpr int GA_ROWS = 128, GA_D = 83, GA_SH = 75;
__constant__ float c_sh_mask[25];
__global__ __launch_bounds__(256) void gaussian_adapter_kernel(const void* raw, int raw_dtype, float* opac,
                                                               float* scales, float* rots, float* sh, float* cov,
                                                               int64_t n) {
  __shared__ float s[GA_ROWS * GA_D];
  __shared__ float so[GA_ROWS * 17];  // per row: 3 scales, 4 rotations, 9 covariance entries (stride 17: conflict-free), copied out coalesced
  const int64_t g0 = (int64_t)blockIdx.x * GA_ROWS;
  const int rows = (int)((n - g0) < GA_ROWS ? (n - g0) : GA_ROWS);
  // 256 threads move the rows (16 bytes per lane on full fp32 blocks: 128 x 83 floats start 16-byte aligned), 128 of them compute
  const int nt = blockDim.x;
  if (raw_dtype == SIU3R_F32 && rows == GA_ROWS) {
    const float4* src = (const float4*)((const float*)raw + g0 * GA_D);
    for (int i = threadIdx.x; i < GA_ROWS * GA_D / 4; i += nt) ((float4*)s)[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < rows * GA_D; i += nt) s[i] = load_as_f32(raw, raw_dtype, g0 * GA_D + i);
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < rows) {
    const float* r = s + t * GA_D;
    opac[g0 + t] = 1.f / (1.f + expf(-r[0]));
    float sc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float x = r[1 + j];
      const float sp = x > 20.f ? x : log1pf(expf(x));
      sc[j] = fminf(0.001f * sp, 0.3f);
      so[t * 17 + j] = sc[j];
    }
    const float qi = r[4], qj = r[5], qk = r[6], qr = r[7];
    so[t * 17 + 3] = qi; so[t * 17 + 4] = qj; so[t * 17 + 5] = qk; so[t * 17 + 6] = qr;
    const float nrm = sqrtf(qi * qi + qj * qj + qk * qk + qr * qr) + 1e-8f;
    const float i_ = qi / nrm, j_ = qj / nrm, k_ = qk / nrm, r_ = qr / nrm;
    const float two_s = 2.f / ((i_ * i_ + j_ * j_ + k_ * k_ + r_ * r_) + 1e-8f);
    float R[9];
    R[0] = 1.f - two_s * (j_ * j_ + k_ * k_);
    R[1] = two_s * (i_ * j_ - k_ * r_);
    R[2] = two_s * (i_ * k_ + j_ * r_);
    R[3] = two_s * (i_ * j_ + k_ * r_);
    R[4] = 1.f - two_s * (i_ * i_ + k_ * k_);
    R[5] = two_s * (j_ * k_ - i_ * r_);
    R[6] = two_s * (i_ * k_ - j_ * r_);
    R[7] = two_s * (j_ * k_ + i_ * r_);
    R[8] = 1.f - two_s * (i_ * i_ + j_ * j_);
    // cov = (R S)(R S)^T, same association as R @ S @ S^T @ R^T up to fp32 rounding
    float M[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) M[a * 3 + b] = R[a * 3 + b] * sc[b];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        float v = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) v += (M[a * 3 + k2] * sc[k2]) * R[b * 3 + k2];
        so[t * 17 + 7 + a * 3 + b] = v;
      }
  }
  __syncthreads();
  // (a thread writing its own row's 3 / 4 / 9 floats leaves every store instruction of the wave strided: staged and copied out instead)
  for (int i = threadIdx.x; i < rows * 3; i += nt) scales[g0 * 3 + i] = so[(i / 3) * 17 + i % 3];
  for (int i = threadIdx.x; i < rows * 4; i += nt) rots[g0 * 4 + i] = so[(i >> 2) * 17 + 3 + (i & 3)];
  for (int i = threadIdx.x; i < rows * 9; i += nt) cov[g0 * 9 + i] = so[(i / 9) * 17 + 7 + i % 9];
  // harmonics: [n, 3, 25] = raw[:, 8:83] * mask[d_sh]
  if (rows == GA_ROWS) {  // 128 x 75 floats: 16-byte aligned, a multiple of four
    for (int i4 = threadIdx.x; i4 < GA_ROWS * GA_SH / 4; i4 += nt) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i4 * 4 + u, rr = i / GA_SH, e = i - rr * GA_SH;
        v[u] = s[rr * GA_D + 8 + e] * c_sh_mask[e % 25];
      }
      ((float4*)(sh + g0 * GA_SH))[i4] = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    for (int i = threadIdx.x; i < rows * GA_SH; i += nt) {
      const int rr = i / GA_SH, e = i - rr * GA_SH;
      sh[g0 * GA_SH + i] = s[rr * GA_D + 8 + e] * c_sh_mask[e % 25];
    }
  }
}

// ================================ Mask2Former attention mask ====================================
__global__ void m2f_mask_kernel(const float* ml, uint8_t* out, int32_t* row_counts, int B, int T, int IH, int IW,
                                int OH, int OW, int Q, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * OH * OW * Q;
  if (idx >= total) return;
  const int q = (int)(idx % Q);
  int64_t r = idx / Q;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  r /= OH;
  const int t = (int)(r % T);
  const int b = (int)(r / T);
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(oy, IH, OH, 0, y0, y1, ly);
  src_index(ox, IW, OW, 0, x0, x1, lx);
  const int64_t base = ((int64_t)b * T + t) * IH * IW;
  const float v00 = ml[(base + (int64_t)y0 * IW + x0) * Q + q], v01 = ml[(base + (int64_t)y0 * IW + x1) * Q + q];
  const float v10 = ml[(base + (int64_t)y1 * IW + x0) * Q + q], v11 = ml[(base + (int64_t)y1 * IW + x1) * Q + q];
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  const float sg = 1.f / (1.f + expf(-v));
  const uint8_t blocked = sg < 0.5f ? 1 : 0;
  const int64_t nk = (int64_t)T * OH * OW;
  out[((int64_t)b * Q + q) * ld + ((int64_t)t * OH + oy) * OW + ox] = blocked;
  // a row whose keys are ALL blocked is un-blocked afterwards (video_seg_decoder.py:1474-1477): "some key is open" needs no count and
  // no atomic -- every open key stores the same 1 (one same-address atomic per blocked key was ~2.4 ns each: most of this kernel)
  if (!blocked) row_counts[b * Q + q] = 1;
}
__global__ void zero_i32_kernel(int32_t* p, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) p[idx] = 0;
}
__global__ void m2f_mask_fix_kernel(uint8_t* out, const int32_t* row_counts, int64_t rows, int64_t nk, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * nk) return;
  const int64_t r = idx / nk;
  if (row_counts[r] == 0) out[r * ld + (idx - r * nk)] = 0;  // no open key in this row
}

// ================================ fp32 -> bf16 hi (+lo) planes, K zero-padded ====================
// x3 (optional): both planes interleaved per 32-deep K tile, [row][kpad / 32][hi 32 | lo 32] -- the W operand of the bf16x3 LDS-DMA GEMM
__global__ void split_bf16_kernel(const float* x, u16* hi, u16* lo, u16* x3, int64_t rows, int k, int kpad, int64_t ldx) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * kpad) return;
  const int64_t r = idx / kpad;
  const int c = (int)(idx - r * kpad);
  const float v = c < k ? x[r * ldx + c] : 0.f;
  const u16 h = f32_to_bf16_bits(v);
  const u16 l = f32_to_bf16_bits(v - bf16_bits_to_f32(h));
  if (hi) hi[idx] = h;
  if (lo) lo[idx] = l;
  if (x3) {
    const int64_t o = r * 2 * kpad + (c >> 5) * 64 + (c & 31);
    x3[o] = h;
    x3[o + 32] = l;
  }
}

inline dim3 grid1d(int64_t total, int block = 256) { return dim3((unsigned)cdiv64(total, block)); }

}  // namespace

// ================================ C ABI ==========================================================
extern "C" int siu3r_rope2d(void* tokens, int dtype, int B, int N, int H, int D, int64_t sb, int64_t sn, int64_t sh,
                            const int64_t* positions, float base, float fwd, void* stream) {
  SIU3R_CHECK(D % 4 == 0, "token dim must be multiple of 4");  // kernels.cu:94
  SIU3R_CHECK(dtype == SIU3R_F32 || dtype == SIU3R_BF16 || dtype == SIU3R_F16 || dtype == SIU3R_F64, "rope_2d: unsupported dtype %d", dtype);
  SIU3R_CHECK(B >= 0 && N >= 0 && H >= 0, "rope_2d: negative size");
  const int64_t total = (int64_t)B * N * (D / 2);
  if (total == 0 || H == 0) return 0;  // empty launch is a no-op, like a 0-block CUDA launch
  SIU3R_CHECK(tokens && positions, "rope_2d: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == SIU3R_F32)
    hipLaunchKernelGGL(rope2d_kernel<SIU3R_F32>, grid1d(total), dim3(256), 0, s, tokens, positions, B, N, H, D, sb, sn, sh, base, fwd);
  else if (dtype == SIU3R_F64)
    hipLaunchKernelGGL(rope2d_kernel<SIU3R_F64>, grid1d(total), dim3(256), 0, s, tokens, positions, B, N, H, D, sb, sn, sh, base, fwd);
  else if (dtype == SIU3R_F16)
    hipLaunchKernelGGL(rope2d_kernel<SIU3R_F16>, grid1d(total), dim3(256), 0, s, tokens, positions, B, N, H, D, sb, sn, sh, base, fwd);
  else
    hipLaunchKernelGGL(rope2d_kernel<SIU3R_BF16>, grid1d(total), dim3(256), 0, s, tokens, positions, B, N, H, D, sb, sn, sh, base, fwd);
  SIU3R_LAUNCH_CHECK("siu3r_rope2d");
  return 0;
}

extern "C" int siu3r_layernorm(const float* x, void* y, int y_dtype, const float* gamma, const float* beta,
                               int64_t rows, int C, int64_t ldx, int64_t ldy, float eps, void* stream) {
  SIU3R_CHECK(x && y && gamma && beta, "layernorm: null pointer");
  SIU3R_CHECK(C % 4 == 0 && C <= 2048 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: C=%d must be a multiple of 4 and <= 2048", C);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, y_dtype, (void*)nullptr, gamma, beta, rows, C, ldx, ldy, (int64_t)0, eps);
  SIU3R_LAUNCH_CHECK("siu3r_layernorm");
  return 0;
}

extern "C" int siu3r_layernorm2(const float* x, void* y, int y_dtype, void* y2_bf16, const float* gamma, const float* beta, int64_t rows,
                                int C, int64_t ldx, int64_t ldy, int64_t ldy2, float eps, void* stream) {
  SIU3R_CHECK(x && y && y2_bf16 && gamma && beta, "layernorm2: null pointer");
  SIU3R_CHECK(C % 4 == 0 && C <= 2048 && ldx % 4 == 0 && ldy % 4 == 0 && ldy2 % 4 == 0, "layernorm2: C=%d must be a multiple of 4 and <= 2048", C);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, y_dtype, y2_bf16, gamma, beta, rows, C, ldx, ldy, ldy2, eps);
  SIU3R_LAUNCH_CHECK("siu3r_layernorm2");
  return 0;
}

extern "C" int siu3r_add(const float* a, const float* b, float* y, int64_t rows, int64_t b_rows, int C, void* stream) {
  SIU3R_CHECK(a && b && y && C % 4 == 0 && b_rows > 0, "add: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(add_kernel, grid1d(rows * (C / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, rows, b_rows, C / 4);
  SIU3R_LAUNCH_CHECK("siu3r_add");
  return 0;
}

extern "C" int siu3r_pack_image_nhwc(const float* img, void* out, int out_dtype, int N, int H, int W, int cpad, void* stream) {
  SIU3R_CHECK(img && out, "pack_image: null pointer");
  SIU3R_CHECK(cpad == 8 || (cpad == 4 && out_dtype == SIU3R_F32), "pack_image: 8 channels (bf16 / fp32) or 4 channels (fp32) per pixel, got %d", cpad);
  hipLaunchKernelGGL(pack_image_kernel, grid1d((int64_t)N * H * W), dim3(256), 0, (hipStream_t)stream, img, out, out_dtype, N, H, W, cpad);
  SIU3R_LAUNCH_CHECK("siu3r_pack_image_nhwc");
  return 0;
}

extern "C" int siu3r_resize_bilinear_strided(const void* x, int x_dtype, void* y, int y_dtype, const void* addend, int add_dtype,
                                             const float* ch_scale, const float* ch_shift, int N, int IH, int IW, int OH, int OW, int C,
                                             int align_corners, int64_t x_batch_stride, int64_t addend_batch_stride, void* stream) {
  SIU3R_CHECK(x && y && C % 4 == 0, "resize_bilinear: bad arguments (C=%d)", C);
  SIU3R_CHECK((ch_scale == nullptr) == (ch_shift == nullptr), "resize_bilinear: scale/shift must come together");
  SIU3R_CHECK(x_batch_stride >= (int64_t)IH * IW * C && x_batch_stride % 4 == 0 && (!addend || (addend_batch_stride >= (int64_t)OH * OW * C && addend_batch_stride % 4 == 0)),
              "resize_bilinear: batch strides must cover one map and keep 4-element alignment");
  hipLaunchKernelGGL(resize_kernel, grid1d((int64_t)N * OH * OW * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, addend, add_dtype, ch_scale, ch_shift, N, IH, IW, OH, OW, C, align_corners, x_batch_stride, addend_batch_stride);
  SIU3R_LAUNCH_CHECK("siu3r_resize_bilinear");
  return 0;
}

extern "C" int siu3r_resize_bilinear(const void* x, int x_dtype, void* y, int y_dtype, const void* addend,
                                     int add_dtype, const float* ch_scale, const float* ch_shift, int N, int IH,
                                     int IW, int OH, int OW, int C, int align_corners, void* stream) {
  return siu3r_resize_bilinear_strided(x, x_dtype, y, y_dtype, addend, add_dtype, ch_scale, ch_shift, N, IH, IW, OH, OW, C, align_corners,
                                       (int64_t)IH * IW * C, (int64_t)OH * OW * C, stream);
}

extern "C" int siu3r_affine_add_strided(const void* x, int x_dtype, const void* addend, int add_dtype, void* y, int y_dtype, const float* ch_scale,
                                        const float* ch_shift, int64_t rows, int C, int64_t rows_per_batch, int64_t x_batch_stride,
                                        int64_t addend_batch_stride, void* stream) {
  SIU3R_CHECK(x && y && C % 4 == 0 && rows_per_batch > 0 && rows % rows_per_batch == 0, "affine_add: bad arguments");
  SIU3R_CHECK(x_batch_stride >= rows_per_batch * C && x_batch_stride % 4 == 0 && (!addend || (addend_batch_stride >= rows_per_batch * C && addend_batch_stride % 4 == 0)),
              "affine_add: batch strides must cover one batch item and keep 4-element alignment");
  hipLaunchKernelGGL(affine_add_kernel, grid1d(rows * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, addend, add_dtype, y, y_dtype, ch_scale, ch_shift, rows, C,
                     rows_per_batch, x_batch_stride, addend_batch_stride);
  SIU3R_LAUNCH_CHECK("siu3r_affine_add");
  return 0;
}

extern "C" int siu3r_affine_add(const void* x, int x_dtype, const void* addend, int add_dtype, void* y, int y_dtype,
                                const float* ch_scale, const float* ch_shift, int64_t rows, int C, void* stream) {
  return siu3r_affine_add_strided(x, x_dtype, addend, add_dtype, y, y_dtype, ch_scale, ch_shift, rows, C, rows, rows * C, rows * C, stream);
}

extern "C" int siu3r_maxpool3x3s2(const void* x, void* y, int dtype, int N, int IH, int IW, int C, void* stream) {
  SIU3R_CHECK(x && y && C % 4 == 0, "maxpool: bad arguments");
  const int OH = (IH + 2 - 3) / 2 + 1, OW = (IW + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_kernel, grid1d((int64_t)N * OH * OW * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, dtype, N, IH, IW, OH, OW, C);
  SIU3R_LAUNCH_CHECK("siu3r_maxpool3x3s2");
  return 0;
}

extern "C" int siu3r_maxpool2x2s2(const void* x, void* y, int dtype, int N, int IH, int IW, int C, void* stream) {
  SIU3R_CHECK(x && y && C % 4 == 0 && IH >= 2 && IW >= 2, "maxpool2x2s2: bad arguments");
  const int OH = IH / 2, OW = IW / 2;
  hipLaunchKernelGGL(maxpool2x2_kernel, grid1d((int64_t)N * OH * OW * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, dtype, N, IH, IW, OH, OW, C);
  SIU3R_LAUNCH_CHECK("siu3r_maxpool2x2s2");
  return 0;
}

extern "C" int siu3r_lpips_layer(const float* f0, const float* f1, const float* w, float* dist, int64_t npix, int C, float eps, void* stream) {
  SIU3R_CHECK(npix == 0 || (f0 && f1 && w && dist), "lpips_layer: null pointer");
  SIU3R_CHECK(C > 0 && C % 4 == 0 && (((uintptr_t)f0 | (uintptr_t)f1 | (uintptr_t)w) & 15) == 0, "lpips_layer: C = %d must be a multiple of 4, pointers 16-byte aligned", C);
  if (npix > 0) hipLaunchKernelGGL(lpips_layer_kernel, dim3((unsigned)cdiv64(npix, 4)), dim3(256), 0, (hipStream_t)stream, f0, f1, w, dist, npix, C, eps);
  SIU3R_LAUNCH_CHECK("siu3r_lpips_layer");
  return 0;
}

extern "C" int siu3r_dwconv3x3_gelu(const void* x, void* y, int dtype, const float* w9c, const float* bias, int B,
                                    int H, int W, int C, void* stream) {
  SIU3R_CHECK(x && y && w9c && bias && C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "dwconv3x3_gelu: bad arguments");
  const int64_t ntok = 21 * (int64_t)(H * W / 4);
  hipLaunchKernelGGL(dwconv_gelu_kernel, grid1d((int64_t)B * ntok * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, y, dtype, w9c, bias, B, H, W, C);
  SIU3R_LAUNCH_CHECK("siu3r_dwconv3x3_gelu");
  return 0;
}

extern "C" int siu3r_msdeform_sample(const void* value, int v_dtype, const float* offs_aw, const float* ref,
                                     const int32_t* shapes_host, void* out, int out_dtype, int B, int S, int Q,
                                     int heads, int d, int L, int P, void* stream) {
  SIU3R_CHECK(value && offs_aw && ref && shapes_host && out, "msdeform_sample: null pointer");
  SIU3R_CHECK(L >= 1 && L <= 4 && d % 4 == 0, "msdeform_sample: L=%d (1..4), d=%d (%%4)", L, d);
  MsdShapes sh;
  int start = 0;
  for (int l = 0; l < L; ++l) {
    sh.h[l] = shapes_host[2 * l];
    sh.w[l] = shapes_host[2 * l + 1];
    sh.start[l] = start;
    start += sh.h[l] * sh.w[l];
  }
  SIU3R_CHECK(start == S, "msdeform_sample: spatial shapes sum %d != S=%d", start, S);
  const int D4 = d / 4;
  const int qb = (256 % D4 == 0) ? 256 / D4 : 0;  // queries per workgroup (0: the flat thread order)
  const dim3 grid = qb ? dim3((unsigned)((int64_t)B * ((Q + qb - 1) / qb) * heads)) : grid1d((int64_t)B * Q * heads * D4);
  const bool al = (((uintptr_t)offs_aw) & 15) == 0 && P == 4;  // (rows of heads * L * 4 * 3 floats: every head's slice is 16-byte aligned)
#define SIU3R_MSD(LT_, PT_) hipLaunchKernelGGL((msdeform_kernel<LT_, PT_>), grid, dim3(256), 0, (hipStream_t)stream, value, v_dtype, offs_aw, ref, sh, out, out_dtype, B, S, Q, heads, d, L, P, qb)
  if (al && L == 1) SIU3R_MSD(1, 4);
  else if (al && L == 3) SIU3R_MSD(3, 4);
  else if (al && L == 4) SIU3R_MSD(4, 4);
  else SIU3R_MSD(0, 0);
#undef SIU3R_MSD
  SIU3R_LAUNCH_CHECK("siu3r_msdeform_sample");
  return 0;
}

extern "C" int siu3r_groupnorm(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma,
                               const float* beta, float* stats_ws, const void* addend, int add_dtype, int relu, int N,
                               int HW, int C, int groups, float eps, void* stream) {
  SIU3R_CHECK(x && y && gamma && beta && stats_ws, "groupnorm: null pointer");
  SIU3R_CHECK(C % groups == 0 && (C / groups) % 4 == 0, "groupnorm: C/groups must be a multiple of 4");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(groups, N), dim3(GN_T), 0, s, x, x_dtype, stats_ws, HW, C, groups, eps);
  hipLaunchKernelGGL(gn_apply_kernel, grid1d((int64_t)N * HW * (C / 4)), dim3(256), 0, s, x, x_dtype, y, y_dtype, gamma, beta, stats_ws, addend, add_dtype, relu, N, HW, C, groups);
  SIU3R_LAUNCH_CHECK("siu3r_groupnorm");
  return 0;
}

extern "C" int siu3r_pts3d_exp(float* xyz, int64_t n, void* stream) {
  SIU3R_CHECK(xyz, "pts3d_exp: null pointer");
  hipLaunchKernelGGL(pts3d_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, xyz, n);
  SIU3R_LAUNCH_CHECK("siu3r_pts3d_exp");
  return 0;
}

extern "C" int siu3r_gaussian_adapter(const void* raw, int raw_dtype, float* opacities, float* scales,
                                      float* rotations, float* harmonics, float* covariances, int64_t n,
                                      void* stream) {
  SIU3R_CHECK(raw && opacities && scales && rotations && harmonics && covariances, "gaussian_adapter: null pointer");
  static bool mask_set = false;
  if (!mask_set) {
    float m[25];
    m[0] = 1.f;
    for (int deg = 1; deg <= 4; ++deg) {
      float v = 0.1f;
      for (int i = 0; i < deg; ++i) v *= 0.25f;  // 0.1 * 0.25**deg  (gaussian_adapter.py:70-71)
      for (int i = deg * deg; i < (deg + 1) * (deg + 1); ++i) m[i] = v;
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_sh_mask), m, sizeof(m)) != hipSuccess) {
      siu3r_set_error("gaussian_adapter: cannot upload SH mask");
      return 2;
    }
    mask_set = true;
  }
  hipLaunchKernelGGL(gaussian_adapter_kernel, dim3((unsigned)cdiv64(n, GA_ROWS)), dim3(256), 0, (hipStream_t)stream, raw, raw_dtype, opacities, scales, rotations, harmonics, covariances, n);
  SIU3R_LAUNCH_CHECK("siu3r_gaussian_adapter");
  return 0;
}

extern "C" int siu3r_m2f_attn_mask(const float* mask_logits, uint8_t* out, int32_t* row_counts_ws, int B, int T,
                                   int IH, int IW, int OH, int OW, int Q, int64_t out_ld, void* stream) {
  SIU3R_CHECK(mask_logits && out && row_counts_ws, "m2f_attn_mask: null pointer");
  SIU3R_CHECK(out_ld >= (int64_t)T * OH * OW, "m2f_attn_mask: out_ld too small");
  hipStream_t s = (hipStream_t)stream;
  // (a kernel, not hipMemsetAsync: inside a captured per-chain graph the memset node was observed not to be ordered before the
  // counting kernel on replay -- stale counts from the previous replay survived)
  hipLaunchKernelGGL(zero_i32_kernel, grid1d((int64_t)B * Q), dim3(256), 0, s, row_counts_ws, (int64_t)B * Q);
  const int64_t nk = (int64_t)T * OH * OW;
  hipLaunchKernelGGL(m2f_mask_kernel, grid1d((int64_t)B * nk * Q), dim3(256), 0, s, mask_logits, out, row_counts_ws, B, T, IH, IW, OH, OW, Q, out_ld);
  hipLaunchKernelGGL(m2f_mask_fix_kernel, grid1d((int64_t)B * Q * nk), dim3(256), 0, s, out, row_counts_ws, (int64_t)B * Q, nk, out_ld);
  SIU3R_LAUNCH_CHECK("siu3r_m2f_attn_mask");
  return 0;
}

extern "C" int siu3r_split_bf16(const float* x, void* hi, void* lo, void* x3, int64_t rows, int k, int kpad, int64_t ldx,
                                void* stream) {
  SIU3R_CHECK(x && (hi || x3) && kpad >= k && (!x3 || kpad % 32 == 0), "split_bf16: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(split_bf16_kernel, grid1d(rows * kpad), dim3(256), 0, (hipStream_t)stream, x, (u16*)hi, (u16*)lo, (u16*)x3, rows, k, kpad, ldx);
  SIU3R_LAUNCH_CHECK("siu3r_split_bf16");
  return 0;
}
