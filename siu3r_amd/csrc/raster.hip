// Tile-binned Gaussian splat rasterizer for gfx950 -- a fresh design (not a hipify of either CUDA rasterizer).
//
// Replaces, behind the reference's call sites, the two un-vendored CUDA packages:
//   mode 0 "K2": diff-gaussian-rasterization-w-pose semantics (src/models/cuda_splatting.py:90-118):
//                SH->RGB, colour + depth + accumulated opacity + radii + n_touched, pixel centres at integers
//   mode 1 "K3": gsplat semantics (src/models/gaussian_renderer.py:92-106): N-channel features + alphas,
//                near/far cull, eps2d, pixel centres at +0.5, channels rendered in chunks of 32
//
// All V views of a call are processed by every launch (blockIdx.y = view; the cameras live in device memory).
// There is NO per-(tile, Gaussian) atomic and NO per-tile sort:
//   project     one lane per (view, Gaussian): EWA projection, extent, tile rect, SH->RGB, 32-bit depth key
//               (0xffffffff = culled); per-view totals (visible count, tile pairs) via one atomic per workgroup
//   depth sort  ONE stable LSD radix sort per view of the G depth keys (4 passes x 8 bits, payload = Gaussian id,
//               initial order = id): afterwards the Gaussians of a view are in (depth, id) order -- exactly the
//               order the published algorithm gives every tile list -- so no list ever needs sorting again.
//               Stable placement inside a 256-key slice: every thread sets its bit in an LDS bit matrix
//               [digit][256 bits]; its rank among equal digits = popcount of the bits below it
//   coarse bins stable counting sort of the depth-ordered Gaussians into bins of cb x cb tiles (cb = 4: 64 x 64 px):
//               per-workgroup LDS histograms -> [bin][chunk] table -> wave prefix scans -> placement with the same
//               bit-matrix ranking (a Gaussian spanning several bins sets one bit per bin).  An entry is 8 bytes:
//               Gaussian id + its tile rect clipped to the bin (4 x 5 bits)
//   composite   (K2) one 16x16 workgroup per tile walks ITS bin's depth-ordered entries in slices of 256: each lane
//               tests one entry's rect against the tile, a wave ballot + prefix popcount compacts the survivors into
//               an LDS staging area (order preserved), and the pixels blend them front to back; a tile stops reading
//               as soon as all its pixels are saturated.  The filter costs one lane-test per entry against ~10^3
//               lane-instructions per blended pair, and the per-tile pair lists (D x 12 B written, sorted and read
//               back in the published design) are never materialised.  n_touched: per-wave ballot -> one LDS add per
//               (entry, wave) -> one global add per staged entry
//   tile lists  (K3, and on request) the same ballot / prefix-popcount filter writes the per-tile id lists
//               (tile_start + ids, front to back): gsplat semantics re-use one list for every 32-channel chunk
// Arithmetic matches oracle/raster_ref.c operation for operation (same expression order, contraction off, explicit fused
// multiply-adds in the blend loop on both sides, and a shared polynomial exp) so that integer outputs are bit-exact and the
// maps agree to fp32 rounding.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TILE = 16;
typedef siu3r_raster_cam Cam;

// exp(x) for x <= 0 from correctly rounded fp32 operations only (fmaf == v_fma_f32; the same sequence in oracle/raster_ref.c, so
// alpha, the transmittance chain and n_touched are bit-identical on both sides): 2^(x*log2e), argument reduced to [-0.5, 0.5],
// degree-7 Taylor of 2^f as a Horner chain of 7 fused multiply-adds (|err| < 1e-7 rel), exact scaling by 2^n.
__device__ __forceinline__ float exp_det(float x) {
  if (x < -87.0f) return 0.0f;
  const float y = x * 1.4426950408889634f;
  const float n = floorf(y + 0.5f);
  const float f = y - n;
  float p = 1.52527338e-5f;
  p = __builtin_fmaf(p, f, 1.54035304e-4f);
  p = __builtin_fmaf(p, f, 1.33335581e-3f);
  p = __builtin_fmaf(p, f, 9.61812911e-3f);
  p = __builtin_fmaf(p, f, 5.55041087e-2f);
  p = __builtin_fmaf(p, f, 2.40226507e-1f);
  p = __builtin_fmaf(p, f, 6.93147181e-1f);
  p = __builtin_fmaf(p, f, 1.0f);
  return ldexpf(p, (int)n);
}
// the same value without a branch (the early return becomes a select; x > 0 or NaN: whatever exp_det returns, i.e. the same chain)
__device__ __forceinline__ float exp_det_sel(float x) {
  const float y = x * 1.4426950408889634f;
  const float n = floorf(y + 0.5f);
  const float f = y - n;
  float p = 1.52527338e-5f;
  p = __builtin_fmaf(p, f, 1.54035304e-4f);
  p = __builtin_fmaf(p, f, 1.33335581e-3f);
  p = __builtin_fmaf(p, f, 9.61812911e-3f);
  p = __builtin_fmaf(p, f, 5.55041087e-2f);
  p = __builtin_fmaf(p, f, 2.40226507e-1f);
  p = __builtin_fmaf(p, f, 6.93147181e-1f);
  p = __builtin_fmaf(p, f, 1.0f);
  const float r = ldexpf(p, (int)n);
  return x < -87.0f ? 0.0f : r;
}
// Mahalanobis half-form q = 0.5 (a dx^2 + c dy^2) + b dx dy of a pixel offset, in the shared fused order
__device__ __forceinline__ float conic_sigma(float ca, float cb, float cc, float dx, float dy) {
  const float q = __builtin_fmaf(cc * dy, dy, (ca * dx) * dx);
  return __builtin_fmaf(cb * dx, dy, 0.5f * q);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// a c - b^2 of a conic without the cancellation of the naive form (Kahan's 2 x 2 determinant: the rounding error of b * b is recovered with
// one fma; accurate to a few ulps of the RESULT).  The footprint tests that cut lists per quadrant divide by it: for a long thin splat
// seen diagonally a c and b^2 agree to 1e-6 and the naive difference is off by tens of per cent -- a box computed too small would drop
// entries that blend.
__device__ __forceinline__ float conic_det(float a, float b, float c) {
  const float w = b * b;
  const float e = __builtin_fmaf(-b, b, w);
  return __builtin_fmaf(a, c, -w) + e;
}

// ---- frame geometry shared by host and device ------------------------------------------------------------------
constexpr int NB_MAX = 1024;            // coarse bins per view (LDS: 40 B per bin in bin_scatter_kernel)
constexpr int RS_ITEMS = 16, RS_CH = 256 * RS_ITEMS;  // radix sort: keys per workgroup
constexpr int BN_ITEMS = 8, BN_CH = 256 * BN_ITEMS;   // coarse binning: sorted Gaussians per workgroup
struct Geo {
  int gw, gh, T, cb, nbx, nby, NB;
};
__host__ __device__ inline Geo make_geo(int width, int height) {
  Geo g;
  g.gw = (width + TILE - 1) / TILE;
  g.gh = (height + TILE - 1) / TILE;
  g.T = g.gw * g.gh;
  g.cb = 4;
  for (;;) {
    g.nbx = (g.gw + g.cb - 1) / g.cb;
    g.nby = (g.gh + g.cb - 1) / g.cb;
    g.NB = g.nbx * g.nby;
    if (g.NB <= NB_MAX || g.cb >= 16) break;
    g.cb *= 2;
  }
  return g;
}

__constant__ float c_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__constant__ float c_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
__constant__ float c_SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

// stats[v] = {visible Gaussians, tile pairs D, coarse entries E, overflow flags}
enum { ST_GV = 0, ST_D = 1, ST_E = 2, ST_FLAGS = 3, ST_N = 4 };

// rec: one 48-byte record per (view, Gaussian) = {mx, my, depth, 0 | conic a, b, c, opacity | r, g, b, 0}: what the composite gathers
// for a surviving list entry sits in one or two cache lines instead of four arrays.
// cov_stride 6: upper-triangular covariances (xx,xy,xz,yy,yz,zz); 9: full row-major 3x3 (Gaussians.covariances as stored).
// sh_planar 0: colors [G, ncoef, 3] (the layout the reference hands to the CUDA rasterizer, cuda_splatting.py:65); 1: [G, 3, ncoef]
// (Gaussians.harmonics as stored: no repacking copy in front of the renderer).
constexpr int PV = 16;  // views per projection chunk (blockIdx.y)
__global__ __launch_bounds__(256) void project_kernel(const Cam* __restrict__ cams, int nviews, int nf_chunk, int64_t G, const float* __restrict__ means,
                                                      const float* __restrict__ cov, int cov_stride, const float* __restrict__ opac,
                                                      const float* __restrict__ colors, int channels, int sh_planar, float* __restrict__ rec,
                                                      int32_t* __restrict__ radii, int32_t* __restrict__ rect, int32_t* __restrict__ tiles_touched,
                                                      uint32_t* __restrict__ keys, unsigned long long* __restrict__ stats) {
  // One thread per Gaussian, looping over the (up to PV) views of blockIdx.y's chunk: mean, covariance, opacity and the 300-byte SH
  // block are read ONCE per chunk instead of once per view (six target views of a pair scene re-read 1.26 GB of SH coefficients
  // before: the kernel was HBM-bound on traffic that the algorithm does not need).  The SH block is fetched at the first view that
  // sees the Gaussian; per-view totals go through LDS (one atomic pair per view and workgroup, as before).
  __shared__ int s_cnt[PV][4], s_pairs[PV][4];
  const int v_begin = blockIdx.y * PV, v_end = min(nviews, v_begin + PV);
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
  const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
  float sh[76];
  bool sh_loaded = false;
  float m0 = 0.f, m1 = 0.f, m2 = 0.f, opacity = 0.f, sxx = 0.f, sxy = 0.f, sxz = 0.f, syy = 0.f, syz = 0.f, szz = 0.f;
  if (g < G) {
    m0 = means[3 * g];
    m1 = means[3 * g + 1];
    m2 = means[3 * g + 2];
    opacity = opac[g];
    const float* cg = cov + (size_t)g * cov_stride;
    const bool tri = cov_stride == 6;
    sxx = cg[0]; sxy = cg[1]; sxz = cg[2]; syy = cg[tri ? 3 : 4]; syz = cg[tri ? 4 : 5]; szz = cg[tri ? 5 : 8];
  }
  for (int v = v_begin; v < v_end; ++v) {
  const Cam& c = cams[v];
  int ntiles = 0;
  bool valid = false;
  if (g < G) {
    const int64_t o = (int64_t)v * G + g;
    const float* V = c.w2c;
    int rx_i = 0, ry_i = 0, tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
    float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
    const float tx = V[0] * m0 + V[1] * m1 + V[2] * m2 + V[3];
    const float ty = V[4] * m0 + V[5] * m1 + V[6] * m2 + V[7];
    const float tz = V[8] * m0 + V[9] * m1 + V[10] * m2 + V[11];
    const int gw = (c.width + TILE - 1) / TILE, gh = (c.height + TILE - 1) / TILE;
    do {
      float fx, fy;
      if (c.mode == 0) {
        if (tz <= c.k2_znear_cull) break;
        fx = c.width / (2.0f * c.tanfovx);
        fy = c.height / (2.0f * c.tanfovy);
      } else {
        if (tz < c.near_plane || tz > c.far_plane) break;
        fx = c.fx;
        fy = c.fy;
      }
      float limx_pos, limx_neg, limy_pos, limy_neg;
      if (c.mode == 0) {
        limx_pos = limx_neg = 1.3f * c.tanfovx;
        limy_pos = limy_neg = 1.3f * c.tanfovy;
      } else {
        const float tfx = 0.5f * c.width / fx, tfy = 0.5f * c.height / fy;
        limx_pos = (c.width - c.cx) / fx + 0.3f * tfx;
        limx_neg = c.cx / fx + 0.3f * tfx;
        limy_pos = (c.height - c.cy) / fy + 0.3f * tfy;
        limy_neg = c.cy / fy + 0.3f * tfy;
      }
      const float rz = 1.0f / tz;
      const float txz = tx * rz, tyz = ty * rz;
      const float cxz = fminf(limx_pos, fmaxf(-limx_neg, txz)), cyz = fminf(limy_pos, fmaxf(-limy_neg, tyz));
      const float ctx = cxz * tz, cty = cyz * tz;
      const float j00 = fx * rz, j02 = -(fx * ctx) * rz * rz, j11 = fy * rz, j12 = -(fy * cty) * rz * rz;
      const float m00 = j00 * V[0] + j02 * V[8], m01 = j00 * V[1] + j02 * V[9], m02 = j00 * V[2] + j02 * V[10];
      const float m10 = j11 * V[4] + j12 * V[8], m11 = j11 * V[5] + j12 * V[9], m12 = j11 * V[6] + j12 * V[10];
      const float a0 = m00 * sxx + m01 * sxy + m02 * sxz, a1 = m00 * sxy + m01 * syy + m02 * syz, a2 = m00 * sxz + m01 * syz + m02 * szz;
      const float b0 = m10 * sxx + m11 * sxy + m12 * sxz, b1 = m10 * sxy + m11 * syy + m12 * syz, b2 = m10 * sxz + m11 * syz + m12 * szz;
      float c00 = a0 * m00 + a1 * m01 + a2 * m02;
      const float c01 = a0 * m10 + a1 * m11 + a2 * m12;
      float c11 = b0 * m10 + b1 * m11 + b2 * m12;
      const float blur = c.mode == 0 ? c.dilation : c.eps2d;
      c00 += blur;
      c11 += blur;
      const float det = c00 * c11 - c01 * c01;
      if (c.mode == 0 ? (det == 0.0f) : (det <= 0.0f)) break;
      const float det_inv = 1.0f / det;
      ca = c11 * det_inv;
      cb = -c01 * det_inv;
      cc = c00 * det_inv;
      if (c.mode == 0) {
        const float* P = c.proj;
        const float hx = P[0] * m0 + P[1] * m1 + P[2] * m2 + P[3];
        const float hy = P[4] * m0 + P[5] * m1 + P[6] * m2 + P[7];
        const float hw = P[12] * m0 + P[13] * m1 + P[14] * m2 + P[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        mx = ((hx * pw + 1.0f) * c.width - 1.0f) * 0.5f;
        my = ((hy * pw + 1.0f) * c.height - 1.0f) * 0.5f;
        const float mid = 0.5f * (c00 + c11);
        const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        const int rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam, lam2)));
        rx_i = ry_i = rad;
        tx0 = clampi((int)((mx - rad) / TILE), 0, gw);
        ty0 = clampi((int)((my - rad) / TILE), 0, gh);
        tx1 = clampi((int)((mx + rad + TILE - 1) / TILE), 0, gw);
        ty1 = clampi((int)((my + rad + TILE - 1) / TILE), 0, gh);
      } else {
        mx = fx * txz + c.cx;
        my = fy * tyz + c.cy;
        float extend = c.extent_sigma;
        if (c.opacity_aware_extent) {
          if (opacity < c.alpha_min) break;
          extend = fminf(extend, sqrtf(2.0f * logf(opacity / c.alpha_min)));
        }
        const float rx = ceilf(extend * sqrtf(c00)), ry = ceilf(extend * sqrtf(c11));
        if (rx <= c.radius_clip && ry <= c.radius_clip) break;
        if (mx + rx <= 0 || mx - rx >= c.width || my + ry <= 0 || my - ry >= c.height) break;
        rx_i = (int)rx;
        ry_i = (int)ry;
        tx0 = clampi((int)floorf((mx - rx) / TILE), 0, gw);
        ty0 = clampi((int)floorf((my - ry) / TILE), 0, gh);
        tx1 = clampi((int)ceilf((mx + rx) / TILE), 0, gw);
        ty1 = clampi((int)ceilf((my + ry) / TILE), 0, gh);
      }
      if ((tx1 - tx0) * (ty1 - ty0) == 0) {
        if (c.mode == 0) rx_i = ry_i = 0;
        break;
      }
      valid = true;
    } while (false);
    ntiles = valid ? (tx1 - tx0) * (ty1 - ty0) : 0;
    radii[2 * o] = rx_i;
    radii[2 * o + 1] = ry_i;
    tiles_touched[o] = ntiles;
    *(int4*)(rect + 4 * o) = valid ? make_int4(tx0, ty0, tx1, ty1) : make_int4(0, 0, 0, 0);
    float4* rp = (float4*)(rec + 12 * o);
    rp[0] = make_float4(mx, my, tz, 0.f);
    rp[1] = make_float4(ca, cb, cc, opacity);
    // positive floats order like their bit patterns; culled Gaussians sort behind every visible one
    // (0xffffffff is reserved for "culled": the sort drops exactly those keys and trusts stats[ST_GV] = the number of valid flags, so
    // the two must agree by construction -- a valid Gaussian whose depth has that bit pattern, a NaN payload, is clamped below it)
    keys[o] = valid ? min(__float_as_uint(tz), 0xfffffffeu) : 0xffffffffu;
    if (valid && c.mode == 1 && colors && channels == 3) {
      // gsplat family with three precomputed colour channels (the viewer's view-dependent RGB): the colour travels in the record, so
      // that the fused sort-free composite (no per-tile lists in HBM) serves this path too
      rp[2] = make_float4(colors[3 * g], colors[3 * g + 1], colors[3 * g + 2], 0.f);
    }
    if (valid && c.mode == 0 && c.sh_degree < 0) {
      // precomputed colours (the package's `colors_precomp`, cuda_splatting.py:112 use_sh = False): [G, 3], blended as given -- no SH
      // evaluation, no +0.5, no clamp at zero (feature-valued colours keep their sign)
      rp[2] = make_float4(colors[3 * g], colors[3 * g + 1], colors[3 * g + 2], 0.f);
    }
    if (valid && c.mode == 0 && c.sh_degree >= 0) {
      const float dx = m0 - c.campos[0], dy = m1 - c.campos[1], dz = m2 - c.campos[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      const float x = dx * inv, y = dy * inv, z = dz * inv;
      const int deg = c.sh_degree;
      // the Gaussian's coefficient block ([coef][rgb], 12 B per coefficient, 4-byte aligned) as 16-byte loads: a lane's
      // block is contiguous, so 12 (19 with band 4) wide loads replace 48 (75) scalar ones that each touched 64 cache lines
      struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
      const float* shp = colors + (size_t)g * channels * 3;
      float col[3];
      // (the coefficients every view of the chunk may need: the views of a call share sh_degree / sh_band4 in practice; the largest
      // request is loaded)
      const int ncf = (deg > 3 && c.sh_band4) ? 25 : (deg > 2 ? 16 : (deg > 1 ? 9 : (deg > 0 ? 4 : 1)));
      const int nf = max(ncf * 3, nf_chunk);
      if (sh_loaded) {
      } else if (!sh_planar) {
#pragma unroll
        for (int q = 0; q < 19; ++q) {
          if (4 * q < nf) {
            if (4 * q + 4 <= channels * 3) {
              const f4u t4 = *(const f4u*)(shp + 4 * q);
              sh[4 * q] = t4.v[0]; sh[4 * q + 1] = t4.v[1]; sh[4 * q + 2] = t4.v[2]; sh[4 * q + 3] = t4.v[3];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) sh[4 * q + e] = (4 * q + e < channels * 3) ? shp[4 * q + e] : 0.f;
            }
          }
        }
      } else {
        // planar block [rgb][25 coefficients] (channels == 25, checked by the launcher): the same 16-byte loads over the lane's
        // contiguous 300 bytes, written to the interleaved register order the polynomial below indexes (all indices compile-time)
#pragma unroll
        for (int q = 0; q < 19; ++q) {
          float t4[4];
          if (4 * q + 4 <= 75) {
            const f4u ld = *(const f4u*)(shp + 4 * q);
            t4[0] = ld.v[0]; t4[1] = ld.v[1]; t4[2] = ld.v[2]; t4[3] = ld.v[3];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = (4 * q + e < 75) ? shp[4 * q + e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            constexpr int NCO = 25;
            const int idx = 4 * q + e;  // = ch * 25 + coef
            if (idx < 75) sh[(idx % NCO) * 3 + idx / NCO] = t4[e];
          }
        }
      }
      sh_loaded = true;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
#define S(i) sh[(i) * 3 + ch]
        float r = SH_C0 * S(0);
        if (deg > 0) {
          r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
          if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + c_SH_C2[0] * xy * S(4) + c_SH_C2[1] * yz * S(5) + c_SH_C2[2] * (2.0f * zz - xx - yy) * S(6) + c_SH_C2[3] * xz * S(7) + c_SH_C2[4] * (xx - yy) * S(8);
            if (deg > 2) {
              r = r + c_SH_C3[0] * y * (3.0f * xx - yy) * S(9) + c_SH_C3[1] * xy * z * S(10) + c_SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                  c_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) + c_SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) +
                  c_SH_C3[5] * z * (xx - yy) * S(14) + c_SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
              if (deg > 3 && c.sh_band4) {
                r = r + c_SH_C4[0] * xy * (xx - yy) * S(16) + c_SH_C4[1] * yz * (3.0f * xx - yy) * S(17) + c_SH_C4[2] * xy * (7.0f * zz - 1.0f) * S(18) +
                    c_SH_C4[3] * yz * (7.0f * zz - 3.0f) * S(19) + c_SH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f) * S(20) +
                    c_SH_C4[5] * xz * (7.0f * zz - 3.0f) * S(21) + c_SH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f) * S(22) +
                    c_SH_C4[7] * xz * (xx - 3.0f * yy) * S(23) + c_SH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)) * S(24);
              }
            }
          }
        }
#undef S
        r += 0.5f;
        col[ch] = r < 0.0f ? 0.0f : r;
      }
      rp[2] = make_float4(col[0], col[1], col[2], 0.f);
    }
  }
  // per-view totals: wave reduction into LDS; one atomic pair per view and workgroup after the loop
  int cnt = valid ? 1 : 0, pairs = ntiles;
#pragma unroll
  for (int o_ = 32; o_ > 0; o_ >>= 1) {
    cnt += __shfl_xor(cnt, o_);
    pairs += __shfl_xor(pairs, o_);
  }
  if (lane_ == 0) {
    s_cnt[v - v_begin][wave_] = cnt;
    s_pairs[v - v_begin][wave_] = pairs;
  }
  }  // views of the chunk
  __syncthreads();
  if ((int)threadIdx.x < v_end - v_begin) {
    const int i = threadIdx.x;
    const int c4 = s_cnt[i][0] + s_cnt[i][1] + s_cnt[i][2] + s_cnt[i][3];
    const long long p4 = (long long)s_pairs[i][0] + s_pairs[i][1] + s_pairs[i][2] + s_pairs[i][3];
    if (c4) atomicAdd(&stats[(v_begin + i) * ST_N + ST_GV], (unsigned long long)c4);
    if (p4) atomicAdd(&stats[(v_begin + i) * ST_N + ST_D], (unsigned long long)p4);
  }
}

// ---- stable LSD radix sort of the depth keys (per view) --------------------------------------------------------
// hist[(v * 256 + digit) * nchunks + chunk]: digit-major, so that the scan below reads rows
// Culled Gaussians (key 0xffffffff) leave the sort in pass 0: they are neither counted nor scattered, so that passes 1-3 (and the
// binning) only see the n = stats[v][ST_GV] visible ones, compact at the front -- 46 % fewer keys on the 2 M-Gaussian stress frame.
// nvalid == nullptr: pass 0 (all G keys, culled ones skipped); else the first nvalid[v * ST_N + ST_GV] keys.
__global__ __launch_bounds__(256) void rs_hist_kernel(const uint32_t* __restrict__ keys, int32_t* __restrict__ hist, int64_t G, int shift, int nchunks,
                                                      const unsigned long long* __restrict__ nvalid) {
  __shared__ int h[256];
  const int v = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int64_t n = nvalid ? (int64_t)nvalid[v * ST_N + ST_GV] : G;
  if ((int64_t)chunk * RS_CH >= n) {  // (uniform) nothing of this chunk takes part: its histogram column is zero
    hist[((int64_t)v * 256 + t) * nchunks + chunk] = 0;
    return;
  }
  h[t] = 0;
  __syncthreads();
  const uint32_t* kp = keys + (int64_t)v * G;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = (int64_t)chunk * RS_CH + i * 256 + t;
    if (idx < n) {
      const uint32_t k = kp[idx];
      if (k != 0xffffffffu) atomicAdd(&h[(k >> shift) & 255u], 1);
    }
  }
  __syncthreads();
  hist[((int64_t)v * 256 + t) * nchunks + chunk] = h[t];
}

// in-place exclusive scan of every row of a [V * nrows, ncols] table; one wave per row; row totals -> tot
__global__ __launch_bounds__(256) void row_scan_kernel(int32_t* __restrict__ tab, int32_t* __restrict__ tot, int nrows, int ncols) {
  const int v = blockIdx.y, lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  int32_t* p = tab + ((int64_t)v * nrows + row) * ncols;
  int carry = 0;
  for (int base = 0; base < ncols; base += 64) {
    const int i = base + lane;
    const int x = i < ncols ? p[i] : 0;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(incl, o);
      if (lane >= o) incl += n;
    }
    if (i < ncols) p[i] = carry + incl - x;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) tot[(int64_t)v * nrows + row] = carry;
}

// popcount of the bits of a 256-bit LDS row strictly below position t, and of the whole row
__device__ __forceinline__ void row_rank(const uint32_t* row, int t, int& rank, int& total) {
  const uint4 a = *(const uint4*)row, b = *(const uint4*)(row + 4);
  const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  const int wi = t >> 5;
  const uint32_t low = (1u << (t & 31)) - 1u;
  rank = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pc = __popc(w[i]);
    total += pc;
    rank += i < wi ? pc : (i == wi ? __popc(w[i] & low) : 0);
  }
}

// block-wide exclusive scan of one value per thread (256 threads); returns the exclusive prefix, total in *tot
__device__ __forceinline__ int block_excl_scan(int x, int* s_w, int* tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(incl, o);
    if (lane >= o) incl += n;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += s_w[w];
  if (tot) *tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return off + incl - x;
}

// Stable scatter of one chunk (RS_CH keys, item order = (i, thread)) with a handful of workgroup barriers instead of three per item:
//   A  every wave ranks its 64 keys of item slice i among themselves -- the lanes that hold the same digit are found with 8 ballots
//      (one per digit bit), rank = popcount of that set below the lane -- and the set's first lane notes its size in cnt[i][wave][digit];
//   B  thread d walks the 64 (i, wave) counts of digit d in item order: exclusive offsets of every (slice, wave) group inside the digit's
//      run of this chunk;
//   C  local position = digits below in this chunk + group offset + rank: the chunk is sorted inside LDS;
//   D  written out in local order: global position = start of the digit in the output (digits below + this digit's keys in earlier
//      chunks) + index inside the chunk's run of the digit.
__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ ids_in,
                                                         uint32_t* __restrict__ keys_out, int32_t* __restrict__ ids_out,
                                                         const int32_t* __restrict__ hist, const int32_t* __restrict__ tot, int64_t G, int shift,
                                                         int nchunks, const unsigned long long* __restrict__ nvalid) {
  __shared__ __attribute__((aligned(16))) unsigned char cnt[RS_ITEMS * 4 * 256];   // [i][wave][digit], <= 64 each
  __shared__ unsigned short off[RS_ITEMS * 4 * 256];                              // [i][wave][digit], < RS_CH
  __shared__ int base[256], lbeg[256];
  __shared__ int32_t sid[RS_CH];
  __shared__ int s_w[4], s_last;
  const int v = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t n = nvalid ? (int64_t)nvalid[v * ST_N + ST_GV] : G;
  if ((int64_t)chunk * RS_CH >= n) return;  // (uniform)
  // start of digit t in the output = digits below + this digit's keys in earlier chunks
  const int ex = block_excl_scan(tot[v * 256 + t], s_w, nullptr);
  base[t] = ex + hist[((int64_t)v * 256 + t) * nchunks + chunk];
  {
    uint4* z = (uint4*)cnt;
#pragma unroll
    for (int i = 0; i < RS_ITEMS * 4 * 256 / 16 / 256; ++i) z[i * 256 + t] = make_uint4(0, 0, 0, 0);
  }
  const uint32_t* kp = keys_in + (int64_t)v * G;
  const int32_t* ip = ids_in ? ids_in + (int64_t)v * G : nullptr;
  uint32_t key[RS_ITEMS];
  int32_t id[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t idx = (int64_t)chunk * RS_CH + i * 256 + t;
    key[i] = idx < n ? kp[idx] : 0xffffffffu;
    id[i] = idx < n ? (ip ? ip[idx] : (int32_t)idx) : 0;
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned char rank[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const bool ok = key[i] != 0xffffffffu;  // (beyond n, or culled: pass 0 drops those)
    const uint32_t d = (key[i] >> shift) & 255u;
    unsigned long long m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int r = __popcll(m & below);
    rank[i] = (unsigned char)r;
    if (ok && r == 0) cnt[(i * 4 + wave) * 256 + d] = (unsigned char)__popcll(m);
  }
  __syncthreads();
  int run_total = 0;
  {
    int run = 0;
#pragma unroll 8
    for (int g = 0; g < RS_ITEMS * 4; ++g) {
      off[g * 256 + t] = (unsigned short)run;
      run += cnt[g * 256 + t];
    }
    run_total = run;  // this chunk's count of digit t
  }
  if (t == 255) s_last = run_total;
  // D: the chunk is first sorted INSIDE LDS (local position = digits below in this chunk + group offset + rank), then written out in
  // local order: consecutive threads then write consecutive addresses of a digit's run (a wave's store touches a handful of lines
  // instead of up to 64: the direct scatter was bound by line requests, not by bytes)
  const int lstart = block_excl_scan(run_total, s_w, nullptr);  // (its barriers also order B's reads of cnt before the reuse below)
  lbeg[t] = lstart;
  __syncthreads();
  uint32_t* skey = (uint32_t*)cnt;  // 16 KiB: the counts are no longer needed
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    if (key[i] != 0xffffffffu) {
      const uint32_t d = (key[i] >> shift) & 255u;
      const int lp = lbeg[d] + off[(i * 4 + wave) * 256 + d] + rank[i];
      skey[lp] = key[i];
      sid[lp] = id[i];
    }
  }
  __syncthreads();
  const int nloc = lbeg[255] + s_last;  // keys of this chunk that take part
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int j = i * 256 + t;
    if (j < nloc) {
      const uint32_t k = skey[j];
      const uint32_t d = (k >> shift) & 255u;
      const int64_t pos = (int64_t)v * G + base[d] + (j - lbeg[d]);
      keys_out[pos] = k;
      ids_out[pos] = sid[j];
    }
  }
}

// ---- coarse bins (cb x cb tiles), filled in depth order ---------------------------------------------------------
__device__ __forceinline__ void coarse_range(int4 r, int cb, int& cx0, int& cy0, int& cx1, int& cy1) {
  cx0 = r.x / cb;
  cy0 = r.y / cb;
  cx1 = (r.z - 1) / cb;
  cy1 = (r.w - 1) / cb;
}

__global__ __launch_bounds__(256) void bin_count_kernel(Geo geo, const uint32_t* __restrict__ keys, const int32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ rect, int32_t* __restrict__ bin_hist, int64_t G, int nchunks,
                                                        const unsigned long long* __restrict__ stats) {
  __shared__ int h[NB_MAX];
  const int v = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  const int64_t n = (int64_t)stats[v * ST_N + ST_GV];  // the sorted arrays hold the visible Gaussians, compact, in [0, n)
  if ((int64_t)chunk * BN_CH >= n) {  // (uniform)
    for (int b = t; b < geo.NB; b += 256) bin_hist[((int64_t)v * geo.NB + b) * nchunks + chunk] = 0;
    return;
  }
  for (int b = t; b < geo.NB; b += 256) h[b] = 0;
  __syncthreads();
  const uint32_t* kp = keys + (int64_t)v * G;
  const int32_t* ip = ids + (int64_t)v * G;
#pragma unroll
  for (int i = 0; i < BN_ITEMS; ++i) {
    const int64_t idx = (int64_t)chunk * BN_CH + i * 256 + t;
    if (idx < n) {
      const int4 r = *(const int4*)(rect + 4 * ((int64_t)v * G + ip[idx]));
      int cx0, cy0, cx1, cy1;
      coarse_range(r, geo.cb, cx0, cy0, cx1, cy1);
      for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) atomicAdd(&h[cy * geo.nbx + cx], 1);
    }
  }
  __syncthreads();
  for (int b = t; b < geo.NB; b += 256) bin_hist[((int64_t)v * geo.NB + b) * nchunks + chunk] = h[b];
}

// entry = (Gaussian id, rect clipped to the bin: x0 | y0 << 5 | x1 << 10 | y1 << 15, each in [0, cb])
__global__ __launch_bounds__(256) void bin_scatter_kernel(Geo geo, const uint32_t* __restrict__ keys, const int32_t* __restrict__ ids,
                                                          const int32_t* __restrict__ rect, const int32_t* __restrict__ bin_hist,
                                                          const int32_t* __restrict__ bin_tot, int32_t* __restrict__ bin_start,
                                                          uint2* __restrict__ entries, int64_t cap_e, unsigned long long* __restrict__ stats, int64_t G,
                                                          int nchunks) {
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
  const int NB = geo.NB;
  uint32_t* M = dyn;                 // [NB][8]: bit t of row b <=> thread t's Gaussian of this slice touches bin b
  int* base = (int*)(dyn + NB * 8);  // [NB] next free slot of the bin for this workgroup
  int* add = base + NB;              // [NB] slots consumed by the slice being placed
  __shared__ int s_w[4];
  const int v = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
  // bin starts = exclusive scan of the bin totals: thread t owns bins 4 t .. 4 t + 3 (NB <= 1024)
  int loc[4], sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = t * 4 + i;
    loc[i] = b < NB ? bin_tot[v * NB + b] : 0;
    sum += loc[i];
  }
  int total_e = 0;
  int run = block_excl_scan(sum, s_w, &total_e);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = t * 4 + i;
    if (b < NB) {
      base[b] = run + bin_hist[((int64_t)v * NB + b) * nchunks + chunk];
      add[b] = 0;
      if (chunk == 0) bin_start[v * (NB + 1) + b] = run;
    }
    run += loc[i];
  }
  if (chunk == 0 && t == 0) {
    bin_start[v * (NB + 1) + NB] = total_e;
    stats[v * ST_N + ST_E] = (unsigned long long)total_e;
    if (total_e > cap_e) stats[v * ST_N + ST_FLAGS] = 1ull;
  }
  const int64_t n = (int64_t)stats[v * ST_N + ST_GV];  // the sorted arrays hold the visible Gaussians, compact, in [0, n)
  if ((int64_t)chunk * BN_CH >= n) return;            // (uniform; chunk 0 has written bin_start and the totals above)
  for (int i = t; i < NB * 8; i += 256) M[i] = 0;
  __syncthreads();
  const int32_t* ip = ids + (int64_t)v * G;
  const int wi = t >> 5;
  const uint32_t bit = 1u << (t & 31);
  const int cb = geo.cb, nbx = geo.nbx;
  for (int i = 0; i < BN_ITEMS; ++i) {
    const int64_t idx = (int64_t)chunk * BN_CH + i * 256 + t;
    const bool ok = idx < n;
    int id = 0, cx0 = 0, cy0 = 0, cx1 = -1, cy1 = -1;
    int4 r = make_int4(0, 0, 0, 0);
    if (ok) {
      id = ip[idx];
      r = *(const int4*)(rect + 4 * ((int64_t)v * G + id));
      coarse_range(r, cb, cx0, cy0, cx1, cy1);
      for (int cy = cy0; cy <= cy1; ++cy)
        for (int cx = cx0; cx <= cx1; ++cx) atomicOr(&M[(cy * nbx + cx) * 8 + wi], bit);
    }
    __syncthreads();
    for (int cy = cy0; cy <= cy1; ++cy)
      for (int cx = cx0; cx <= cx1; ++cx) {
        const int b = cy * nbx + cx;
        int rank, total;
        row_rank(&M[b * 8], t, rank, total);
        if (rank == 0) add[b] = total;
        const int64_t pos = (int64_t)base[b] + rank;
        if (pos < cap_e) {
          const int ox = cx * cb, oy = cy * cb;
          const uint32_t x0 = (uint32_t)(max(r.x, ox) - ox), y0 = (uint32_t)(max(r.y, oy) - oy);
          const uint32_t x1 = (uint32_t)(min(r.z, ox + cb) - ox), y1 = (uint32_t)(min(r.w, oy + cb) - oy);
          entries[(int64_t)v * cap_e + pos] = make_uint2((uint32_t)id, x0 | (y0 << 5) | (x1 << 10) | (y1 << 15));
        }
      }
    __syncthreads();
    for (int cy = cy0; cy <= cy1; ++cy)
      for (int cx = cx0; cx <= cx1; ++cx) M[(cy * nbx + cx) * 8 + wi] = 0;
    for (int b = t; b < NB; b += 256) {
      const int a = add[b];
      if (a) {
        base[b] += a;
        add[b] = 0;
      }
    }
    __syncthreads();
  }
}

#ifndef SIU3R_COMP_DBG
#define SIU3R_COMP_DBG 0  // tuning builds (tools/ab_raster.sh): 1 = no blend walk (scan + stage only), 2 = no per-wave list building either
#endif
// does the packed bin-relative rect cover tile (rtx, rty) of the bin?
__device__ __forceinline__ bool entry_covers(uint32_t pr, int rtx, int rty) {
  const int x0 = pr & 31, y0 = (pr >> 5) & 31, x1 = (pr >> 10) & 31, y1 = (pr >> 15) & 31;
  return rtx >= x0 && rtx < x1 && rty >= y0 && rty < y1;
}

// ---- K2 composite: colour [3,H,W] + depth + accumulated opacity (+ n_touched), fused with the per-tile filter -----
constexpr int STG = 512;       // staging capacity (survivors awaiting the blend)
constexpr int STG_PULL = 192;  // refill while fewer than this are staged (<= STG - 256: one more slice always fits)
constexpr int FK = 4;          // slices of 256 entries tested per refill round
// K3 (compile time): the gsplat family's conventions on the same walk -- pixel centres at +0.5, saturation test nT <= t_min, channel-last
// [H,W,3] output without background or depth (siu3r_raster_composite_rgb with mode-1 cameras: the viewer's RGB render).
template <bool NT, bool K3 = false>
__global__ __launch_bounds__(256) void composite_rgb_kernel(const Cam* __restrict__ cams, Geo geo, const int32_t* __restrict__ bin_start,
                                                            const uint2* __restrict__ entries, int64_t cap_e, const float* __restrict__ rec, int64_t G,
                                                            float* __restrict__ image, float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                            int32_t* __restrict__ n_touched) {
  // staged survivors, 52 B each: {mx, my, depth, id} {conic a, b, c, opacity} {r, g, b, n_touched partial} + quadrant mask.
  // A wave owns one 8x8 quadrant of the tile; a survivor's mask says which quadrants its alpha >= alpha_min footprint can reach
  // (bounding box of the ellipse sigma <= ln(opacity / alpha_min), padded), so that the other waves skip it after one LDS read:
  // pixel-aligned splats are a few pixels wide, and three of four waves would otherwise evaluate them for nothing.
  __shared__ int s_m[STG];
  __shared__ __attribute__((aligned(16))) float s_a[STG][4];
  __shared__ __attribute__((aligned(16))) float s_co[STG][4];
  __shared__ __attribute__((aligned(16))) float s_c[STG][4];
  __shared__ int s_wcnt[4][FK];
  __shared__ __attribute__((aligned(16))) unsigned short s_list[4][STG];
  const int v = blockIdx.y;
  const Cam& c = cams[v];
  const int tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const int bx = tx / geo.cb, by = ty / geo.cb, bin = by * geo.nbx + bx;
  const int rtx = tx - bx * geo.cb, rty = ty - by * geo.cb;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);  // wave = 8x8 quadrant
  const int px = tx * TILE + lx, py = ty * TILE + ly;
  const bool inside = px < c.width && py < c.height;
  const float pxf = (float)px + (K3 ? 0.5f : 0.0f), pyf = (float)py + (K3 ? 0.5f : 0.0f);
  const int64_t ebeg = bin_start[v * (geo.NB + 1) + bin];
  const int64_t eend = min((int64_t)bin_start[v * (geo.NB + 1) + bin + 1], cap_e);
  const uint2* ep = entries + (int64_t)v * cap_e;
  const int64_t vg = (int64_t)v * G;
  const float alpha_min = c.alpha_min, alpha_max = c.alpha_max, t_min = c.t_min;
  const bool nt_post = c.nt_post_blend != 0;
  const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);
  const int wbit = 1 << wave;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, O = 0.f;
  bool done = !inside;
  int staged = 0;
  int64_t base = ebeg;
  while (true) {
    if (__syncthreads_count(done) == 256) break;  // also fences the previous round's LDS reads
    // Refill: up to FK slices of 256 entries per round -- their loads, and then the survivors' record gathers, are in flight
    // together, and one barrier pair covers all of them (the walk is otherwise two dependent memory latencies and two barriers per
    // 256 entries, of which typically a tenth survive).  Slices are committed in order while they fit the staging area; the rest
    // is simply read again in the next round.
    while (staged < STG_PULL && base < eend) {
      uint2 e[FK];
      bool pass[FK];
      unsigned long long m[FK];
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        const int64_t i = base + k * 256 + t;
        e[k] = i < eend ? ep[i] : make_uint2(0, 0);
      }
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        pass[k] = (base + k * 256 + t < eend) && entry_covers(e[k].y, rtx, rty);
        m[k] = __ballot(pass[k]);
        if (lane == 0) s_wcnt[wave][k] = __popcll(m[k]);
      }
      __syncthreads();
      int kk = 0, off_k[FK], run = staged;
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        const int tk = s_wcnt[0][k] + s_wcnt[1][k] + s_wcnt[2][k] + s_wcnt[3][k];
        int o = run;
        for (int w = 0; w < wave; ++w) o += s_wcnt[w][k];
        off_k[k] = o + __popcll(m[k] & ((1ull << lane) - 1ull));
        if (kk == k && run + tk <= STG) {  // slice k fits (slice 0 always does: staged < STG_PULL <= STG - 256)
          kk = k + 1;
          run += tk;
        }
      }
#pragma unroll
      for (int k = 0; k < FK; ++k) {
        if (k < kk && pass[k]) {
          const int off = off_k[k];
          const float4* rp = (const float4*)(rec + 12 * (vg + e[k].x));
          float4 r0 = rp[0];
          const float4 r1 = rp[1];
          float4 r2 = rp[2];
          r0.w = __int_as_float((int)e[k].x);
          r2.w = __int_as_float(0);
          *(float4*)s_a[off] = r0;
          *(float4*)s_co[off] = r1;
          *(float4*)s_c[off] = r2;
          // footprint box: alpha >= alpha_min  =>  sigma <= L = ln(opacity / alpha_min)  =>  |dx| <= sqrt(2 L cov_xx), cov = conic^-1.
          // Padded by 1 % + 0.05 px (the exact per-pixel tests below still decide; the box only has to be conservative)
          int mk = 15;  // NaN / degenerate conics: no culling, the exact tests decide
          const float L = __logf(r1.w / alpha_min);
          const float det = conic_det(r1.x, r1.y, r1.z);
          if (L <= 0.f) {
            mk = 0;  // opacity below alpha_min: alpha = min(alpha_max, opacity * exp(<= 0)) can never reach it
          } else if (det > 0.f) {
            // (K3: pixel centres sit half a pixel further: the box grows by that much)
            const float ex = sqrtf(2.f * L * r1.z / det) * 1.01f + (K3 ? 0.55f : 0.05f), ey = sqrtf(2.f * L * r1.x / det) * 1.01f + (K3 ? 0.55f : 0.05f);
            const float x0 = r0.x - ex - tile_x0, x1 = r0.x + ex - tile_x0, y0 = r0.y - ey - tile_y0, y1 = r0.y + ey - tile_y0;
            const int cx = (x0 <= 7.f && x1 >= 0.f ? 1 : 0) | (x0 <= 15.f && x1 >= 8.f ? 2 : 0);
            const int cy = (y0 <= 7.f && y1 >= 0.f ? 1 : 0) | (y0 <= 15.f && y1 >= 8.f ? 2 : 0);
            mk = ((cy & 1) ? cx : 0) | ((cy & 2) ? (cx << 2) : 0);
          }
          s_m[off] = mk;
        }
      }
      staged = run;
      base += 256 * kk;
      __syncthreads();
    }
    if (staged == 0) break;
    // this wave's own list (order kept) of the staged survivors that can reach its quadrant: ballot + prefix popcount again, wave-local
    // (LDS operations of one wave complete in order: no barrier)
    int nq = 0;
    for (int b = 0; b < ((SIU3R_COMP_DBG & 2) ? 0 : staged); b += 64) {
      const int i = b + lane;
      const bool hit = i < staged && (s_m[i] & wbit);
      const unsigned long long mh = __ballot(hit);
      if (hit) s_list[wave][nq + __popcll(mh & ((1ull << lane) - 1ull))] = (unsigned short)i;
      nq += __popcll(mh);
    }
    // The walk.  Rounds 2-5 evaluated one entry per iteration behind four nested early-outs (power > 0, x < -87 inside the exponential,
    // alpha < alpha_min, saturation): ~55 vector + ~35 scalar / branch instructions per (wave, entry), nine of the vector ones register
    // moves that rotated the software prefetch, and the early-outs almost never fire wave-wide (a listed entry nearly always has SOME
    // pixel of the quadrant inside its footprint).  Now: alpha of every lane evaluated branch-free (same operations in the same order
    // per lane: identical bits), ONE predicated block for the lanes that blend, four entries per iteration in two alternating register
    // sets (records of the next pair and indices of the pair after it in flight while a pair is evaluated; nothing to rotate).
    // Entries at or behind nq are dead (index 0, never blended).
    if (SIU3R_COMP_DBG & 1) nq = 0;
    const unsigned short* lst = s_list[wave];
    auto blend = [&](const int j, const float4 A, const float4 Q, const bool live) {
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float power = -conic_sigma(Q.x, Q.y, Q.z, dx, dy);
      const float a = fminf(alpha_max, Q.w * exp_det_sel(power));
      const float nT = __builtin_fmaf(-T, a, T);  // T (1 - a)
      const bool reach = live && !done && !(power > 0.0f) && !(a < alpha_min);
      const bool sat = reach && (K3 ? (nT <= t_min) : (nT < t_min));
      done = done || sat;
      if (reach && !sat) {
        const float w = a * T;
        const float4 Cj = *(const float4*)s_c[j];
        C0 = __builtin_fmaf(Cj.x, w, C0);
        C1 = __builtin_fmaf(Cj.y, w, C1);
        C2 = __builtin_fmaf(Cj.z, w, C2);
        D = __builtin_fmaf(A.z, w, D);
        O += w;
        if (NT) {
          // pixels that count this Gaussian: ballot over the lanes that reached this point, one LDS add per wave
          const bool cnt = (nt_post ? nT : T) > 0.5f;
          const unsigned long long mc = __ballot(cnt);
          if (cnt && lane == (int)__ffsll((long long)mc) - 1) atomicAdd((int*)&s_c[j][3], (int)__popcll(mc));
        }
        T = nT;
      }
    };
    // (two 16-bit indices per 32-bit read: the lists start 4-byte aligned and pairs start at even positions)
    auto pair_idx = [&](int i, int& ja, int& jb) {
      const unsigned u = i < nq ? *(const unsigned*)(lst + i) : 0u;
      ja = (int)(u & 0xffffu);
      jb = i + 1 < nq ? (int)(u >> 16) : 0;
    };
    int ja0, jb0, ja1, jb1;
    pair_idx(0, ja0, jb0);
    pair_idx(2, ja1, jb1);
    float4 Aa0 = *(const float4*)s_a[ja0], Qa0 = *(const float4*)s_co[ja0], Ab0 = *(const float4*)s_a[jb0], Qb0 = *(const float4*)s_co[jb0];
    for (int ii = 0; ii < nq; ii += 4) {
      // set 1 <- records of pair ii + 2 (indices loaded one half-iteration ago); indices of pair ii + 4
      const float4 Aa1 = *(const float4*)s_a[ja1], Qa1 = *(const float4*)s_co[ja1], Ab1 = *(const float4*)s_a[jb1], Qb1 = *(const float4*)s_co[jb1];
      int jan, jbn;
      pair_idx(ii + 4, jan, jbn);
      blend(ja0, Aa0, Qa0, true);
      blend(jb0, Ab0, Qb0, ii + 1 < nq);
      if (__ballot(!done) == 0ull) break;
      // set 0 <- records of pair ii + 4; indices of pair ii + 6
      Aa0 = *(const float4*)s_a[jan], Qa0 = *(const float4*)s_co[jan], Ab0 = *(const float4*)s_a[jbn], Qb0 = *(const float4*)s_co[jbn];
      const int ja1c = ja1, jb1c = jb1;
      pair_idx(ii + 6, ja1, jb1);
      blend(ja1c, Aa1, Qa1, ii + 2 < nq);
      blend(jb1c, Ab1, Qb1, ii + 3 < nq);
      ja0 = jan, jb0 = jbn;
      if (__ballot(!done) == 0ull) break;
    }
    if (NT) {
      __syncthreads();
      for (int j = t; j < staged; j += 256) {
        const int n = __float_as_int(s_c[j][3]);
        if (n) atomicAdd(&n_touched[vg + __float_as_int(s_a[j][3])], n);
      }
    }
    staged = 0;
  }
  // A view whose coarse-bin entries did not fit cap_e lost its farthest splats: its outputs are POISONED (NaN) instead of being subtly
  // wrong, so that a caller may defer reading the overflow counters (stats) to a convenient moment (raster.py: check_overflow="deferred")
  if (inside && (int64_t)bin_start[v * (geo.NB + 1) + geo.NB] > cap_e) {
    const float qnan = __int_as_float(0x7fc00000);
    T = qnan, C0 = qnan, C1 = qnan, C2 = qnan, D = qnan, O = qnan;
  }
  if (inside && K3) {
    const size_t hw = (size_t)c.width * c.height, pix = (size_t)py * c.width + px;
    float* o = image + ((size_t)v * hw + pix) * 3;
    o[0] = C0;
    o[1] = C1;
    o[2] = C2;
    out_alpha[(size_t)v * hw + pix] = O;
  } else if (inside) {
    const size_t hw = (size_t)c.width * c.height, pix = (size_t)py * c.width + px;
    float* img = image + (size_t)v * 3 * hw;
    img[pix] = __builtin_fmaf(T, c.bg[0], C0);
    img[hw + pix] = __builtin_fmaf(T, c.bg[1], C1);
    img[2 * hw + pix] = __builtin_fmaf(T, c.bg[2], C2);
    out_depth[(size_t)v * hw + pix] = D;
    out_alpha[(size_t)v * hw + pix] = O;
  }
}

// ---- per-tile lists (front to back) out of the coarse bins: count, scan, write -------------------------------------
__global__ __launch_bounds__(256) void tl_count_kernel(Geo geo, const int32_t* __restrict__ bin_start, const uint2* __restrict__ entries,
                                                       int64_t cap_e, int32_t* __restrict__ tile_count) {
  __shared__ int s_w[4];
  const int v = blockIdx.y, tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const int bx = tx / geo.cb, by = ty / geo.cb, bin = by * geo.nbx + bx;
  const int rtx = tx - bx * geo.cb, rty = ty - by * geo.cb;
  const int64_t ebeg = bin_start[v * (geo.NB + 1) + bin];
  const int64_t eend = min((int64_t)bin_start[v * (geo.NB + 1) + bin + 1], cap_e);
  const uint2* ep = entries + (int64_t)v * cap_e;
  int n = 0;
  for (int64_t i = ebeg + threadIdx.x; i < eend; i += 256) n += entry_covers(ep[i].y, rtx, rty) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) tile_count[(int64_t)v * geo.T + tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of the tile counts of a view -> tile_start[T + 2]: [0..T] clamped to cap (so that lists never overrun the
// id buffer), [T + 1] = the true total; single workgroup of 1024 threads per view
__global__ __launch_bounds__(1024) void tl_scan_kernel(const int32_t* __restrict__ tile_count, int32_t* __restrict__ tile_start, int T, int64_t cap,
                                                       unsigned long long* __restrict__ stats) {
  __shared__ int32_t wsum[16];
  __shared__ long long carry;
  const int v = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int32_t* cnt = tile_count + (int64_t)v * T;
  int32_t* ts = tile_start + (int64_t)v * (T + 2);
  if (t == 0) carry = 0;
  __syncthreads();
  for (int b = 0; b < T; b += 1024) {
    const int i = b + t;
    const int x = i < T ? cnt[i] : 0;
    int incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(incl, o);
      if (lane >= o) incl += n;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const long long excl = carry + woff + incl - x;
    if (i < T) ts[i] = (int32_t)min(excl, (long long)cap);
    __syncthreads();
    if (t == 1023) carry = excl + x;
    __syncthreads();
  }
  if (t == 0) {
    ts[T] = (int32_t)min(carry, (long long)cap);
    ts[T + 1] = (int32_t)min(carry, (long long)0x7fffffff);
    if (carry > cap) stats[v * ST_N + ST_FLAGS] |= 2ull;
  }
}

__global__ __launch_bounds__(256) void tl_write_kernel(Geo geo, const int32_t* __restrict__ bin_start, const uint2* __restrict__ entries,
                                                       int64_t cap_e, const int32_t* __restrict__ tile_start, int32_t* __restrict__ ids, int64_t cap_d) {
  __shared__ int s_wcnt[4];
  const int v = blockIdx.y, tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const int bx = tx / geo.cb, by = ty / geo.cb, bin = by * geo.nbx + bx;
  const int rtx = tx - bx * geo.cb, rty = ty - by * geo.cb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ebeg = bin_start[v * (geo.NB + 1) + bin];
  const int64_t eend = min((int64_t)bin_start[v * (geo.NB + 1) + bin + 1], cap_e);
  const uint2* ep = entries + (int64_t)v * cap_e;
  const int32_t* ts = tile_start + (int64_t)v * (geo.T + 2);
  const int64_t obeg = ts[tile], oend = ts[tile + 1];  // clamped to cap_d by the scan
  int32_t* out = ids + (int64_t)v * cap_d;
  int64_t run = obeg;
  for (int64_t b = ebeg; b < eend; b += 256) {
    const int64_t i = b + threadIdx.x;
    bool pass = false;
    uint2 e = make_uint2(0, 0);
    if (i < eend) {
      e = ep[i];
      pass = entry_covers(e.y, rtx, rty);
    }
    const unsigned long long m = __ballot(pass);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int64_t off = run + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    if (pass && off < oend) out[off] = (int32_t)e.x;
    run += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
}

// ---- K3 composite: one 32-channel chunk of the feature matrix per blockIdx.y; colours [H,W,C]; alpha written by chunk 0
constexpr int CHUNK = 32;
__global__ __launch_bounds__(256) void composite_feat_kernel(const Cam* __restrict__ cams, Geo geo, const int32_t* __restrict__ tile_start,
                                                             const int32_t* __restrict__ ids, int64_t cap_d, const float* __restrict__ rec,
                                                             const float* __restrict__ feats, int channels, int64_t G, float* __restrict__ out,
                                                             float* __restrict__ out_alpha) {
  __shared__ float s_xy[128][2];
  __shared__ __attribute__((aligned(16))) float s_co[128][4];
  __shared__ float s_f[128][CHUNK + 1];
  const int v = blockIdx.z;
  const Cam& c = cams[v];
  const int tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const int ch0 = blockIdx.y * CHUNK, nch = min(CHUNK, channels - ch0);
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int px = tx * TILE + lx, py = ty * TILE + ly;
  const bool inside = px < c.width && py < c.height;
  const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
  const int32_t* ts = tile_start + (int64_t)v * (geo.T + 2);
  const int32_t* idp = ids + (int64_t)v * cap_d;
  const int64_t vg = (int64_t)v * G;
  const int beg = ts[tile], end = ts[tile + 1];
  const float alpha_min = c.alpha_min, alpha_max = c.alpha_max, t_min = c.t_min;
  float T = 1.0f, O = 0.f;
  float acc[CHUNK];
#pragma unroll
  for (int k = 0; k < CHUNK; ++k) acc[k] = 0.f;
  bool done = !inside;
  for (int base = beg; base < end; base += 128) {
    if (__syncthreads_count(done) == 256) break;
    const int cnt = min(128, end - base);
    if (threadIdx.x < cnt) {
      const float4* rp = (const float4*)(rec + 12 * (vg + idp[base + threadIdx.x]));
      const float4 r0 = rp[0];
      *(float2*)s_xy[threadIdx.x] = make_float2(r0.x, r0.y);
      *(float4*)s_co[threadIdx.x] = rp[1];
    }
    for (int e = threadIdx.x; e < cnt * CHUNK; e += 256) {
      const int j = e / CHUNK, k = e - j * CHUNK;
      s_f[j][k] = k < nch ? feats[(size_t)idp[base + j] * channels + ch0 + k] : 0.f;
    }
    __syncthreads();
    for (int j = 0; !done && j < cnt; ++j) {
      const float dx = s_xy[j][0] - pxf, dy = s_xy[j][1] - pyf;
      const float sigma = conic_sigma(s_co[j][0], s_co[j][1], s_co[j][2], dx, dy);
      if (sigma < 0.0f) continue;
      const float a = fminf(alpha_max, s_co[j][3] * exp_det(-sigma));
      if (a < alpha_min) continue;
      const float nT = __builtin_fmaf(-T, a, T);
      if (nT <= t_min) {
        done = true;
        continue;
      }
      const float w = a * T;
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) acc[k] = __builtin_fmaf(s_f[j][k], w, acc[k]);
      O += w;
      T = nT;
    }
  }
  if (inside) {
    const size_t hw = (size_t)c.width * c.height, pix = (size_t)py * c.width + px;
    float* o = out + ((size_t)v * hw + pix) * channels + ch0;
    for (int k = 0; k < nch; ++k) o[k] = acc[k];
    if (blockIdx.y == 0 && out_alpha) out_alpha[(size_t)v * hw + pix] = O;
  }
}

__global__ void scale_kernel(float* x, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

inline dim3 g1(int64_t n, int b = 256) { return dim3((unsigned)cdiv64(n, b)); }

// every view of a call shares the frame size and the mode
int check_views(const Cam* cams, int V, const char* who) {
  SIU3R_CHECK(cams && V >= 1 && V <= 65535, "%s: bad view array (V = %d)", who, V);
  for (int v = 0; v < V; ++v) {
    SIU3R_CHECK(cams[v].mode == cams[0].mode && cams[v].width == cams[0].width && cams[v].height == cams[0].height,
                "%s: the views of one call must share mode and frame size", who);
    SIU3R_CHECK(cams[v].mode == 0 || cams[v].mode == 1, "%s: bad mode %d", who, cams[v].mode);
    SIU3R_CHECK(cams[v].width > 0 && cams[v].height > 0, "%s: empty frame", who);
    SIU3R_CHECK(cams[v].mode == 0 ? cams[v].k2_znear_cull >= 0.f : cams[v].near_plane > 0.f,
                "%s: the near cull must be positive (depth keys are the bit patterns of positive floats)", who);
  }
  return 0;
}

}  // namespace

// ---- viewer-semantics helpers (reference viewer.py:301-336: gsplat.rasterization fed with quats / exp(scales) / SH) ----------------
__global__ void quat_scale_cov6_kernel(int64_t G, const float* quats, const float* scales, float* cov6) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float4 q = *(const float4*)(quats + 4 * g);
  float w = q.x, x = q.y, y = q.z, z = q.w;
  const float inv = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
  w *= inv; x *= inv; y *= inv; z *= inv;
  const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  const float R[9] = {1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy), 2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx),
                      2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2)};
  const float s0 = scales[3 * g], s1 = scales[3 * g + 1], s2 = scales[3 * g + 2];
  float M[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) { M[3 * r] = R[3 * r] * s0; M[3 * r + 1] = R[3 * r + 1] * s1; M[3 * r + 2] = R[3 * r + 2] * s2; }
  int o = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = r; c < 3; ++c) cov6[6 * g + o++] = M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1] + M[3 * r + 2] * M[3 * c + 2];
}

// The workgroup's 256 coefficient blocks ([coef][rgb], 12 * ncoef bytes each, contiguous) are copied to LDS with coalesced 16-byte
// loads and read back lane by lane: a lane reading its own 300-byte block straight from global memory touches 64 cache lines per load
// instruction (the kernel then ran at 1.6 TB/s; the algorithm reads every byte once).
template <int DEG>
__global__ __launch_bounds__(256) void sh_eval_kernel(int64_t G, int ncoef, const float* means, float cx, float cy, float cz, const float* sh, float* rgb,
                                                      const float* campos_dev) {
  extern __shared__ __attribute__((aligned(16))) float s_sh[];
  if (campos_dev) {  // camera position handed over in device memory (no host round trip of the pose)
    cx = campos_dev[0];
    cy = campos_dev[1];
    cz = campos_dev[2];
  }
  const int nf = ncoef * 3;
  {
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x;
    const int64_t nfl = (int64_t)min((int64_t)blockDim.x, G - g0) * nf;  // floats of this workgroup
    const float* blk = sh + g0 * nf;
    if ((((uintptr_t)blk) & 15) == 0) {
      for (int64_t i = threadIdx.x; i < nfl / 4; i += blockDim.x) ((float4*)s_sh)[i] = ((const float4*)blk)[i];
      for (int64_t i = (nfl & ~3ll) + threadIdx.x; i < nfl; i += blockDim.x) s_sh[i] = blk[i];
    } else {
      for (int64_t i = threadIdx.x; i < nfl; i += blockDim.x) s_sh[i] = blk[i];
    }
  }
  __syncthreads();
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float dx = means[3 * g] - cx, dy = means[3 * g + 1] - cy, dz = means[3 * g + 2] - cz;
  const float inorm = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
  constexpr int NB = (DEG + 1) * (DEG + 1);
  float b[25];
  b[0] = 0.2820947917738781f;
  if (DEG >= 1) { b[1] = -0.48860251190292f * y; b[2] = 0.48860251190292f * z; b[3] = -0.48860251190292f * x; }
  float z2 = 0, fC1 = 0, fS1 = 0, fC2 = 0, fS2 = 0;
  if (DEG >= 2) {
    z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    fC1 = x * x - y * y;
    fS1 = 2.0f * x * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f; b[7] = fTmp0B * x; b[5] = fTmp0B * y; b[8] = 0.5462742152960395f * fC1; b[4] = 0.5462742152960395f * fS1;
  }
  if (DEG >= 3) {
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp1B = 1.445305721320277f * z;
    fC2 = x * fC1 - y * fS1;
    fS2 = x * fS1 + y * fC1;
    b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f); b[13] = fTmp0C * x; b[11] = fTmp0C * y; b[14] = fTmp1B * fC1; b[10] = fTmp1B * fS1;
    b[15] = -0.5900435899266435f * fC2; b[9] = -0.5900435899266435f * fS2;
  }
  if (DEG >= 4) {
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f), fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6]; b[21] = fTmp0D * x; b[19] = fTmp0D * y; b[22] = fTmp1C * fC1; b[18] = fTmp1C * fS1;
    b[23] = fTmp2B * fC2; b[17] = fTmp2B * fS2; b[24] = 0.6258357354491763f * fC3; b[16] = 0.6258357354491763f * fS3;
  }
  float c[NB * 3];
#pragma unroll
  for (int i = 0; i < NB * 3; ++i) c[i] = s_sh[(size_t)threadIdx.x * nf + i];  // (nf is odd for 25 coefficients: consecutive lanes, distinct banks)
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) r = r + b[i] * c[i * 3 + ch];
    r += 0.5f;
    rgb[3 * g + ch] = r < 0.0f ? 0.0f : r;
  }
}

__global__ void blend_bg_kernel(int64_t n, int C, float* colors, const float* alpha, float b0, float b1, float b2, const float* bg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = i / C;
  const int c = (int)(i - p * C);
  const float b = bg ? bg[c] : (c == 0 ? b0 : (c == 1 ? b1 : b2));
  colors[i] = colors[i] + (1.0f - alpha[p]) * b;
}

// ---- K3 composite, matrix-core form: the blend as rank-2 updates ---------------------------------------------------------------------------
// (Round 5 also built and measured two other all-channel forms, both bit-identical and both slower: lanes = channels with per-pixel
// sparse pair lists in LDS -- bound by ~95 bookkeeping instructions per pixel and batch -- and lanes = pixels with the feature row through
// scalar loads as SGPR operands of packed FMAs -- bound by the scalar-cache round trip per 16 channels: 1.66 / 2.2 ms per 168-channel
// frame against 1.93 for the 32-channel kernel and 0.69 here)
// out[pixel][channel] += w[pixel] * f[channel] per list entry is an outer product; two consecutive entries of a quadrant's list make one
// v_mfma_f32_32x32x2_f32 per (32-pixel, 32-channel) block: A[i][k] = the blending weight of pixel i for entry k, B[k][j] = channel j of
// entry k's feature row.  The f32-input MFMA is exact f32 and accumulates like an fmaf chain in k order (the guide: bitwise), i.e. it
// performs the oracle's FMAs in the oracle's order -- a pixel an entry does not reach takes part with w = 0, which leaves its sums
// untouched.  What the matrix pipe buys here is not FLOPs (its f32 rate is the packed-VALU rate) but OPERAND DELIVERY: the feature row
// enters as the B operand, one LDS read of 32 consecutive floats per block and entry pair, instead of being broadcast to 64 lanes value by
// value (one LDS serves four SIMDs: the 32 broadcast reads per entry and chunk are what bound the 32-channel kernel), and the VALU is
// left to the alpha / transmittance steps (the SIMD's other wave runs its own while this wave's MFMAs execute).  lane = pixel of the wave's 8 x 8 quadrant for that step (once per entry, for all
// channels; lists cut to the quadrant by the extent test at load time); v_permlane32_swap turns two weight registers into the two A
// operands (pixels 0-31 / 32-63 of the quadrant, k = entry).  Accumulators: 2 x NP blocks of 16 registers.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---- no workgroup barrier: per-QUADRANT lists + wave-private staging -------------------------------------------------------------------
// (Round 5's first matrix-core form staged a tile's list in batches that its four waves shared: every batch ended in a barrier that the
// wave with the longest quadrant list decided -- counters: 54 % of the wave cycles parked, 0.69 ms per 168-channel frame against 0.40
// here; removed in round 6, callers without a workspace get the 32-channel kernel.)  The lists are cut per 8 x 8 quadrant FIRST
// (ql_build_kernel: the same conservative extent test, order kept; 4 B per (quadrant, entry) pair), and each wave of the composite walks
// its own quadrant's list with its own LDS ring (3 chunks of 8 entries: records + feature rows by LDS-DMA two chunks ahead, ids three
// ahead): no __syncthreads in the kernel, a saturated quadrant's wave simply leaves, pairs are formed over the whole quadrant list
// (one padded half pair per quadrant instead of one per batch).  Same arithmetic: per pixel the fmaf chain over the list in order.
//
// quadrant lists: region of tile t = qids[4 * tile_start[t], 4 * tile_start[t + 1]); quadrant q uses the q-th quarter of it, qcnt[t][q] entries.
__global__ __launch_bounds__(256) void ql_build_kernel(const Cam* __restrict__ cams, Geo geo, const int32_t* __restrict__ tile_start,
                                                       const int32_t* __restrict__ ids, int64_t cap_d, const float* __restrict__ rec, int64_t G,
                                                       int32_t* __restrict__ qids, int32_t* __restrict__ qcnt) {
  __shared__ int s_w[4][4];  // [quadrant][wave] survivors of the current slice
  const int v = blockIdx.y, tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const Cam& c = cams[v];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* ts = tile_start + (int64_t)v * (geo.T + 2);
  const int32_t* idp = ids + (int64_t)v * cap_d;
  const int beg = ts[tile], end = ts[tile + 1], len = end - beg;
  int32_t* qo = qids + (int64_t)v * 4 * cap_d + 4 * (int64_t)beg;
  const float alpha_min = c.alpha_min;
  const float tx0 = (float)(tx * TILE), ty0 = (float)(ty * TILE);
  int run[4] = {0, 0, 0, 0};
  for (int b = beg; b < end; b += 256) {
    const int i = b + threadIdx.x;
    unsigned bits = 0;
    int id = 0;
    if (i < end) {
      id = idp[i];
      const float4* rp = (const float4*)(rec + 12 * ((int64_t)v * G + id));
      const float4 r0 = rp[0], r1 = rp[1];
      // alpha >= alpha_min  <=>  sigma <= L = ln(opacity / alpha_min); on that ellipse |dx| <= sqrt(2 L c / det), |dy| <= sqrt(2 L a / det).
      // Conservative (margins far above the rounding of exp_det and of this bound): an entry dropped here can never pass the per-pixel
      // test of the composite, an entry kept needlessly only costs time.
      const float det = conic_det(r1.x, r1.y, r1.z);
      const float L = logf(r1.w / alpha_min) * 1.001f + 0.001f;
      bits = 0xfu;
      if (L < 0.f) bits = 0;
      else if (det > 0.f && L == L) {
        const float ex = sqrtf(2.0f * L * r1.z / det) + 0.01f, ey = sqrtf(2.0f * L * r1.x / det) + 0.01f;
        if (ex == ex && ey == ey) {
          const bool xl = r0.x - ex <= tx0 + 7.5f && r0.x + ex >= tx0 + 0.5f, xr = r0.x - ex <= tx0 + 15.5f && r0.x + ex >= tx0 + 8.5f;
          const bool yt = r0.y - ey <= ty0 + 7.5f && r0.y + ey >= ty0 + 0.5f, yb = r0.y - ey <= ty0 + 15.5f && r0.y + ey >= ty0 + 8.5f;
          bits = (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
        }
      }
    }
    unsigned long long m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      m[q] = __ballot((bits >> q) & 1u);
      if (lane == 0) s_w[q][wave] = __popcll(m[q]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int off = run[q] + __popcll(m[q] & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; ++w) off += s_w[q][w];
      if ((bits >> q) & 1u) qo[(int64_t)q * len + off] = id;
      run[q] += s_w[q][0] + s_w[q][1] + s_w[q][2] + s_w[q][3];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) qcnt[((int64_t)v * geo.T + tile) * 4 + threadIdx.x] = run[threadIdx.x];
}

template <int NP>
__global__ __launch_bounds__(256, 2) void composite_feat5_kernel(const Cam* __restrict__ cams, Geo geo, const int32_t* __restrict__ tile_start,
                                                                 const int32_t* __restrict__ qids, const int32_t* __restrict__ qcnt, int64_t cap_d,
                                                                 const float* __restrict__ rec, const float* __restrict__ feats, int channels, int64_t G,
                                                                 int V, float* __restrict__ out, float* __restrict__ out_alpha) {
  constexpr int CH = 8, R = 3, CW = 32 * NP;  // entries per chunk, chunks in the ring, channels per workgroup
  __shared__ __attribute__((aligned(16))) float s_rec[4][R][CH][8];  // per wave: {mx, my, depth, 0 | conic a, b, c, opacity}
  __shared__ __attribute__((aligned(16))) float s_f[4][R][CH][CW];   // per wave: feature rows (column cw = channel window of block cw / 32)
  const int v = blockIdx.z;
  const Cam& c = cams[v];
  const int tile = blockIdx.x, tx = tile % geo.gw, ty = tile / geo.gw;
  const int ch0 = min((int)blockIdx.y * CW, max(0, channels - CW)), nch = min(CW, channels - ch0);
  const int last_off = nch - 32;  // (shifted last window, as in composite_feat4_kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qx0 = tx * TILE + (wave & 1) * 8, qy0 = ty * TILE + (wave >> 1) * 8;
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const int width = c.width, height = c.height;
  const bool inside = px < width && py < height;
  const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
  const int32_t* ts = tile_start + (int64_t)v * (geo.T + 2);
  const int beg = ts[tile], len = ts[tile + 1] - beg;
  const int n = __builtin_amdgcn_readfirstlane(qcnt[((int64_t)v * geo.T + tile) * 4 + wave]);
  const int32_t* ql = qids + (int64_t)v * 4 * cap_d + 4 * (int64_t)beg + (int64_t)wave * len;
  const float alpha_min = c.alpha_min, alpha_max = c.alpha_max, t_min = c.t_min;
  float T = 1.0f, O = 0.f;
  f32x16 acc[2][NP];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int nb = 0; nb < NP; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][nb][r] = 0.f;
  bool done = !inside;
  const __amdgpu_buffer_rsrc_t r_rec = __builtin_amdgcn_make_buffer_rsrc((void*)rec, (short)0, (int)((int64_t)V * G * 48), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_feat = __builtin_amdgcn_make_buffer_rsrc((void*)feats, (short)0, (int)((int64_t)G * channels * 4), 0x00020000);
  const unsigned vg48 = (unsigned)((int64_t)v * G * 48);
  const int nchunks = (n + CH - 1) / CH;

  // ids of chunk k: lane l < CH holds the id of list entry k * CH + l (0 beyond the list: a valid row whose weights are forced to zero)
  auto load_ids = [&](int k) -> int { return (lane < CH && k * CH + lane < n) ? ql[k * CH + lane] : 0; };
  // DMAs of chunk k (ids in `idv`): records = 16 pieces (lane = 2 * entry + half, lanes 0..15 only), feature rows = 8 * (CW / 4) pieces =
  // NP full wave instructions (piece p = entry * (CW / 4) + 16-byte column)
  auto issue = [&](int idv, int buf) {
    const int idr = __shfl(idv, lane >> 1);
    if (lane < 2 * CH)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rec, (lds_ptr_t)&s_rec[wave][buf][0][0], 16, vg48 + (unsigned)idr * 48u + (unsigned)(lane & 1) * 16u, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int p = k * 64 + lane;
      const int j = p / (CW / 4), cw = (p - j * (CW / 4)) * 4;
      const int co = (cw >> 5) == NP - 1 ? last_off + (cw & 31) : cw;
      const unsigned off = ((unsigned)__shfl(idv, j) * (unsigned)channels + (unsigned)(ch0 + co)) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_feat, (lds_ptr_t)((char*)&s_f[wave][buf][0][0] + k * 1024), 16, off, 0, 0, 0);
    }
  };

  const bool lo = lane < 32;
  auto other_half = [&](float x) {  // the value lane ^ 32 holds (single-operand v_permlane32_swap: right whichever registers are picked)
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, lo ? sw[1] : sw[0]);
  };

  // prologue: chunks 0 and 1 on their way, ids of chunk 2 in a register
  int idv_next = 0;  // ids of the chunk whose DMAs are issued next
  if (nchunks > 0) {
    int i0 = load_ids(0), i1 = load_ids(1);
    idv_next = load_ids(2);
    issue(i0, 0);
    if (nchunks > 1) issue(i1, 1);
  }
  for (int k = 0; k < nchunks; ++k) {
    // chunk k's DMAs (issued two iterations ago) and the ids of chunk k + 2 have landed; chunk k + 1's NP + 1 DMAs may still be in flight
    if (k + 1 < nchunks) {
      if (NP + 1 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (NP + 1 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (NP + 1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (NP + 1 == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (NP + 1 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (__ballot(!done) == 0ull) break;  // the quadrant is saturated
    const int buf = k % R;
    if (k + 2 < nchunks) {
      const int idv = idv_next;
      idv_next = load_ids(k + 3);
      issue(idv, (k + 2) % R);
    }
    const float (*rb)[8] = s_rec[wave][buf];
    const float (*fb)[CW] = s_f[wave][buf];
#pragma unroll
    for (int pr = 0; pr < CH / 2; ++pr) {
      float w[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * pr + e;
        // branch-free per lane (round 6; four nested tests before: the same operations in the same order for every lane that blends,
        // a zero weight for the others -- O + 0 and T stay as they are)
        const float dx = rb[j][0] - pxf, dy = rb[j][1] - pyf;
        const float4 co = *(const float4*)&rb[j][4];
        const float sigma = conic_sigma(co.x, co.y, co.z, dx, dy);
        const float a = fminf(alpha_max, co.w * exp_det_sel(-sigma));
        const float nT = __builtin_fmaf(-T, a, T);
        const bool reach = !done && k * CH + j < n && sigma >= 0.0f && a >= alpha_min;
        const bool sat = reach && nT <= t_min;
        done = done || sat;
        const bool ok = reach && !sat;
        const float wv = ok ? a * T : 0.f;
        O += wv;
        T = ok ? nT : T;
        w[e] = wv;
      }
      if (__ballot(w[0] != 0.f || w[1] != 0.f) == 0ull) continue;
      const float w1_other = other_half(w[1]), w0_other = other_half(w[0]);
      const float a_lo = lo ? w[0] : w1_other, a_hi = lo ? w0_other : w[1];
      const float* frow = &fb[2 * pr + (lo ? 0 : 1)][lane & 31];
      float b[NP];
#pragma unroll
      for (int nb = 0; nb < NP; ++nb) b[nb] = frow[32 * nb];
      // a 32-pixel block (the quadrant's upper / lower four rows) that neither entry of the pair reaches keeps its sums: its MFMAs are skipped
      if (__ballot(a_lo != 0.f) != 0ull) {
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, b[nb], acc[0][nb], 0, 0, 0);
      }
      if (__ballot(a_hi != 0.f) != 0ull) {
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, b[nb], acc[1][nb], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (an early exit leaves DMAs in flight: they must land before the LDS is released)
  // C/D layout: column (channel) = lane & 31, row (pixel of the 32-pixel block) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const size_t hw = (size_t)width * height;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int ox = qx0 + (p & 7), oy = qy0 + (p >> 3);
      if (ox < width && oy < height) {
        float* o = out + ((size_t)v * hw + (size_t)oy * width + ox) * channels + ch0 + (lane & 31);
#pragma unroll
        for (int nb = 0; nb < NP; ++nb) o[nb == NP - 1 ? last_off : 32 * nb] = acc[a][nb][r];
      }
    }
  if (inside && blockIdx.y == 0 && out_alpha) out_alpha[(size_t)v * hw + (size_t)py * width + px] = O;
}

// ---- C ABI -------------------------------------------------------------------------------------------------------
extern "C" int siu3r_raster_geometry(int width, int height, int64_t G, int32_t* out8) {
  SIU3R_CHECK(out8 && width > 0 && height > 0 && G >= 0, "raster_geometry: bad arguments");
  const Geo g = make_geo(width, height);
  out8[0] = g.gw; out8[1] = g.gh; out8[2] = g.T; out8[3] = g.cb; out8[4] = g.NB;
  out8[5] = (int32_t)cdiv64(G > 0 ? G : 1, RS_CH);   // radix-sort chunks (columns of the [V,256,.] histogram table)
  out8[6] = (int32_t)cdiv64(G > 0 ? G : 1, BN_CH);   // binning chunks (columns of the [V,NB,.] table)
  out8[7] = (int32_t)sizeof(siu3r_raster_cam);
  return 0;
}

// gsplat family, pose handed over in DEVICE memory: the per-view world->camera matrices [V,4,4] and pixel-unit intrinsics [V,3,3] (row-major, as
// gsplat.rasterization receives them) overwrite those fields of the uploaded camera blocks -- the host never reads the pose back
__global__ void cam_pose_kernel(Cam* cams, int V, const float* viewmats, const float* Ks) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  for (int i = 0; i < 16; ++i) cams[v].w2c[i] = viewmats[16 * v + i];
  cams[v].fx = Ks[9 * v + 0];
  cams[v].fy = Ks[9 * v + 4];
  cams[v].cx = Ks[9 * v + 2];
  cams[v].cy = Ks[9 * v + 5];
}

// Both families, pose handed over as the REFERENCE holds it (gaussian_renderer.py:29-41): camera-to-world extrinsics [V,4,4] and NORMALISED
// intrinsics [V,3,3] in device memory.  One thread per view derives what the host-side preparation of cuda_splatting.render_cuda /
// SplattingCUDA.forward computes -- world->camera = inverse(extrinsics with the translation scaled by t_scale, :43-44), the camera centre,
// K2: field of view from the K^-1 edge rays (utils/projection.py:247-261), tan(fov / 2), the [0, 1]-depth projection matrix
// (cuda_splatting.py:16-43; near / far from the uploaded block's k2_near / k2_far) and P = Proj * W2C; K3: pixel-unit fx, fy, cx, cy --
// and writes it into the uploaded camera blocks, so that the host never reads the pose (no .cpu() per render call).  Evaluated in fp64
// and rounded once: the reference evaluates the same formulas in fp32 on its device; both are parameter preparation, the values agree
// to fp32 rounding (tests read the finished block back and hand it to the oracle).
__device__ inline bool inv4x4_d(const double* m, double* o) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  if (det == 0.0) return false;
  const double r = 1.0 / det;
  for (int i = 0; i < 16; ++i) o[i] = inv[i] * r;
  return true;
}
__global__ void cam_pose_c2w_kernel(Cam* cams, int V, const float* c2w, const float* Kn, float t_scale) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  Cam& c = cams[v];
  double E[16], W[16];
  for (int i = 0; i < 16; ++i) E[i] = (double)c2w[16 * v + i];
  // (the reference scales the translation in fp32, gaussian_renderer.py:44: the product is rounded to fp32 first)
  E[3] = (double)((float)E[3] * t_scale);
  E[7] = (double)((float)E[7] * t_scale);
  E[11] = (double)((float)E[11] * t_scale);
  if (!inv4x4_d(E, W))
    for (int i = 0; i < 16; ++i) W[i] = __builtin_nan("");  // a singular pose poisons the view instead of rendering something
  for (int i = 0; i < 16; ++i) c.w2c[i] = (float)W[i];
  c.campos[0] = (float)E[3];
  c.campos[1] = (float)E[7];
  c.campos[2] = (float)E[11];
  const float* K = Kn + 9 * v;
  if (c.mode == 0) {
    // rays K^-1 (x, y, 1) of the four edge midpoints, normalised; fov = acos(left . right), acos(top . bottom)
    const double a = K[0], b = K[1], cc = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], k = K[8];
    const double det = a * (e * k - f * h) - b * (d * k - f * g) + cc * (d * h - e * g), rd = 1.0 / det;
    const double Ki[9] = {(e * k - f * h) * rd, (cc * h - b * k) * rd, (b * f - cc * e) * rd, (f * g - d * k) * rd, (a * k - cc * g) * rd,
                          (cc * d - a * f) * rd, (d * h - e * g) * rd, (b * g - a * h) * rd, (a * e - b * d) * rd};
    auto ray = [&](double x, double y, double* r) {
      r[0] = Ki[0] * x + Ki[1] * y + Ki[2];
      r[1] = Ki[3] * x + Ki[4] * y + Ki[5];
      r[2] = Ki[6] * x + Ki[7] * y + Ki[8];
      const double n = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
      r[0] /= n;
      r[1] /= n;
      r[2] /= n;
    };
    double l[3], r[3], t[3], bt[3];
    ray(0.0, 0.5, l);
    ray(1.0, 0.5, r);
    ray(0.5, 0.0, t);
    ray(0.5, 1.0, bt);
    auto clamp1 = [](double x) { return x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x); };
    const double fov_x = acos(clamp1(l[0] * r[0] + l[1] * r[1] + l[2] * r[2])), fov_y = acos(clamp1(t[0] * bt[0] + t[1] * bt[1] + t[2] * bt[2]));
    const double tan_x = tan(0.5 * fov_x), tan_y = tan(0.5 * fov_y);
    c.tanfovx = (float)tan_x;
    c.tanfovy = (float)tan_y;
    const double near = c.k2_near, far = c.k2_far;
    const double top = tan_y * near, right = tan_x * near;
    double P[16] = {0};
    P[0] = 2.0 * near / (2.0 * right);
    P[5] = 2.0 * near / (2.0 * top);
    P[2] = 0.0;   // (right + left) / (right - left) with left = -right
    P[6] = 0.0;
    P[14] = 1.0;
    P[10] = far / (far - near);
    P[11] = -(far * near) / (far - near);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int q = 0; q < 4; ++q) s += P[4 * i + q] * W[4 * q + j];
        c.proj[4 * i + j] = (float)s;
      }
  } else {
    c.fx = K[0] * (float)c.width;
    c.fy = K[4] * (float)c.height;
    c.cx = K[2] * (float)c.width;
    c.cy = K[5] * (float)c.height;
  }
}

static int project_impl(const siu3r_raster_cam* cams_host, int V, void* cams_dev, int64_t G, const float* means, const float* cov,
                        int cov_stride, const float* opacities, const float* colors, int channels, int sh_planar, float* rec,
                        int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys, uint64_t* stats, void* stream,
                        const float* viewmats_dev, const float* Ks_dev, const float* c2w_dev = nullptr, float t_scale = 1.f);

extern "C" int siu3r_raster_project(const siu3r_raster_cam* cams_host, int V, void* cams_dev, int64_t G, const float* means, const float* cov,
                                    int cov_stride, const float* opacities, const float* colors, int channels, int sh_planar, float* rec,
                                    int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys, uint64_t* stats, void* stream) {
  return project_impl(cams_host, V, cams_dev, G, means, cov, cov_stride, opacities, colors, channels, sh_planar, rec, radii, rect, tiles_touched, keys, stats,
                      stream, nullptr, nullptr);
}

extern "C" int siu3r_raster_project_dp(const siu3r_raster_cam* cams_host, int V, void* cams_dev, const float* viewmats_dev, const float* Ks_dev, int64_t G,
                                       const float* means, const float* cov, int cov_stride, const float* opacities, const float* colors, int channels,
                                       int sh_planar, float* rec, int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys,
                                       uint64_t* stats, void* stream) {
  SIU3R_CHECK(viewmats_dev && Ks_dev, "raster_project_dp: null pose pointer");
  SIU3R_CHECK(cams_host && cams_host[0].mode == 1, "raster_project_dp: device-side poses belong to the gsplat family (mode 1)");
  return project_impl(cams_host, V, cams_dev, G, means, cov, cov_stride, opacities, colors, channels, sh_planar, rec, radii, rect, tiles_touched, keys, stats,
                      stream, viewmats_dev, Ks_dev);
}

extern "C" int siu3r_raster_project_c2w(const siu3r_raster_cam* cams_host, int V, void* cams_dev, const float* c2w_dev, const float* Kn_dev, float t_scale,
                                        int64_t G, const float* means, const float* cov, int cov_stride, const float* opacities, const float* colors,
                                        int channels, int sh_planar, float* rec, int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys,
                                        uint64_t* stats, void* stream) {
  SIU3R_CHECK(c2w_dev && Kn_dev, "raster_project_c2w: null pose pointer");
  if (cams_host && V > 0 && cams_host[0].mode == 0)
    for (int v = 0; v < V; ++v)
      SIU3R_CHECK(cams_host[v].k2_near > 0.f && cams_host[v].k2_far > cams_host[v].k2_near, "raster_project_c2w: view %d needs 0 < k2_near < k2_far", v);
  return project_impl(cams_host, V, cams_dev, G, means, cov, cov_stride, opacities, colors, channels, sh_planar, rec, radii, rect, tiles_touched, keys, stats,
                      stream, nullptr, Kn_dev, c2w_dev, t_scale);
}

static int project_impl(const siu3r_raster_cam* cams_host, int V, void* cams_dev, int64_t G, const float* means, const float* cov,
                        int cov_stride, const float* opacities, const float* colors, int channels, int sh_planar, float* rec,
                        int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys, uint64_t* stats, void* stream,
                        const float* viewmats_dev, const float* Ks_dev, const float* c2w_dev, float t_scale) {
  if (int rc = check_views(cams_host, V, "raster_project")) return rc;
  SIU3R_CHECK(cams_dev && stats, "raster_project: null pointer");
  SIU3R_CHECK(G >= 0 && G < (1ll << 31), "raster_project: G = %ld out of range", (long)G);
  SIU3R_CHECK(G == 0 || (means && cov && opacities && rec && radii && rect && tiles_touched && keys), "raster_project: null per-Gaussian pointer");
  SIU3R_CHECK(cov_stride == 6 || cov_stride == 9, "raster_project: cov_stride must be 6 (upper triangle) or 9 (3x3)");
  SIU3R_CHECK(cams_host[0].mode == 1 || G == 0 || colors, "raster_project: SH colours missing");
  SIU3R_CHECK(!sh_planar || channels == 25, "raster_project: the planar SH layout [G,3,25] needs 25 coefficients (got %d)", channels);
  SIU3R_CHECK(((uintptr_t)rec & 15) == 0, "raster_project: rec must be 16-byte aligned");
  for (int v = 0; v < V; ++v)
    SIU3R_CHECK(cams_host[v].mode == 1 || G == 0 || (cams_host[v].sh_degree < 0 ? (channels == 1 && !sh_planar && cams_host[v].sh_degree == cams_host[0].sh_degree)
                                                                                 : channels >= (cams_host[v].sh_degree + 1) * (cams_host[v].sh_degree + 1)),
                "raster_project: too few SH coefficients (or sh_degree < 0 = precomputed [G,3] colours: channels must be 1, for every view of the call)");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyAsync(cams_dev, cams_host, sizeof(Cam) * V, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemsetAsync(stats, 0, sizeof(uint64_t) * ST_N * V, s) != hipSuccess) {
    siu3r_set_error("raster_project: camera upload / stats reset failed");
    return 2;
  }
  if (viewmats_dev) hipLaunchKernelGGL(cam_pose_kernel, dim3((V + 63) / 64), dim3(64), 0, s, (Cam*)cams_dev, V, viewmats_dev, Ks_dev);
  if (c2w_dev) hipLaunchKernelGGL(cam_pose_c2w_kernel, dim3((V + 63) / 64), dim3(64), 0, s, (Cam*)cams_dev, V, c2w_dev, Ks_dev, t_scale);
  // SH floats the views of the call may read (the block is loaded once per Gaussian, at the first view that sees it)
  int nf_chunk = 0;
  for (int v = 0; v < V; ++v) {
    const int deg = cams_host[v].sh_degree;
    const int ncf = (deg > 3 && cams_host[v].sh_band4) ? 25 : (deg > 2 ? 16 : (deg > 1 ? 9 : (deg > 0 ? 4 : 1)));
    if (cams_host[v].mode == 0 && ncf * 3 > nf_chunk) nf_chunk = ncf * 3;
  }
  if (G > 0)
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)cdiv64(G, 256), (V + PV - 1) / PV), dim3(256), 0, s, (const Cam*)cams_dev, V, nf_chunk, G, means, cov,
                       cov_stride, opacities, colors, channels, sh_planar, rec, radii, rect, tiles_touched, keys, (unsigned long long*)stats);
  SIU3R_LAUNCH_CHECK("siu3r_raster_project");
  return 0;
}

extern "C" int siu3r_raster_sort(int V, int64_t G, uint32_t* keys_a, uint32_t* keys_b, int32_t* ids_a, int32_t* ids_b, int32_t* rs_hist,
                                 int32_t* rs_tot, const uint64_t* stats, void* stream) {
  SIU3R_CHECK(V >= 1 && G >= 0, "raster_sort: bad sizes");
  if (G == 0) return 0;
  SIU3R_CHECK(keys_a && keys_b && ids_a && ids_b && rs_hist && rs_tot && stats, "raster_sort: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (int)cdiv64(G, RS_CH);
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t* kin = (pass & 1) ? keys_b : keys_a;
    uint32_t* kout = (pass & 1) ? keys_a : keys_b;
    const int32_t* iin = pass == 0 ? nullptr : ((pass & 1) ? ids_b : ids_a);  // pass 0: the payload is the index itself
    int32_t* iout = (pass & 1) ? ids_a : ids_b;
    const unsigned long long* nv = pass == 0 ? nullptr : (const unsigned long long*)stats;  // pass 0 drops the culled keys: n visible ones remain
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nchunks, V), dim3(256), 0, s, kin, rs_hist, G, pass * 8, nchunks, nv);
    hipLaunchKernelGGL(row_scan_kernel, dim3(64, V), dim3(256), 0, s, rs_hist, rs_tot, 256, nchunks);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nchunks, V), dim3(256), 0, s, kin, iin, kout, iout, rs_hist, rs_tot, G, pass * 8, nchunks, nv);
  }
  SIU3R_LAUNCH_CHECK("siu3r_raster_sort");
  return 0;  // four passes: the sorted keys / ids are back in keys_a / ids_a
}

extern "C" int siu3r_raster_bin(const siu3r_raster_cam* cams_host, int V, int64_t G, const uint32_t* keys, const int32_t* ids, const int32_t* rect,
                                int32_t* bin_hist, int32_t* bin_tot, int32_t* bin_start, void* entries, int64_t cap_e, uint64_t* stats,
                                void* stream) {
  if (int rc = check_views(cams_host, V, "raster_bin")) return rc;
  SIU3R_CHECK(bin_hist && bin_tot && bin_start && stats && cap_e > 0 && cap_e < (1ll << 31), "raster_bin: bad arguments");
  SIU3R_CHECK(G == 0 || (keys && ids && rect && entries), "raster_bin: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const Geo geo = make_geo(cams_host[0].width, cams_host[0].height);
  SIU3R_CHECK(geo.NB <= NB_MAX, "raster_bin: frame of %d x %d tiles exceeds the coarse-bin table", geo.gw, geo.gh);
  const int nchunks = (int)cdiv64(G > 0 ? G : 1, BN_CH);
  if (G == 0) {
    if (hipMemsetAsync(bin_start, 0, sizeof(int32_t) * (size_t)V * (geo.NB + 1), s) != hipSuccess) {
      siu3r_set_error("raster_bin: memset failed");
      return 2;
    }
    return 0;
  }
  hipLaunchKernelGGL(bin_count_kernel, dim3(nchunks, V), dim3(256), 0, s, geo, keys, ids, rect, bin_hist, G, nchunks, (const unsigned long long*)stats);
  hipLaunchKernelGGL(row_scan_kernel, dim3((geo.NB + 3) / 4, V), dim3(256), 0, s, bin_hist, bin_tot, geo.NB, nchunks);
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(nchunks, V), dim3(256), (size_t)geo.NB * 40, s, geo, keys, ids, rect, bin_hist, bin_tot, bin_start,
                     (uint2*)entries, cap_e, (unsigned long long*)stats, G, nchunks);
  SIU3R_LAUNCH_CHECK("siu3r_raster_bin");
  return 0;
}

extern "C" int siu3r_raster_composite_rgb(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* bin_start,
                                          const void* entries, int64_t cap_e, const float* rec, float* image, float* out_depth, float* out_alpha,
                                          int32_t* n_touched, void* stream) {
  if (int rc = check_views(cams_host, V, "raster_composite_rgb")) return rc;
  const bool k3 = cams_host[0].mode == 1;  // gsplat family: image is [V,H,W,3] channel-last, no background, out_depth unused (may be NULL)
  SIU3R_CHECK(cams_dev && bin_start && image && (out_depth || k3) && out_alpha && (G == 0 || (entries && rec)), "raster_composite_rgb: null pointer");
  SIU3R_CHECK(!(k3 && n_touched), "raster_composite_rgb: n_touched belongs to the 3DGS family (mode 0)");
  hipStream_t s = (hipStream_t)stream;
  const Geo geo = make_geo(cams_host[0].width, cams_host[0].height);
  if (n_touched && G > 0 && hipMemsetAsync(n_touched, 0, sizeof(int32_t) * (size_t)V * G, s) != hipSuccess) {
    siu3r_set_error("raster_composite_rgb: memset failed");
    return 2;
  }
  if (k3)
    hipLaunchKernelGGL((composite_rgb_kernel<false, true>), dim3(geo.T, V), dim3(256), 0, s, (const Cam*)cams_dev, geo, bin_start, (const uint2*)entries, cap_e, rec, G,
                       image, out_depth, out_alpha, n_touched);
  else if (n_touched)
    hipLaunchKernelGGL(composite_rgb_kernel<true>, dim3(geo.T, V), dim3(256), 0, s, (const Cam*)cams_dev, geo, bin_start, (const uint2*)entries, cap_e, rec, G,
                       image, out_depth, out_alpha, n_touched);
  else
    hipLaunchKernelGGL(composite_rgb_kernel<false>, dim3(geo.T, V), dim3(256), 0, s, (const Cam*)cams_dev, geo, bin_start, (const uint2*)entries, cap_e, rec, G,
                       image, out_depth, out_alpha, n_touched);
  SIU3R_LAUNCH_CHECK("siu3r_raster_composite_rgb");
  return 0;
}

extern "C" int siu3r_raster_tile_lists(const siu3r_raster_cam* cams_host, int V, const int32_t* bin_start, const void* entries, int64_t cap_e,
                                       int32_t* tile_count, int32_t* tile_start, int32_t* ids, int64_t cap_d, uint64_t* stats, void* stream) {
  if (int rc = check_views(cams_host, V, "raster_tile_lists")) return rc;
  SIU3R_CHECK(bin_start && entries && tile_count && tile_start && ids && stats && cap_d > 0 && cap_d < (1ll << 31), "raster_tile_lists: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const Geo geo = make_geo(cams_host[0].width, cams_host[0].height);
  hipLaunchKernelGGL(tl_count_kernel, dim3(geo.T, V), dim3(256), 0, s, geo, bin_start, (const uint2*)entries, cap_e, tile_count);
  hipLaunchKernelGGL(tl_scan_kernel, dim3(1, V), dim3(1024), 0, s, tile_count, tile_start, geo.T, cap_d, (unsigned long long*)stats);
  hipLaunchKernelGGL(tl_write_kernel, dim3(geo.T, V), dim3(256), 0, s, geo, bin_start, (const uint2*)entries, cap_e, tile_start, ids, cap_d);
  SIU3R_LAUNCH_CHECK("siu3r_raster_tile_lists");
  return 0;
}

// tuning switches (siu3r_raster_tune; tests and tools A/B the forms through it -- no environment variable is read on the render path)
static int g_feat_form = 0;  // 0: matrix-core form where it applies; 1: the 32-channel kernel everywhere
static int g_feat_np = 6;    // accumulator blocks (of 32 channels) per chunk of the matrix-core form, 1 .. 6
extern "C" int siu3r_raster_tune(int key, int value) {
  SIU3R_CHECK(key == 0 || key == 1, "raster_tune: unknown key %d", key);
  if (key == 0) g_feat_form = value == 1 ? 1 : 0;
  else g_feat_np = value < 1 ? 6 : (value > 6 ? 6 : value);
  return 0;
}

extern "C" int siu3r_raster_composite_feat(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* tile_start,
                                           const int32_t* ids, int64_t cap_d, const float* rec, const float* feats, int channels, float* out,
                                           float* out_alpha, void* stream) {
  if (int rc = check_views(cams_host, V, "raster_composite_feat")) return rc;
  SIU3R_CHECK(cams_dev && tile_start && (ids || cap_d == 0 || G == 0) && ((rec && feats) || G == 0) && out && channels > 0, "raster_composite_feat: bad arguments");  // (an empty scene has empty lists: nothing is dereferenced)
  const Geo geo = make_geo(cams_host[0].width, cams_host[0].height);
  const int nchunk = (channels + CHUNK - 1) / CHUNK;
  SIU3R_CHECK(nchunk <= 65535 && V <= 65535, "raster_composite_feat: too many channel chunks / views");
  hipLaunchKernelGGL(composite_feat_kernel, dim3(geo.T, nchunk, V), dim3(256), 0, (hipStream_t)stream, (const Cam*)cams_dev, geo, tile_start, ids, cap_d, rec,
                     feats, channels, G, out, out_alpha);
  SIU3R_LAUNCH_CHECK("siu3r_raster_composite_feat");
  return 0;
}

extern "C" int64_t siu3r_raster_composite_feat_ws_bytes(int width, int height, int V, int64_t cap_d) {
  const Geo geo = make_geo(width, height);
  return (int64_t)V * (16 * cap_d + 16 * (int64_t)geo.T);
}

extern "C" int siu3r_raster_composite_feat_ws(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* tile_start,
                                              const int32_t* ids, int64_t cap_d, const float* rec, const float* feats, int channels, float* out,
                                              float* out_alpha, void* ws, int64_t ws_bytes, void* stream) {
  if (int rc = check_views(cams_host, V, "raster_composite_feat_ws")) return rc;
  SIU3R_CHECK(cams_dev && tile_start && (ids || cap_d == 0 || G == 0) && ((rec && feats) || G == 0) && out && channels > 0, "raster_composite_feat_ws: bad arguments");
  const Geo geo = make_geo(cams_host[0].width, cams_host[0].height);
  // (the matrix-core form addresses records and features through buffer resources: 32-bit byte offsets; rows need only 4-byte alignment)
  const bool fits32 = (int64_t)V * G * 48 < (1ll << 31) * 2 - 64 && (int64_t)G * channels * 4 < (1ll << 31) * 2 - 64 && (((uintptr_t)feats) & 3) == 0;
  const int64_t need = (int64_t)V * (16 * cap_d + 16 * (int64_t)geo.T);
  if (g_feat_form == 1 || channels < 32 || !fits32 || !ws || ws_bytes < need || (((uintptr_t)ws) & 3))
    return siu3r_raster_composite_feat(cams_host, V, cams_dev, G, tile_start, ids, cap_d, rec, feats, channels, out, out_alpha, stream);
  hipStream_t s = (hipStream_t)stream;
  int32_t* qids = (int32_t*)ws;
  int32_t* qcnt = qids + (int64_t)V * 4 * cap_d;
  hipLaunchKernelGGL(ql_build_kernel, dim3(geo.T, V), dim3(256), 0, s, (const Cam*)cams_dev, geo, tile_start, ids, cap_d, rec, G, qids, qcnt);
  const int np_max = g_feat_np;
  const int np = channels >= 32 * np_max ? np_max : (channels + 31) / 32;
  const int nchunk = (channels + 32 * np - 1) / (32 * np);
  SIU3R_CHECK(nchunk <= 65535 && V <= 65535, "raster_composite_feat_ws: too many channel chunks / views");
  const dim3 grid(geo.T, nchunk, V);
#define SIU3R_F5(N) hipLaunchKernelGGL(composite_feat5_kernel<N>, grid, dim3(256), 0, s, (const Cam*)cams_dev, geo, tile_start, qids, qcnt, cap_d, rec, feats, channels, G, V, out, out_alpha)
  switch (np) {
    case 1: SIU3R_F5(1); break;
    case 2: SIU3R_F5(2); break;
    case 3: SIU3R_F5(3); break;
    case 4: SIU3R_F5(4); break;
    case 5: SIU3R_F5(5); break;
    default: SIU3R_F5(6); break;
  }
#undef SIU3R_F5
  SIU3R_LAUNCH_CHECK("siu3r_raster_composite_feat_ws");
  return 0;
}

extern "C" int siu3r_scale_inplace(float* x, int64_t n, float s, void* stream) {
  SIU3R_CHECK(x || n == 0, "scale_inplace: null pointer");
  if (n > 0) hipLaunchKernelGGL(scale_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, x, n, s);
  SIU3R_LAUNCH_CHECK("siu3r_scale_inplace");
  return 0;
}

extern "C" int siu3r_quat_scale_to_cov6(const float* quats_wxyz, const float* scales, float* cov6, int64_t G, void* stream) {
  SIU3R_CHECK(G == 0 || (quats_wxyz && scales && cov6), "quat_scale_to_cov6: null pointer");
  SIU3R_CHECK(((uintptr_t)quats_wxyz & 15) == 0, "quat_scale_to_cov6: quats must be 16-byte aligned");
  if (G > 0) hipLaunchKernelGGL(quat_scale_cov6_kernel, g1(G), dim3(256), 0, (hipStream_t)stream, G, quats_wxyz, scales, cov6);
  SIU3R_LAUNCH_CHECK("siu3r_quat_scale_to_cov6");
  return 0;
}

static int sh_eval_impl(const float* means, const float* campos3_host, const float* campos3_dev, const float* sh, int ncoef, int degree, float* rgb, int64_t G,
                        void* stream) {
  SIU3R_CHECK(G == 0 || (means && (campos3_host || campos3_dev) && sh && rgb), "sh_eval: null pointer");
  SIU3R_CHECK(degree >= 0 && degree <= 4 && ncoef >= (degree + 1) * (degree + 1), "sh_eval: degree %d needs %d coefficients, got %d", degree,
              (degree + 1) * (degree + 1), ncoef);
  if (G == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float cx = campos3_host ? campos3_host[0] : 0.f, cy = campos3_host ? campos3_host[1] : 0.f, cz = campos3_host ? campos3_host[2] : 0.f;
  SIU3R_CHECK(ncoef <= 49, "sh_eval: at most 49 coefficients (got %d)", ncoef);
  const size_t lds = (size_t)256 * ncoef * 3 * sizeof(float);  // <= 147 KiB
  static bool attr_set = false;
  if (!attr_set) {  // dynamic LDS beyond 64 KiB has to be requested once per kernel
    const int cap = 160 * 1024;
    hipFuncSetAttribute((const void*)sh_eval_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute((const void*)sh_eval_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute((const void*)sh_eval_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute((const void*)sh_eval_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    hipFuncSetAttribute((const void*)sh_eval_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_set = true;
  }
  switch (degree) {
    case 0: hipLaunchKernelGGL(sh_eval_kernel<0>, g1(G), dim3(256), lds, s, G, ncoef, means, cx, cy, cz, sh, rgb, campos3_dev); break;
    case 1: hipLaunchKernelGGL(sh_eval_kernel<1>, g1(G), dim3(256), lds, s, G, ncoef, means, cx, cy, cz, sh, rgb, campos3_dev); break;
    case 2: hipLaunchKernelGGL(sh_eval_kernel<2>, g1(G), dim3(256), lds, s, G, ncoef, means, cx, cy, cz, sh, rgb, campos3_dev); break;
    case 3: hipLaunchKernelGGL(sh_eval_kernel<3>, g1(G), dim3(256), lds, s, G, ncoef, means, cx, cy, cz, sh, rgb, campos3_dev); break;
    default: hipLaunchKernelGGL(sh_eval_kernel<4>, g1(G), dim3(256), lds, s, G, ncoef, means, cx, cy, cz, sh, rgb, campos3_dev); break;
  }
  SIU3R_LAUNCH_CHECK("siu3r_sh_eval");
  return 0;
}

extern "C" int siu3r_sh_eval(const float* means, const float* campos3_host, const float* sh, int ncoef, int degree, float* rgb, int64_t G,
                             void* stream) {
  SIU3R_CHECK(G == 0 || campos3_host, "sh_eval: null camera position");
  return sh_eval_impl(means, campos3_host, nullptr, sh, ncoef, degree, rgb, G, stream);
}

extern "C" int siu3r_sh_eval_dp(const float* means, const float* campos3_dev, const float* sh, int ncoef, int degree, float* rgb, int64_t G,
                                void* stream) {
  SIU3R_CHECK(G == 0 || campos3_dev, "sh_eval_dp: null camera position");
  return sh_eval_impl(means, nullptr, campos3_dev, sh, ncoef, degree, rgb, G, stream);
}

extern "C" int siu3r_blend_background(float* colors, const float* alpha, const float* bg_host, int channels, int64_t pixels, void* stream) {
  SIU3R_CHECK(pixels == 0 || (colors && alpha && bg_host), "blend_background: null pointer");
  SIU3R_CHECK(channels >= 1 && channels <= 3, "blend_background: 1..3 channels (got %d)", channels);
  if (pixels > 0)
    hipLaunchKernelGGL(blend_bg_kernel, g1(pixels * channels), dim3(256), 0, (hipStream_t)stream, pixels * channels, channels, colors, alpha, bg_host[0],
                       channels > 1 ? bg_host[1] : 0.f, channels > 2 ? bg_host[2] : 0.f, (const float*)nullptr);
  SIU3R_LAUNCH_CHECK("siu3r_blend_background");
  return 0;
}

extern "C" int siu3r_blend_background_dp(float* colors, const float* alpha, const float* bg_dev, int channels, int64_t pixels, void* stream) {
  SIU3R_CHECK(pixels == 0 || (colors && alpha && bg_dev), "blend_background_dp: null pointer");
  SIU3R_CHECK(channels >= 1, "blend_background_dp: channels = %d", channels);
  if (pixels > 0)
    hipLaunchKernelGGL(blend_bg_kernel, g1(pixels * channels), dim3(256), 0, (hipStream_t)stream, pixels * channels, channels, colors, alpha, 0.f, 0.f, 0.f, bg_dev);
  SIU3R_LAUNCH_CHECK("siu3r_blend_background_dp");
  return 0;
}
