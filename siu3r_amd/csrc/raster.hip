// Tile-binned Gaussian splat rasterizer for gfx950 -- a fresh design (not a hipify of either CUDA rasterizer).
//
// Replaces, behind the reference's call sites, the two un-vendored CUDA packages:
//   mode 0 "K2": diff-gaussian-rasterization-w-pose semantics (src/models/cuda_splatting.py:90-118):
//                SH->RGB, colour + depth + accumulated opacity + radii + n_touched, pixel centres at integers
//   mode 1 "K3": gsplat semantics (src/models/gaussian_renderer.py:92-106): N-channel features + alphas,
//                near/far cull, eps2d, pixel centres at +0.5, channels rendered in chunks of 32
// Pipeline (all on the caller's stream):
//   project   one lane per Gaussian: EWA projection, extent, tile rect, SH->RGB; per-tile population counts with
//             wavefront-aggregated atomics
//   scan      exclusive scan of the tile counts (single workgroup, wave prefix sums via DPP/shuffles)
//   fill      (depth, id) pairs appended to each touched tile's segment
//   sort      one workgroup per tile: bitonic sort of the 64-bit keys (depth bits << 32 | Gaussian id) in LDS --
//             unique keys, so the order (and hence every integer output) is deterministic run to run
//   composite one 16x16 workgroup per tile: Gaussians staged through LDS in batches of 256, front-to-back alpha
//             blend, wave-ballot early termination
// Arithmetic matches oracle/raster_ref.c operation for operation (same expression order, contraction off, and a
// shared polynomial exp) so that integer outputs are bit-exact and the maps agree to fp32 rounding.
#include "common.h"

namespace {

constexpr int TILE = 16;

struct Cam {
  int mode, width, height;
  float w2c[16], proj[16];
  float tanfovx, tanfovy, campos[3], bg[3];
  int sh_degree, sh_band4;
  float k2_znear_cull;
  float fx, fy, cx, cy, near_plane, far_plane, eps2d, radius_clip, extent_sigma;
  int opacity_aware_extent;
  float alpha_min, alpha_max, t_min, dilation;
};

// exp(x) for x <= 0 with plain fp32 operations only (identical in oracle/raster_ref.c): 2^(x*log2e), argument
// reduced to [-0.5, 0.5], degree-7 Taylor of 2^f (|err| < 1e-7 rel), exact scaling by 2^n.
__device__ __forceinline__ float exp_det(float x) {
  if (x < -87.0f) return 0.0f;
  const float y = x * 1.4426950408889634f;
  const float n = floorf(y + 0.5f);
  const float f = y - n;
  float p = 1.52527338e-5f;
  p = p * f + 1.54035304e-4f;
  p = p * f + 1.33335581e-3f;
  p = p * f + 9.61812911e-3f;
  p = p * f + 5.55041087e-2f;
  p = p * f + 2.40226507e-1f;
  p = p * f + 6.93147181e-1f;
  p = p * f + 1.0f;
  return ldexpf(p, (int)n);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

constexpr int NPART = 8;  // counter sets of the binning atomics (see scan_kernel)
__constant__ float c_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__constant__ float c_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
__constant__ float c_SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

__global__ void project_kernel(Cam c, int64_t G, const float* means, const float* cov6, const float* opac,
                               const float* colors, int channels, float* mean2d, float* conic_op, float* depth,
                               int32_t* radii, int32_t* rect, int32_t* tiles_touched, float* rgb, int32_t* tile_count) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
  const float* V = c.w2c;
  const float m0 = means[3 * g], m1 = means[3 * g + 1], m2 = means[3 * g + 2];
  int rx_i = 0, ry_i = 0, tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
  bool valid = false;
  float mx = 0.f, my = 0.f, ca = 0.f, cb = 0.f, cc = 0.f;
  const float tx = V[0] * m0 + V[1] * m1 + V[2] * m2 + V[3];
  const float ty = V[4] * m0 + V[5] * m1 + V[6] * m2 + V[7];
  const float tz = V[8] * m0 + V[9] * m1 + V[10] * m2 + V[11];
  const int gw = (c.width + TILE - 1) / TILE, gh = (c.height + TILE - 1) / TILE;
  const float opacity = opac[g];
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
    const float sxx = cov6[6 * g], sxy = cov6[6 * g + 1], sxz = cov6[6 * g + 2], syy = cov6[6 * g + 3], syz = cov6[6 * g + 4], szz = cov6[6 * g + 5];
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
  radii[2 * g] = rx_i;
  radii[2 * g + 1] = ry_i;
  tiles_touched[g] = valid ? (tx1 - tx0) * (ty1 - ty0) : 0;
  rect[4 * g] = valid ? tx0 : 0;
  rect[4 * g + 1] = valid ? ty0 : 0;
  rect[4 * g + 2] = valid ? tx1 : 0;
  rect[4 * g + 3] = valid ? ty1 : 0;
  mean2d[2 * g] = mx;
  mean2d[2 * g + 1] = my;
  conic_op[4 * g] = ca;
  conic_op[4 * g + 1] = cb;
  conic_op[4 * g + 2] = cc;
  conic_op[4 * g + 3] = opacity;
  depth[g] = tz;
  if (!valid) return;
  if (c.mode == 0) {
    const float dx = m0 - c.campos[0], dy = m1 - c.campos[1], dz = m2 - c.campos[2];
    const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    const int deg = c.sh_degree;
    // the Gaussian's coefficient block ([coef][rgb], 12 B per coefficient, 4-byte aligned) as 16-byte loads: a lane's
    // block is contiguous, so 12 (19 with band 4) wide loads replace 48 (75) scalar ones that each touched 64 cache lines
    struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
    const float* shp = colors + (size_t)g * channels * 3;
    float sh[76];
    const int nf = ((deg > 3 && c.sh_band4) ? 25 : (deg > 2 ? 16 : (deg > 1 ? 9 : (deg > 0 ? 4 : 1)))) * 3;
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
      rgb[3 * g + ch] = r < 0.0f ? 0.0f : r;
    }
  }
  for (int ty_ = ty0; ty_ < ty1; ++ty_)
    for (int tx_ = tx0; tx_ < tx1; ++tx_)
      atomicAdd(&tile_count[(blockIdx.x & (NPART - 1)) * (gw * gh) + ty_ * gw + tx_], 1);  // one of NPART counter sets: see scan_kernel
}

// The per-tile counters are hot addresses (a 1080p frame has 8160 of them for ~14 M (Gaussian, tile) pairs): device-scope atomics on
// one address serialise.  Workgroup b of project_kernel / fill_kernel therefore uses counter set b % NPART; the sets are summed here,
// and set c of a tile gets the cursor base tile_start + (count of sets < c), so every pair still lands in its tile's range.  Which
// workgroup a Gaussian belongs to is the same in both kernels (same grid), and the per-tile sort makes the final order independent
// of the partition.
// exclusive scan of sum_c tile_count[c][T] -> tile_start[T+1], cursor[c][T]; single workgroup of 1024 threads
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t* tile_count, int32_t* tile_start, int32_t* cursor, int T, int cap) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < T; base += 1024) {
    const int i = base + t;
    int cnt[NPART];
    int v = 0;
#pragma unroll
    for (int c = 0; c < NPART; ++c) {
      cnt[c] = i < T ? tile_count[c * T + i] : 0;
      v += cnt[c];
    }
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int n = __shfl_up(incl, o);
      if (lane >= o) incl += n;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int excl = carry + woff + incl - v;
    if (i < T) {
      tile_start[i] = min(excl, cap);  // ranges are clamped to the pair buffers' capacity (see siu3r_raster_bin)
      int run = excl;
#pragma unroll
      for (int c = 0; c < NPART; ++c) {
        cursor[c * T + i] = run;
        run += cnt[c];
      }
    }
    __syncthreads();
    if (t == 1023) carry = excl + v;
    __syncthreads();
  }
  if (t == 0) {
    tile_start[T] = min(carry, cap);
    tile_start[T + 1] = carry;  // the true pair count: the host compares it with cap after the frame has been enqueued
  }
}

__global__ void fill_kernel(int64_t G, const int32_t* rect, const float* depth, int32_t* cursor, uint64_t* keys, int gw, int T, int cap) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  const int tx0 = rect[4 * g], ty0 = rect[4 * g + 1], tx1 = rect[4 * g + 2], ty1 = rect[4 * g + 3];
  if (tx1 <= tx0 || ty1 <= ty0) return;
  const uint64_t key = ((uint64_t)__float_as_uint(depth[g]) << 32) | (uint32_t)g;
  for (int ty = ty0; ty < ty1; ++ty)
    for (int tx = tx0; tx < tx1; ++tx) {
      const int pos = atomicAdd(&cursor[(blockIdx.x & (NPART - 1)) * T + ty * gw + tx], 1);
      if (pos < cap) keys[pos] = key;
    }
}

// One workgroup per tile: bitonic sort of the tile's (depth bits << 32 | id) keys in LDS.  Three instantiations by
// capacity (a tile is handled by the smallest one that holds its padded length): small tiles get small LDS footprints
// and therefore many resident workgroups.  Every thread is busy in every pass, and two butterfly stages (strides j and
// j/2) are done per LDS round trip on four keys held in registers: half the LDS traffic and barriers of the textbook
// loop.  Tiles beyond 8192 keys fall back to an in-place global-memory bitonic (rare).
constexpr int SORT_CAP = 8192;  // 64 KiB of LDS
__device__ __forceinline__ void cmpx(uint64_t& a, uint64_t& b, bool up) {
  if ((a > b) == up) {
    const uint64_t t = a;
    a = b;
    b = t;
  }
}
template <int CAP, int CAP_PREV>
__global__ __launch_bounds__(256) void sort_kernel(const int32_t* tile_start, uint64_t* keys, int32_t* ids) {
  __shared__ uint64_t s[CAP];
  const int tile = blockIdx.x;
  const int beg = tile_start[tile], n = tile_start[tile + 1] - beg;
  if (n <= 0) return;
  int np = 1, lg = 0;
  while (np < n) {
    np <<= 1;
    ++lg;
  }
  if (np <= CAP_PREV) return;  // a smaller instantiation owns this tile
  if (np <= CAP) {
    for (int i = threadIdx.x; i < np; i += 256) s[i] = i < n ? keys[beg + i] : ~0ull;
    __syncthreads();
    for (int lk = 1; lk <= lg; ++lk) {       // merge width k = 1 << lk
      const int k = 1 << lk;
      int lj = lk - 1;                        // stride j = 1 << lj
      for (; lj >= 1; lj -= 2) {              // two stages per pass: strides j and h = j / 2
        const int j = 1 << lj, h = j >> 1;
        for (int q = threadIdx.x; q < (np >> 2); q += 256) {
          const int low = q & (h - 1);
          const int i0 = ((q >> (lj - 1)) << (lj + 1)) | low;
          const bool up = (i0 & k) == 0;
          uint64_t a = s[i0], b = s[i0 + h], c = s[i0 + j], d = s[i0 + j + h];
          cmpx(a, c, up);
          cmpx(b, d, up);
          cmpx(a, b, up);
          cmpx(c, d, up);
          s[i0] = a;
          s[i0 + h] = b;
          s[i0 + j] = c;
          s[i0 + j + h] = d;
        }
        __syncthreads();
      }
      if (lj == 0) {                          // odd number of stages: the stride-1 stage alone
        for (int q = threadIdx.x; q < (np >> 1); q += 256) {
          const int i0 = q << 1;
          const bool up = (i0 & k) == 0;
          uint64_t a = s[i0], b = s[i0 + 1];
          cmpx(a, b, up);
          s[i0] = a;
          s[i0 + 1] = b;
        }
        __syncthreads();
      }
    }
    for (int i = threadIdx.x; i < n; i += 256) {
      keys[beg + i] = s[i];
      ids[beg + i] = (int32_t)(s[i] & 0xffffffffu);
    }
  } else if (CAP == SORT_CAP) {
    // oversized tile: in-place bitonic on the (virtually padded) global segment; slow path, rare
    uint64_t* k_ = keys + beg;
    for (int k = 2; k <= np; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < np; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const uint64_t a = i < n ? k_[i] : ~0ull, b = l < n ? k_[l] : ~0ull;
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
              if (i < n) k_[i] = b;
              if (l < n) k_[l] = a;
            }
          }
        }
        __threadfence_block();
        __syncthreads();
      }
    for (int i = threadIdx.x; i < n; i += 256) ids[beg + i] = (int32_t)(k_[i] & 0xffffffffu);
  }
}

// K2 composite: colour [3,H,W] + depth + accumulated opacity + n_touched
__global__ __launch_bounds__(256) void composite_rgb_kernel(Cam c, const int32_t* tile_start, const int32_t* ids,
                                                            const float* mean2d, const float* conic_op, const float* depth,
                                                            const float* rgb, float* image, float* out_depth,
                                                            float* out_alpha, int32_t* n_touched) {
  __shared__ float s_xy[256][2], s_co[256][4], s_rgbd[256][4];
  __shared__ int s_id[256];
  const int gw = (c.width + TILE - 1) / TILE;
  const int tile = blockIdx.x, tx = tile % gw, ty = tile / gw;
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int px = tx * TILE + lx, py = ty * TILE + ly;
  const bool inside = px < c.width && py < c.height;
  const float pxf = (float)px, pyf = (float)py;
  const int beg = tile_start[tile], end = tile_start[tile + 1];
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f, O = 0.f;
  bool done = !inside;
  for (int base = beg; base < end; base += 256) {
    if (__syncthreads_count(done) == 256) break;
    const int i = base + threadIdx.x;
    if (i < end) {
      const int g = ids[i];
      s_id[threadIdx.x] = g;
      s_xy[threadIdx.x][0] = mean2d[2 * g];
      s_xy[threadIdx.x][1] = mean2d[2 * g + 1];
      const float4 co = *(const float4*)(conic_op + 4 * (size_t)g);
      s_co[threadIdx.x][0] = co.x; s_co[threadIdx.x][1] = co.y; s_co[threadIdx.x][2] = co.z; s_co[threadIdx.x][3] = co.w;
      s_rgbd[threadIdx.x][0] = rgb[3 * g]; s_rgbd[threadIdx.x][1] = rgb[3 * g + 1]; s_rgbd[threadIdx.x][2] = rgb[3 * g + 2];
      s_rgbd[threadIdx.x][3] = depth[g];
    }
    __syncthreads();
    const int cnt = min(256, end - base);
    for (int j = 0; !done && j < cnt; ++j) {
      const float dx = s_xy[j][0] - pxf, dy = s_xy[j][1] - pyf;
      const float power = -0.5f * (s_co[j][0] * dx * dx + s_co[j][2] * dy * dy) - s_co[j][1] * dx * dy;
      if (power > 0.0f) continue;
      const float a = fminf(c.alpha_max, s_co[j][3] * exp_det(power));
      if (a < c.alpha_min) continue;
      const float nT = T * (1.0f - a);
      if (nT < c.t_min) {
        done = true;
        continue;
      }
      const float w = a * T;
      C0 += s_rgbd[j][0] * w;
      C1 += s_rgbd[j][1] * w;
      C2 += s_rgbd[j][2] * w;
      D += s_rgbd[j][3] * w;
      O += w;
      if (T > 0.5f) atomicAdd(&n_touched[s_id[j]], 1);
      T = nT;
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * c.width + px, hw = (size_t)c.width * c.height;
    image[pix] = C0 + T * c.bg[0];
    image[hw + pix] = C1 + T * c.bg[1];
    image[2 * hw + pix] = C2 + T * c.bg[2];
    out_depth[pix] = D;
    out_alpha[pix] = O;
  }
}

// K3 composite: one 32-channel chunk of the feature matrix per blockIdx.y; colours [H,W,C]; alpha written by chunk 0
constexpr int CHUNK = 32;
__global__ __launch_bounds__(256) void composite_feat_kernel(Cam c, const int32_t* tile_start, const int32_t* ids,
                                                             const float* mean2d, const float* conic_op,
                                                             const float* feats, int channels, float* out, float* out_alpha) {
  __shared__ float s_xy[128][2], s_co[128][4];
  __shared__ float s_f[128][CHUNK + 1];
  const int gw = (c.width + TILE - 1) / TILE;
  const int tile = blockIdx.x, tx = tile % gw, ty = tile / gw;
  const int ch0 = blockIdx.y * CHUNK, nch = min(CHUNK, channels - ch0);
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int px = tx * TILE + lx, py = ty * TILE + ly;
  const bool inside = px < c.width && py < c.height;
  const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
  const int beg = tile_start[tile], end = tile_start[tile + 1];
  float T = 1.0f, O = 0.f;
  float acc[CHUNK];
#pragma unroll
  for (int k = 0; k < CHUNK; ++k) acc[k] = 0.f;
  bool done = !inside;
  for (int base = beg; base < end; base += 128) {
    if (__syncthreads_count(done) == 256) break;
    const int cnt = min(128, end - base);
    if (threadIdx.x < cnt) {
      const int g = ids[base + threadIdx.x];
      s_xy[threadIdx.x][0] = mean2d[2 * g];
      s_xy[threadIdx.x][1] = mean2d[2 * g + 1];
      const float4 co = *(const float4*)(conic_op + 4 * (size_t)g);
      s_co[threadIdx.x][0] = co.x; s_co[threadIdx.x][1] = co.y; s_co[threadIdx.x][2] = co.z; s_co[threadIdx.x][3] = co.w;
    }
    for (int e = threadIdx.x; e < cnt * CHUNK; e += 256) {
      const int j = e / CHUNK, k = e - j * CHUNK;
      s_f[j][k] = k < nch ? feats[(size_t)ids[base + j] * channels + ch0 + k] : 0.f;
    }
    __syncthreads();
    for (int j = 0; !done && j < cnt; ++j) {
      const float dx = s_xy[j][0] - pxf, dy = s_xy[j][1] - pyf;
      const float sigma = 0.5f * (s_co[j][0] * dx * dx + s_co[j][2] * dy * dy) + s_co[j][1] * dx * dy;
      if (sigma < 0.0f) continue;
      const float a = fminf(c.alpha_max, s_co[j][3] * exp_det(-sigma));
      if (a < c.alpha_min) continue;
      const float nT = T * (1.0f - a);
      if (nT <= c.t_min) {
        done = true;
        continue;
      }
      const float w = a * T;
#pragma unroll
      for (int k = 0; k < CHUNK; ++k) acc[k] += s_f[j][k] * w;
      O += w;
      T = nT;
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * c.width + px;
    for (int k = 0; k < nch; ++k) out[pix * channels + ch0 + k] = acc[k];
    if (blockIdx.y == 0 && out_alpha) out_alpha[pix] = O;
  }
}

__global__ void scale_kernel(float* x, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= s;
}

inline dim3 g1(int64_t n, int b = 256) { return dim3((unsigned)cdiv64(n, b)); }

void to_cam(const siu3r_raster_cam* in, Cam* c) {
  c->mode = in->mode; c->width = in->width; c->height = in->height;
  memcpy(c->w2c, in->w2c, sizeof(c->w2c));
  memcpy(c->proj, in->proj, sizeof(c->proj));
  c->tanfovx = in->tanfovx; c->tanfovy = in->tanfovy;
  memcpy(c->campos, in->campos, sizeof(c->campos));
  memcpy(c->bg, in->bg, sizeof(c->bg));
  c->sh_degree = in->sh_degree; c->sh_band4 = in->sh_band4; c->k2_znear_cull = in->k2_znear_cull;
  c->fx = in->fx; c->fy = in->fy; c->cx = in->cx; c->cy = in->cy;
  c->near_plane = in->near_plane; c->far_plane = in->far_plane; c->eps2d = in->eps2d; c->radius_clip = in->radius_clip;
  c->extent_sigma = in->extent_sigma; c->opacity_aware_extent = in->opacity_aware_extent;
  c->alpha_min = in->alpha_min; c->alpha_max = in->alpha_max; c->t_min = in->t_min; c->dilation = in->dilation;
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

template <int DEG>
__global__ void sh_eval_kernel(int64_t G, int ncoef, const float* means, float cx, float cy, float cz, const float* sh, float* rgb) {
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
  // the coefficient block ([coef][rgb], 4-byte aligned) as 16-byte loads
  struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
  const float* shp = sh + (size_t)g * ncoef * 3;
  float c[NB * 3 + 3];
#pragma unroll
  for (int q = 0; q < (NB * 3 + 3) / 4; ++q) {
    if (4 * q + 4 <= ncoef * 3) {
      const f4u t4 = *(const f4u*)(shp + 4 * q);
      c[4 * q] = t4.v[0]; c[4 * q + 1] = t4.v[1]; c[4 * q + 2] = t4.v[2]; c[4 * q + 3] = t4.v[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) c[4 * q + e] = (4 * q + e < ncoef * 3) ? shp[4 * q + e] : 0.f;
    }
  }
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

// ---- C ABI -------------------------------------------------------------------------------------------------------
extern "C" int siu3r_raster_bin(const siu3r_raster_cam* cam, int64_t G, const float* means, const float* cov6,
                                const float* opacities, const float* colors, int channels, float* mean2d, float* conic_op,
                                float* depth, int32_t* radii, int32_t* rect, int32_t* tiles_touched, float* rgb,
                                int32_t* tile_count, int32_t* tile_start, int32_t* cursor, int64_t cap, void* stream) {
  SIU3R_CHECK(cam && tile_count && tile_start && cursor, "raster_bin: null pointer");
  SIU3R_CHECK(cap > 0 && cap < (1ll << 31), "raster_bin: pair capacity %ld out of range", (long)cap);
  SIU3R_CHECK(G == 0 || (means && cov6 && opacities && mean2d && conic_op && depth && radii && rect && tiles_touched), "raster_bin: null per-Gaussian pointer");
  SIU3R_CHECK(cam->mode == 0 || cam->mode == 1, "raster_bin: bad mode %d", cam->mode);
  SIU3R_CHECK(cam->mode == 1 || G == 0 || (colors && rgb && channels >= (cam->sh_degree + 1) * (cam->sh_degree + 1)), "raster_bin: SH colours missing / too few coefficients");
  Cam c;
  to_cam(cam, &c);
  hipStream_t s = (hipStream_t)stream;
  const int T = ((c.width + TILE - 1) / TILE) * ((c.height + TILE - 1) / TILE);
  if (hipMemsetAsync(tile_count, 0, sizeof(int32_t) * T * NPART, s) != hipSuccess) {
    siu3r_set_error("raster_bin: memset failed");
    return 2;
  }
  if (G > 0)
    hipLaunchKernelGGL(project_kernel, g1(G), dim3(256), 0, s, c, G, means, cov6, opacities, colors, channels, mean2d, conic_op, depth, radii, rect, tiles_touched, rgb, tile_count);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, tile_count, tile_start, cursor, T, (int)cap);
  SIU3R_LAUNCH_CHECK("siu3r_raster_bin");
  return 0;
}

extern "C" int siu3r_raster_sort(const siu3r_raster_cam* cam, int64_t G, const int32_t* rect, const float* depth,
                                 const int32_t* tile_start, int32_t* cursor, uint64_t* keys, int32_t* ids, int64_t cap, void* stream) {
  SIU3R_CHECK(cam && tile_start && cursor && keys && ids && (G == 0 || (rect && depth)), "raster_sort: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int gw = (cam->width + TILE - 1) / TILE, T = gw * ((cam->height + TILE - 1) / TILE);
  if (G > 0) hipLaunchKernelGGL(fill_kernel, g1(G), dim3(256), 0, s, G, rect, depth, cursor, keys, gw, T, (int)cap);
  hipLaunchKernelGGL((sort_kernel<1024, 0>), dim3(T), dim3(256), 0, s, tile_start, keys, ids);
  hipLaunchKernelGGL((sort_kernel<4096, 1024>), dim3(T), dim3(256), 0, s, tile_start, keys, ids);
  hipLaunchKernelGGL((sort_kernel<SORT_CAP, 4096>), dim3(T), dim3(256), 0, s, tile_start, keys, ids);
  SIU3R_LAUNCH_CHECK("siu3r_raster_sort");
  return 0;
}

extern "C" int siu3r_raster_composite_rgb(const siu3r_raster_cam* cam, const int32_t* tile_start, const int32_t* ids,
                                          const float* mean2d, const float* conic_op, const float* depth, const float* rgb,
                                          float* image, float* out_depth, float* out_alpha, int32_t* n_touched, int64_t G,
                                          void* stream) {
  SIU3R_CHECK(cam && tile_start && ids && image && out_depth && out_alpha && (G == 0 || (mean2d && conic_op && depth && rgb && n_touched)), "raster_composite_rgb: null pointer");
  Cam c;
  to_cam(cam, &c);
  hipStream_t s = (hipStream_t)stream;
  const int T = ((c.width + TILE - 1) / TILE) * ((c.height + TILE - 1) / TILE);
  if (G > 0 && hipMemsetAsync(n_touched, 0, sizeof(int32_t) * G, s) != hipSuccess) {
    siu3r_set_error("raster_composite_rgb: memset failed");
    return 2;
  }
  hipLaunchKernelGGL(composite_rgb_kernel, dim3(T), dim3(256), 0, s, c, tile_start, ids, mean2d, conic_op, depth, rgb, image, out_depth, out_alpha, n_touched);
  SIU3R_LAUNCH_CHECK("siu3r_raster_composite_rgb");
  return 0;
}

extern "C" int siu3r_raster_composite_feat(const siu3r_raster_cam* cam, const int32_t* tile_start, const int32_t* ids,
                                           const float* mean2d, const float* conic_op, const float* feats, int channels,
                                           float* out, float* out_alpha, void* stream) {
  SIU3R_CHECK(cam && tile_start && ids && mean2d && conic_op && feats && out && channels > 0, "raster_composite_feat: bad arguments");
  Cam c;
  to_cam(cam, &c);
  const int T = ((c.width + TILE - 1) / TILE) * ((c.height + TILE - 1) / TILE);
  hipLaunchKernelGGL(composite_feat_kernel, dim3(T, (channels + CHUNK - 1) / CHUNK), dim3(256), 0, (hipStream_t)stream, c, tile_start, ids, mean2d, conic_op, feats, channels, out, out_alpha);
  SIU3R_LAUNCH_CHECK("siu3r_raster_composite_feat");
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

extern "C" int siu3r_sh_eval(const float* means, const float* campos3_host, const float* sh, int ncoef, int degree, float* rgb, int64_t G,
                             void* stream) {
  SIU3R_CHECK(G == 0 || (means && campos3_host && sh && rgb), "sh_eval: null pointer");
  SIU3R_CHECK(degree >= 0 && degree <= 4 && ncoef >= (degree + 1) * (degree + 1), "sh_eval: degree %d needs %d coefficients, got %d", degree,
              (degree + 1) * (degree + 1), ncoef);
  if (G == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const float cx = campos3_host[0], cy = campos3_host[1], cz = campos3_host[2];
  switch (degree) {
    case 0: hipLaunchKernelGGL(sh_eval_kernel<0>, g1(G), dim3(256), 0, s, G, ncoef, means, cx, cy, cz, sh, rgb); break;
    case 1: hipLaunchKernelGGL(sh_eval_kernel<1>, g1(G), dim3(256), 0, s, G, ncoef, means, cx, cy, cz, sh, rgb); break;
    case 2: hipLaunchKernelGGL(sh_eval_kernel<2>, g1(G), dim3(256), 0, s, G, ncoef, means, cx, cy, cz, sh, rgb); break;
    case 3: hipLaunchKernelGGL(sh_eval_kernel<3>, g1(G), dim3(256), 0, s, G, ncoef, means, cx, cy, cz, sh, rgb); break;
    default: hipLaunchKernelGGL(sh_eval_kernel<4>, g1(G), dim3(256), 0, s, G, ncoef, means, cx, cy, cz, sh, rgb); break;
  }
  SIU3R_LAUNCH_CHECK("siu3r_sh_eval");
  return 0;
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
