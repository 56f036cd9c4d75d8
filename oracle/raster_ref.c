/*
 * raster_ref.c -- CPU oracle for the Gaussian splat rasterizers on the SIU3R path.
 *
 * TEST INFRASTRUCTURE, NOT THE PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  PARITY UNPINNED: the two rasterizers the reference calls are third-party CUDA
 * packages that are NOT in /root/reference and have no CPU path:
 *   K2  diff-gaussian-rasterization-w-pose @ 43e21bff (uv.lock:439-441), called from
 *       src/models/cuda_splatting.py:90-118 (GaussianRasterizer: image, radii, depth, opacity, n_touched)
 *   K3  gsplat 1.5.2 @ 961678f4 (uv.lock:757-759), called from src/models/gaussian_renderer.py:92-106
 *       (rasterization(): N-channel colours + alphas)
 * so this file restates their PUBLISHED algorithms (3DGS forward: project / EWA / 0.3 px dilation / 3-sigma
 * extent / tile binning / (tile, depth) sort / front-to-back alpha blend; thresholds 1/255, 0.99 resp. 0.999,
 * 1e-4) with every constant exposed as a named parameter, and is anchored on the
 * reference's own call sites for argument conventions (row-vector matrices, 6-float upper-triangular
 * covariance, SH [G,25,3], x10 scene scale, near = 1).  What it pins for the HIP renderer: tiles_touched,
 * radii, per-tile sorted Gaussian lists, n_touched (integers, bit-exact) and the rendered maps (fp32 tolerance).
 *
 * Plain C99, single thread by default (OpenMP over tiles when compiled with -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

typedef struct {
  int32_t mode;        /* 0 = K2 (3DGS family), 1 = K3 (gsplat family) */
  int32_t width, height;
  float w2c[16];       /* world->camera, row-major 4x4 (column-vector convention) */
  /* K2 */
  float proj[16];      /* full projection P = Proj * W2C, row-major (p_hom = P * p) */
  float tanfovx, tanfovy;
  float campos[3];
  float bg[3];
  int32_t sh_degree;   /* 0..4; band 4 (coefficients 16..24) only evaluated when sh_band4 != 0; -1: colors = precomputed [G,3] */
  int32_t sh_band4;
  float k2_znear_cull; /* 0.2: p_view.z <= this is culled (the constant inside the CUDA kernel) */
  /* K3 */
  float fx, fy, cx, cy; /* pixel units */
  float near_plane, far_plane;
  float eps2d;         /* 0.3 */
  float radius_clip;   /* 0 in the pipeline, 0.1 in the viewer */
  float extent_sigma;  /* 3.33 */
  int32_t opacity_aware_extent; /* tighten the extent with log(opacity/threshold) */
  /* shared blend constants */
  float alpha_min;     /* 1/255 */
  float alpha_max;     /* 0.99 (K2) / 0.999 (K3) */
  float t_min;         /* 1e-4 */
  float dilation;      /* 0.3 (K2 low-pass) */
  int32_t nt_post_blend; /* n_touched counts a pixel when the transmittance AFTER (1: `test_T > 0.5f`, MonoGS fork) or BEFORE (0) the blend is > 0.5 */
  float k2_near, k2_far; /* (device-side pose preparation of the product only: unused here) */
} raster_cam;

typedef struct {
  float mx, my;        /* pixel-space mean */
  float ca, cb, cc;    /* conic */
  float opacity;
  float depth;
  int32_t radius_x, radius_y;
  int32_t tx0, ty0, tx1, ty1; /* tile rect [min, max) */
  int32_t valid;
} proj_g;

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
static const float SH_C4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f, 0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f, 0.6258357354491761f};

/* SH -> RGB for one Gaussian; sh is [n_coef][3] (the reference rearranges to 'g n xyz', cuda_splatting.py:65) */
static void sh_to_rgb(const raster_cam* c, const float* mean, const float* sh, float* rgb) {
  float dx = mean[0] - c->campos[0], dy = mean[1] - c->campos[1], dz = mean[2] - c->campos[2];
  float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  float x = dx * inv, y = dy * inv, z = dz * inv;
  int deg = c->sh_degree;
  if (deg < 0) { /* `colors_precomp` of the CUDA package (cuda_splatting.py:112, use_sh = False): blended as given */
    rgb[0] = sh[0];
    rgb[1] = sh[1];
    rgb[2] = sh[2];
    return;
  }
  for (int ch = 0; ch < 3; ++ch) {
#define S(i) sh[(i) * 3 + ch]
    float r = SH_C0 * S(0);
    if (deg > 0) {
      r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) + SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
        if (deg > 2) {
          r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) + SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) + SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) +
              SH_C3[5] * z * (xx - yy) * S(14) + SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
          if (deg > 3 && c->sh_band4) {
            r = r + SH_C4[0] * xy * (xx - yy) * S(16) + SH_C4[1] * yz * (3.0f * xx - yy) * S(17) + SH_C4[2] * xy * (7.0f * zz - 1.0f) * S(18) +
                SH_C4[3] * yz * (7.0f * zz - 3.0f) * S(19) + SH_C4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f) * S(20) +
                SH_C4[5] * xz * (7.0f * zz - 3.0f) * S(21) + SH_C4[6] * (xx - yy) * (7.0f * zz - 1.0f) * S(22) +
                SH_C4[7] * xz * (xx - 3.0f * yy) * S(23) + SH_C4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)) * S(24);
          }
        }
      }
    }
#undef S
    r += 0.5f;
    rgb[ch] = r < 0.0f ? 0.0f : r;
  }
}

/* exp(x) for x <= 0 from correctly rounded fp32 operations only (fmaf: one rounding, like the GPU's v_fma_f32; the HIP renderer
 * uses the very same sequence, so alpha, the transmittance chain and hence n_touched are bit-identical on both sides):
 * 2^(x*log2e), degree-7 Taylor of 2^f as a Horner chain of fused multiply-adds. */
static float exp_det(float x) {
  if (x < -87.0f) return 0.0f;
  const float y = x * 1.4426950408889634f;
  const float n = floorf(y + 0.5f);
  const float f = y - n;
  float p = 1.52527338e-5f;
  p = fmaf(p, f, 1.54035304e-4f);
  p = fmaf(p, f, 1.33335581e-3f);
  p = fmaf(p, f, 9.61812911e-3f);
  p = fmaf(p, f, 5.55041087e-2f);
  p = fmaf(p, f, 2.40226507e-1f);
  p = fmaf(p, f, 6.93147181e-1f);
  p = fmaf(p, f, 1.0f);
  return ldexpf(p, (int)n);
}
/* q = 0.5 (a dx^2 + c dy^2) + b dx dy in the fused order shared with the HIP renderer */
static float conic_sigma(float ca, float cb, float cc, float dx, float dy) {
  const float q = fmaf(cc * dy, dy, (ca * dx) * dx);
  return fmaf(cb * dx, dy, 0.5f * q);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* project one Gaussian; cov6 = (xx, xy, xz, yy, yz, zz) (cuda_splatting.py:107,115) */
static void project_one(const raster_cam* c, const float* mean, const float* cov6, float opacity, proj_g* o) {
  const float* V = c->w2c;
  memset(o, 0, sizeof(*o));
  float tx = V[0] * mean[0] + V[1] * mean[1] + V[2] * mean[2] + V[3];
  float ty = V[4] * mean[0] + V[5] * mean[1] + V[6] * mean[2] + V[7];
  float tz = V[8] * mean[0] + V[9] * mean[1] + V[10] * mean[2] + V[11];
  const int gw = (c->width + TILE - 1) / TILE, gh = (c->height + TILE - 1) / TILE;
  float fx, fy;
  if (c->mode == 0) {
    if (tz <= c->k2_znear_cull) return;
    fx = c->width / (2.0f * c->tanfovx);
    fy = c->height / (2.0f * c->tanfovy);
  } else {
    if (tz < c->near_plane || tz > c->far_plane) return;
    fx = c->fx;
    fy = c->fy;
  }
  /* clamp the view-space direction before linearising (EWA) */
  float limx_pos, limx_neg, limy_pos, limy_neg;
  if (c->mode == 0) {
    limx_pos = limx_neg = 1.3f * c->tanfovx;
    limy_pos = limy_neg = 1.3f * c->tanfovy;
  } else {
    float tfx = 0.5f * c->width / fx, tfy = 0.5f * c->height / fy;
    limx_pos = (c->width - c->cx) / fx + 0.3f * tfx;
    limx_neg = c->cx / fx + 0.3f * tfx;
    limy_pos = (c->height - c->cy) / fy + 0.3f * tfy;
    limy_neg = c->cy / fy + 0.3f * tfy;
  }
  float rz = 1.0f / tz;
  float txz = tx * rz, tyz = ty * rz;
  float cxz = fminf(limx_pos, fmaxf(-limx_neg, txz)), cyz = fminf(limy_pos, fmaxf(-limy_neg, tyz));
  float ctx = cxz * tz, cty = cyz * tz;
  /* M = J * R (2x3), J = [[fx/tz, 0, -fx*tx/tz^2], [0, fy/tz, -fy*ty/tz^2]] */
  float j00 = fx * rz, j02 = -(fx * ctx) * rz * rz, j11 = fy * rz, j12 = -(fy * cty) * rz * rz;
  float m00 = j00 * V[0] + j02 * V[8], m01 = j00 * V[1] + j02 * V[9], m02 = j00 * V[2] + j02 * V[10];
  float m10 = j11 * V[4] + j12 * V[8], m11 = j11 * V[5] + j12 * V[9], m12 = j11 * V[6] + j12 * V[10];
  float sxx = cov6[0], sxy = cov6[1], sxz = cov6[2], syy = cov6[3], syz = cov6[4], szz = cov6[5];
  /* cov2d = M * Sigma * M^T */
  float a0 = m00 * sxx + m01 * sxy + m02 * sxz, a1 = m00 * sxy + m01 * syy + m02 * syz, a2 = m00 * sxz + m01 * syz + m02 * szz;
  float b0 = m10 * sxx + m11 * sxy + m12 * sxz, b1 = m10 * sxy + m11 * syy + m12 * syz, b2 = m10 * sxz + m11 * syz + m12 * szz;
  float c00 = a0 * m00 + a1 * m01 + a2 * m02;
  float c01 = a0 * m10 + a1 * m11 + a2 * m12;
  float c11 = b0 * m10 + b1 * m11 + b2 * m12;
  float blur = c->mode == 0 ? c->dilation : c->eps2d;
  c00 += blur;
  c11 += blur;
  float det = c00 * c11 - c01 * c01;
  if (c->mode == 0 ? (det == 0.0f) : (det <= 0.0f)) return;
  float det_inv = 1.0f / det;
  o->ca = c11 * det_inv;
  o->cb = -c01 * det_inv;
  o->cc = c00 * det_inv;
  o->opacity = opacity;
  o->depth = tz;
  if (c->mode == 0) {
    const float* P = c->proj;
    float hx = P[0] * mean[0] + P[1] * mean[1] + P[2] * mean[2] + P[3];
    float hy = P[4] * mean[0] + P[5] * mean[1] + P[6] * mean[2] + P[7];
    float hw = P[12] * mean[0] + P[13] * mean[1] + P[14] * mean[2] + P[15];
    float pw = 1.0f / (hw + 0.0000001f);
    o->mx = ((hx * pw + 1.0f) * c->width - 1.0f) * 0.5f;  /* ndc2Pix: pixel centres at integers */
    o->my = ((hy * pw + 1.0f) * c->height - 1.0f) * 0.5f;
    float mid = 0.5f * (c00 + c11);
    float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    float lam2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    int rad = (int)ceilf(3.0f * sqrtf(fmaxf(lam, lam2)));
    o->radius_x = o->radius_y = rad;
    o->tx0 = clampi((int)((o->mx - rad) / TILE), 0, gw);
    o->ty0 = clampi((int)((o->my - rad) / TILE), 0, gh);
    o->tx1 = clampi((int)((o->mx + rad + TILE - 1) / TILE), 0, gw);
    o->ty1 = clampi((int)((o->my + rad + TILE - 1) / TILE), 0, gh);
  } else {
    o->mx = fx * txz + c->cx;  /* pixel centres at +0.5 */
    o->my = fy * tyz + c->cy;
    float extend = c->extent_sigma;
    if (c->opacity_aware_extent) {
      if (opacity < c->alpha_min) return;
      extend = fminf(extend, sqrtf(2.0f * logf(opacity / c->alpha_min)));
    }
    float rx = ceilf(extend * sqrtf(c00)), ry = ceilf(extend * sqrtf(c11));
    if (rx <= c->radius_clip && ry <= c->radius_clip) return;
    if (o->mx + rx <= 0 || o->mx - rx >= c->width || o->my + ry <= 0 || o->my - ry >= c->height) return;
    o->radius_x = (int)rx;
    o->radius_y = (int)ry;
    o->tx0 = clampi((int)floorf((o->mx - rx) / TILE), 0, gw);
    o->ty0 = clampi((int)floorf((o->my - ry) / TILE), 0, gh);
    o->tx1 = clampi((int)ceilf((o->mx + rx) / TILE), 0, gw);
    o->ty1 = clampi((int)ceilf((o->my + ry) / TILE), 0, gh);
  }
  if ((o->tx1 - o->tx0) * (o->ty1 - o->ty0) == 0) {
    if (c->mode == 0) o->radius_x = o->radius_y = 0;  /* the CUDA kernel returns before writing radii */
    return;
  }
  o->valid = 1;
}

typedef struct {
  uint32_t depth_bits;
  int32_t id;
} kv;
static int kv_cmp(const void* a, const void* b) {
  const kv *x = (const kv*)a, *y = (const kv*)b;
  if (x->depth_bits != y->depth_bits) return x->depth_bits < y->depth_bits ? -1 : 1;
  return x->id < y->id ? -1 : (x->id > y->id);
}

/*
 * Full forward.  means [G,3]; cov6 [G,6]; opacities [G]; colors: mode 0 -> SH [G, ncoef, 3] (ncoef = channels),
 * mode 1 -> features [G, channels].  Outputs (any may be NULL): radii [G,2] i32 (x,y), tiles_touched [G] i32,
 * n_touched [G] i32 (K2), image: mode 0 [3,H,W], mode 1 [H,W,channels]; depth [H,W] (K2), alpha [H,W]
 * (K2: accumulated opacity, K3: alphas), tile_start [T+1] i32, sorted_ids (capacity ids_cap) i32.
 * Returns D (number of tile-Gaussian pairs) or -1 if ids_cap is too small.
 */
int64_t raster_ref_forward(const raster_cam* c, int64_t G, const float* means, const float* cov6, const float* opacities,
                           const float* colors, int32_t channels, int32_t* radii, int32_t* tiles_touched, int32_t* n_touched,
                           float* image, float* depth, float* alpha, int32_t* tile_start, int32_t* sorted_ids, int64_t ids_cap) {
  const int W = c->width, H = c->height;
  const int gw = (W + TILE - 1) / TILE, gh = (H + TILE - 1) / TILE, T = gw * gh;
  proj_g* pg = (proj_g*)malloc(sizeof(proj_g) * (size_t)(G > 0 ? G : 1));
  float* rgb = c->mode == 0 ? (float*)malloc(sizeof(float) * 3 * (size_t)(G > 0 ? G : 1)) : NULL;
  int32_t* cnt = (int32_t*)calloc((size_t)T + 1, sizeof(int32_t));
  for (int64_t g = 0; g < G; ++g) {
    project_one(c, means + 3 * g, cov6 + 6 * g, opacities[g], &pg[g]);
    if (radii) {
      radii[2 * g] = pg[g].radius_x;
      radii[2 * g + 1] = pg[g].radius_y;
    }
    int tt = pg[g].valid ? (pg[g].tx1 - pg[g].tx0) * (pg[g].ty1 - pg[g].ty0) : 0;
    if (tiles_touched) tiles_touched[g] = tt;
    if (n_touched) n_touched[g] = 0;
    if (!pg[g].valid) continue;
    if (c->mode == 0) sh_to_rgb(c, means + 3 * g, colors + (size_t)g * channels * 3, rgb + 3 * g);
    for (int ty = pg[g].ty0; ty < pg[g].ty1; ++ty)
      for (int tx = pg[g].tx0; tx < pg[g].tx1; ++tx) cnt[ty * gw + tx + 1]++;
  }
  for (int t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
  const int64_t D = cnt[T];
  kv* list = (kv*)malloc(sizeof(kv) * (size_t)(D > 0 ? D : 1));
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)T);
  memcpy(cur, cnt, sizeof(int32_t) * (size_t)T);
  for (int64_t g = 0; g < G; ++g) {
    if (!pg[g].valid) continue;
    uint32_t db;
    memcpy(&db, &pg[g].depth, 4);
    for (int ty = pg[g].ty0; ty < pg[g].ty1; ++ty)
      for (int tx = pg[g].tx0; tx < pg[g].tx1; ++tx) {
        kv* e = &list[cur[ty * gw + tx]++];
        e->depth_bits = db;  /* positive floats order like their bit patterns (key = tile | depth bits) */
        e->id = (int32_t)g;
      }
  }
  for (int t = 0; t < T; ++t) qsort(list + cnt[t], (size_t)(cnt[t + 1] - cnt[t]), sizeof(kv), kv_cmp);  /* ties: Gaussian index */
  if (tile_start) memcpy(tile_start, cnt, sizeof(int32_t) * ((size_t)T + 1));
  int64_t ret = D;
  if (sorted_ids) {
    if (D > ids_cap) ret = -1;
    else for (int64_t i = 0; i < D; ++i) sorted_ids[i] = list[i].id;
  }
  const int C = c->mode == 0 ? 3 : channels;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int t = 0; t < T; ++t) {
    const int tx = t % gw, ty = t / gw;
    float* acc = (float*)malloc(sizeof(float) * (size_t)C);
    for (int py = ty * TILE; py < (ty + 1) * TILE && py < H; ++py)
      for (int px = tx * TILE; px < (tx + 1) * TILE && px < W; ++px) {
        const float pxf = c->mode == 0 ? (float)px : (float)px + 0.5f, pyf = c->mode == 0 ? (float)py : (float)py + 0.5f;
        float Tr = 1.0f, dacc = 0.0f, oacc = 0.0f;
        for (int k = 0; k < C; ++k) acc[k] = 0.0f;
        for (int i = cnt[t]; i < cnt[t + 1]; ++i) {
          const int g = list[i].id;
          const proj_g* q = &pg[g];
          const float dx = q->mx - pxf, dy = q->my - pyf;
          float a;
          if (c->mode == 0) {
            const float power = -conic_sigma(q->ca, q->cb, q->cc, dx, dy);
            if (power > 0.0f) continue;
            a = fminf(c->alpha_max, q->opacity * exp_det(power));
          } else {
            const float sigma = conic_sigma(q->ca, q->cb, q->cc, dx, dy);
            if (sigma < 0.0f) continue;
            a = fminf(c->alpha_max, q->opacity * exp_det(-sigma));
          }
          if (a < c->alpha_min) continue;
          const float nT = fmaf(-Tr, a, Tr); /* T (1 - a) */
          if (c->mode == 0 ? (nT < c->t_min) : (nT <= c->t_min)) break;
          const float wgt = a * Tr;
          if (c->mode == 0) {
            for (int k = 0; k < 3; ++k) acc[k] = fmaf(rgb[3 * g + k], wgt, acc[k]);
            dacc = fmaf(q->depth, wgt, dacc);
            oacc += wgt;
            if (n_touched && (c->nt_post_blend ? nT : Tr) > 0.5f) {
#ifdef _OPENMP
#pragma omp atomic
#endif
              n_touched[g]++;
            }
          } else {
            const float* f = colors + (size_t)g * channels;
            for (int k = 0; k < C; ++k) acc[k] = fmaf(f[k], wgt, acc[k]);
            oacc += wgt;
          }
          Tr = nT;
        }
        const size_t pix = (size_t)py * W + px;
        if (image) {
          if (c->mode == 0)
            for (int k = 0; k < 3; ++k) image[(size_t)k * H * W + pix] = fmaf(Tr, c->bg[k], acc[k]);
          else
            for (int k = 0; k < C; ++k) image[pix * C + k] = acc[k];
        }
        if (depth) depth[pix] = dacc;
        if (alpha) alpha[pix] = oacc;
      }
    free(acc);
  }
  free(pg);
  free(rgb);
  free(cnt);
  free(list);
  free(cur);
  return ret;
}

int raster_ref_struct_size(void) { return (int)sizeof(raster_cam); }

/* exported for the SH-basis consistency test */
void raster_ref_sh_to_rgb(const raster_cam* c, const float* mean, const float* sh, float* rgb) { sh_to_rgb(c, mean, sh, rgb); }

/* ---- viewer-semantics helpers (reference viewer.py:301-336 -> gsplat.rasterization with quats/scales/SH colours; gsplat@961678f4 is not
 * in /root/reference: published algorithm restated, PARITY UNPINNED like the rest of this file) ------------------------------------- */

/* gsplat quat_scale_to_covar: q = (w,x,y,z) normalised, M = R diag(s), Sigma = M M^T, returned as the 6 upper-triangular entries */
void raster_ref_quat_scale_to_cov6(int64_t G, const float* quats, const float* scales, float* cov6) {
  for (int64_t g = 0; g < G; ++g) {
    float w = quats[4 * g], x = quats[4 * g + 1], y = quats[4 * g + 2], z = quats[4 * g + 3];
    const float inv = 1.0f / sqrtf(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    const float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    const float R[9] = {1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy), 2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx),
                        2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2)};
    const float s0 = scales[3 * g], s1 = scales[3 * g + 1], s2 = scales[3 * g + 2];
    float M[9];
    for (int r = 0; r < 3; ++r) { M[3 * r] = R[3 * r] * s0; M[3 * r + 1] = R[3 * r + 1] * s1; M[3 * r + 2] = R[3 * r + 2] * s2; }
    int o = 0;
    for (int r = 0; r < 3; ++r)
      for (int c = r; c < 3; ++c) cov6[6 * g + o++] = M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1] + M[3 * r + 2] * M[3 * c + 2];
  }
}

/* gsplat spherical_harmonics (sh_coeffs_to_color_fast, Sloan's recurrences) + rasterization()'s clamp_min(c + 0.5, 0);
 * sh [G, ncoef, 3], dirs = mean - campos (normalised here), degree <= 4 */
void raster_ref_sh_eval(int64_t G, int degree, int ncoef, const float* means, const float* campos, const float* sh, float* rgb) {
  for (int64_t g = 0; g < G; ++g) {
    const float dx = means[3 * g] - campos[0], dy = means[3 * g + 1] - campos[1], dz = means[3 * g + 2] - campos[2];
    const float inorm = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
    float b[25];
    b[0] = 0.2820947917738781f;
    if (degree >= 1) { b[1] = -0.48860251190292f * y; b[2] = 0.48860251190292f * z; b[3] = -0.48860251190292f * x; }
    float z2 = 0, fC1 = 0, fS1 = 0, fC2 = 0, fS2 = 0;
    if (degree >= 2) {
      z2 = z * z;
      const float fTmp0B = -1.092548430592079f * z;
      fC1 = x * x - y * y;
      fS1 = 2.0f * x * y;
      b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f; b[7] = fTmp0B * x; b[5] = fTmp0B * y; b[8] = 0.5462742152960395f * fC1; b[4] = 0.5462742152960395f * fS1;
    }
    if (degree >= 3) {
      const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp1B = 1.445305721320277f * z;
      fC2 = x * fC1 - y * fS1;
      fS2 = x * fS1 + y * fC1;
      b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f); b[13] = fTmp0C * x; b[11] = fTmp0C * y; b[14] = fTmp1B * fC1; b[10] = fTmp1B * fS1;
      b[15] = -0.5900435899266435f * fC2; b[9] = -0.5900435899266435f * fS2;
    }
    if (degree >= 4) {
      const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f), fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp2B = -1.770130769779931f * z;
      const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
      b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6]; b[21] = fTmp0D * x; b[19] = fTmp0D * y; b[22] = fTmp1C * fC1; b[18] = fTmp1C * fS1;
      b[23] = fTmp2B * fC2; b[17] = fTmp2B * fS2; b[24] = 0.6258357354491763f * fC3; b[16] = 0.6258357354491763f * fS3;
    }
    const int nb = (degree + 1) * (degree + 1);
    for (int ch = 0; ch < 3; ++ch) {
      float r = 0.0f;
      for (int i = 0; i < nb; ++i) r = r + b[i] * sh[((size_t)g * ncoef + i) * 3 + ch];
      r += 0.5f;
      rgb[3 * g + ch] = r < 0.0f ? 0.0f : r;
    }
  }
}

/* colors[P, C] += (1 - alpha[P]) * bg[C]  (gsplat `backgrounds`) */
void raster_ref_blend_background(int64_t P, int C, float* colors, const float* alpha, const float* bg) {
  for (int64_t p = 0; p < P; ++p)
    for (int c = 0; c < C; ++c) colors[p * C + c] = colors[p * C + c] + (1.0f - alpha[p]) * bg[c];
}
