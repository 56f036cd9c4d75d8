#include <hip/hip_runtime.h>
#include <stdint.h>
__device__ __forceinline__ void src_idx(int o, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float src = scale * (o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
}


__device__ __forceinline__ float sample256(const float* p256, int64_t base, int MS, int Q, int q, int y0, int y1, int x0,
                                           int x1, float ly, float lx) {
  const float v00 = p256[(base + (int64_t)y0 * MS + x0) * Q + q], v01 = p256[(base + (int64_t)y0 * MS + x1) * Q + q];
  const float v10 = p256[(base + (int64_t)y1 * MS + x0) * Q + q], v11 = p256[(base + (int64_t)y1 * MS + x1) * Q + q];
#define SIU3R_PIN(x) asm volatile("" : "+v"(x))
  float a = (1.f - lx) * v00; SIU3R_PIN(a);
  float b = lx * v01; SIU3R_PIN(b);
  float c = (1.f - lx) * v10; SIU3R_PIN(c);
  float d = lx * v11; SIU3R_PIN(d);
  float top = a + b; SIU3R_PIN(top);
  float bot = c + d; SIU3R_PIN(bot);
  float g = (1.f - ly) * top; SIU3R_PIN(g);
  float h = ly * bot; SIU3R_PIN(h);
#undef SIU3R_PIN
  return g + h;
}
extern "C" __global__ __launch_bounds__(256) void ppa_fix(const float* p256, const float* scores, const int32_t* kept_idx,
                                                        const int32_t* n_keep, int32_t* lab_map, int32_t* area,
                                                        int32_t* orig, int T, int H, int W, int MS, int Q,
                                                        float mask_thr) {
  const int b = blockIdx.y;
  const int nk = n_keep[b];
  __shared__ int s_area[128], s_orig[128];
  for (int i = threadIdx.x; i < 128; i += 256) s_area[i] = s_orig[i] = 0;
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t npix = (int64_t)T * H * W;
  if (pix < npix && nk > 0) {
    const unsigned row = (unsigned)pix / (unsigned)W;
    const int x = (int)((unsigned)pix - row * (unsigned)W);
    const int t = (int)(row / (unsigned)H);
    const int y = (int)(row - (unsigned)t * (unsigned)H);
    int y0, y1, x0, x1;
    float ly, lx;
    src_idx(y, MS, H, y0, y1, ly);
    src_idx(x, MS, W, x0, x1, lx);
    const int64_t base = ((int64_t)b * T + t) * MS * MS;
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < nk; ++k) {
      const int q = kept_idx[b * Q + k];
      const float wv = sample256(p256, base, MS, Q, q, y0, y1, x0, x1, ly, lx) * scores[b * Q + q];
      if (wv > best) {  // strict: first maximum wins, like torch.argmax
        best = wv;
        bk = k;
      }
      if (wv >= mask_thr) atomicAdd(&s_orig[k], 1);
    }
    lab_map[(int64_t)b * npix + pix] = bk;
    atomicAdd(&s_area[bk], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nk && i < 128; i += 256) {
    if (s_area[i]) atomicAdd(&area[b * Q + i], s_area[i]);
    if (s_orig[i]) atomicAdd(&orig[b * Q + i], s_orig[i]);
  }
}
