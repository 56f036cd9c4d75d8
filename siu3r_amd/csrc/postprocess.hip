// Panoptic post-processing on device (integer outputs), restating
// VideoMask2FormerImageProcessor.post_process_panoptic_segmentation
// (reference src/models/mask2former/image_processing_video_mask2former.py:1238-1481) and the label scatter
// of SIU3RModel.post_process_gaussians (reference src/models/model.py:267-294).
//
// Pipeline (no host sync until the segment table is read back):
//   pp_class   : softmax over classes, (score,label)=max, keep = label != void && score > thr, compaction
//   pp_mask256 : mask logits [B,T,h,w,Q] -> bilinear (256,256) -> sigmoid            (the hard-coded mask_size)
//   pp_argmax  : bilinear (256,256)->(H,W) of the kept queries' probabilities, argmax_k(prob*score),
//                area_k = #(argmax==k), orig_k = #(prob*score >= 0.5)
//   pp_accept  : sequential acceptance (area/orig > 0.8), segment ids, stuff fusing       (1 thread / item)
//   pp_write   : segmentation / semantic / instance maps from the per-query table
//   pp_qcl     : query_class_logits[(t,y,x), j, c] = class_prob[k_j, c] * mask_prob[t, k_j, y, x]
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ void src_idx(int o, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = (float)in / (float)out;
  float src = scale * (o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - i0;
}

// one block (64 threads) per batch item; thread q handles query q (Q <= 1024 via loop)
__global__ void pp_class_kernel(const float* logits, float* probs, float* scores, int32_t* labels, int32_t* kept_idx,
                                int32_t* n_keep, int Q, int C, float thr, int32_t* area, int32_t* orig) {
  const int b = blockIdx.x;
  __shared__ int keep_flag[1024];
  // the pixel counters of pp_argmax_kernel start at zero: cleared here (no memset node between the kernels of the stage)
  for (int q = threadIdx.x; q < Q; q += blockDim.x) area[b * Q + q] = orig[b * Q + q] = 0;
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    const float* l = logits + ((int64_t)b * Q + q) * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, l[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(l[c] - mx);
    float best = -1.f;
    int bi = 0;
    for (int c = 0; c < C; ++c) {
      const float pr = expf(l[c] - mx) / den;
      probs[((int64_t)b * Q + q) * C + c] = pr;
      if (pr > best) {
        best = pr;
        bi = c;
      }
    }
    scores[b * Q + q] = best;
    labels[b * Q + q] = bi;
    keep_flag[q] = (bi != C - 1) && (best > thr);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int q = 0; q < Q; ++q)
      if (keep_flag[q]) kept_idx[b * Q + n++] = q;
    n_keep[b] = n;
  }
}

// The 256 x 256 probability planes of the KEPT queries only, query-major: out[(b, t), k, oy, ox] for k < n_keep[b] (round 6; rounds 2-5
// wrote all Q queries channel-last: the argmax then gathered one float per 400-byte pixel record -- 64 cache lines per load instruction
// -- and three quarters of the volume were never read).  A plane is contiguous: the argmax's and the logit volume's bilinear corners
// of neighbouring pixels sit next to each other.  Same arithmetic per value as before.
// grid: (ceil(OS * OS / 256), Q, B * T)
__global__ __launch_bounds__(256) void pp_mask256_kernel(const float* ml, const int32_t* kept_idx, const int32_t* n_keep, float* out, int T, int IH, int IW, int OS, int Q) {
  const int bt = blockIdx.z, b = bt / T, k = blockIdx.y;
  if (k >= n_keep[b]) return;
  const int q = kept_idx[b * Q + k];
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= (unsigned)(OS * OS)) return;
  const int oy = (int)(pix / (unsigned)OS), ox = (int)(pix - (unsigned)oy * (unsigned)OS);
  int y0, y1, x0, x1;
  float ly, lx;
  src_idx(oy, IH, OS, y0, y1, ly);
  src_idx(ox, IW, OS, x0, x1, lx);
  const int64_t base = (int64_t)bt * IH * IW;
  const float v00 = ml[(base + (int64_t)y0 * IW + x0) * Q + q], v01 = ml[(base + (int64_t)y0 * IW + x1) * Q + q];
  const float v10 = ml[(base + (int64_t)y1 * IW + x0) * Q + q], v11 = ml[(base + (int64_t)y1 * IW + x1) * Q + q];
  const float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  out[((int64_t)bt * Q + k) * OS * OS + pix] = 1.f / (1.f + expf(-v));
}

// Bilinear sample of the 256^2 probability volume: PLAIN loads, and arithmetic the compiler cannot pack.
//
// Round 4 found the argmax kernel giving ~100 border pixels of a B = 8 batch's first item to a neighbouring segment in one forward of
// four, with bit-identical inputs; it looked like a stale read of the volume and was worked around with sc0 sc1 loads.  Round 6 took the
// round-4 code object apart (tools/pk_hazard_probe.py, tools/probes/pk_hazard/; profiles/r06_pk_hazard.txt; DESIGN.md section 5, round
// 6, item 9).  Nothing is stale: the volume is 470 us old when the argmax starts, a NaN-filled or persistent volume changes nothing, a
// second and third launch of the same kernel milliseconds later are wrong as often as the first.  The SLP vectoriser had turned
// (a + b, c + d) of the lerp into  v_pk_add_f32 v[18:19], v[18:19], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]  -- and on gfx950 a packed fp32
// instruction whose LOW result selects the HIGH half of its second source (op_sel:[x,1]) returns a wrong low half in 0.1-0.3 % of its
// executions while a bf16 MFMA of ANOTHER wave runs on the same SIMD (tools/probes/pk_hazard/xwave2.hip: 90 lines, no library code;
// every other selection is right, an f32 MFMA neighbour is harmless).  The argmax's first workgroups find room on CUs that run the
// other streams' two-per-CU bf16x3 GEMM workgroups; no wait count, idle cycle or register choice of the victim closes a hazard between
// waves (one-edit variants of the compiled assembly: only dropping the crossed packed add does).  The library is built without the SLP
// vectoriser (siu3r_amd/build.py; tools/scan_pk_opsel_hazard.py finds no such instruction in the shipped objects), and the form below
// leaves nothing to pack in any case: every product and sum pinned in its own register by an empty asm (same operations in the same
// order: identical bits); tests/test_abi.py checks both.
// plane: the MS x MS probabilities of one (item, view, kept query)
__device__ __forceinline__ float sample256(const float* plane, int MS, int y0, int y1, int x0, int x1, float ly, float lx) {
  const float v00 = plane[y0 * MS + x0], v01 = plane[y0 * MS + x1];
  const float v10 = plane[y1 * MS + x0], v11 = plane[y1 * MS + x1];
#ifdef SIU3R_AB_PACKED_LERP  // A/B builds only (tests/test_postprocess_gpu.py documents how): the rounds 3-5 form, which the SLP vectoriser packs
  return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
#endif
#define SIU3R_PIN(x) asm volatile("" : "+v"(x))
  float a = (1.f - lx) * v00;
  SIU3R_PIN(a);
  float b = lx * v01;
  SIU3R_PIN(b);
  float c = (1.f - lx) * v10;
  SIU3R_PIN(c);
  float d = lx * v11;
  SIU3R_PIN(d);
  float top = a + b;
  SIU3R_PIN(top);
  float bot = c + d;
  SIU3R_PIN(bot);
  float g = (1.f - ly) * top;
  SIU3R_PIN(g);
  float h = ly * bot;
  SIU3R_PIN(h);
#undef SIU3R_PIN
  return g + h;
}

// grid: (ceil(T*H*W/256), B)
__global__ __launch_bounds__(256) void pp_argmax_kernel(const float* p256, const float* scores, const int32_t* kept_idx,
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
    const int64_t base = ((int64_t)b * T + t) * Q;  // first plane of this (item, view)
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < nk; ++k) {
      const int q = kept_idx[b * Q + k];
      const float wv = sample256(p256 + (base + k) * MS * MS, MS, y0, y1, x0, x1, ly, lx) * scores[b * Q + q];
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

// table per (b, k): seg_id (0 = rejected), label, fused flag, score ; n_acc[b]; acc_list[b, j] = k of j-th accepted
__global__ void pp_accept_kernel(const int32_t* area, const int32_t* orig, const int32_t* kept_idx, const int32_t* n_keep,
                                 const int32_t* labels, const float* scores, int32_t* seg_id, int32_t* seg_label,
                                 int32_t* seg_fused, float* seg_score, int32_t* acc_list, int32_t* n_acc, int Q,
                                 float overlap, uint32_t fuse_mask, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nk = n_keep[b];
  int cur = 0, na = 0;
  int stuff_mem[32];
  for (int i = 0; i < 32; ++i) stuff_mem[i] = 0;
  for (int k = 0; k < Q; ++k) seg_id[b * Q + k] = 0;
  for (int k = 0; k < nk; ++k) {
    const int q = kept_idx[b * Q + k];
    const int cls = labels[b * Q + q];
    const bool fuse = cls < 32 && ((fuse_mask >> cls) & 1u);
    const int a = area[b * Q + k], o = orig[b * Q + k];
    bool exists = a > 0 && o > 0;
    if (exists) {
      const float ratio = (float)a / (float)o;
      if (!(ratio > overlap)) exists = false;
    }
    if (!exists) continue;
    int sid_f;
    if (cls < 32 && stuff_mem[cls] != 0) {
      sid_f = stuff_mem[cls];
    } else {
      cur += 1;
      sid_f = cur;
    }
    const int sid = fuse ? sid_f : cur;
    seg_id[b * Q + k] = sid;
    seg_label[b * Q + k] = cls;
    seg_fused[b * Q + k] = fuse ? 1 : 0;
    seg_score[b * Q + k] = scores[b * Q + q];
    acc_list[b * Q + na++] = k;
    if (fuse && cls < 32 && stuff_mem[cls] == 0) stuff_mem[cls] = cur;
  }
  n_acc[b] = na;
}

__global__ void pp_write_kernel(const int32_t* lab_map, const int32_t* seg_id, const int32_t* seg_label,
                                const int32_t* n_keep, int32_t* seg, int32_t* sem, int32_t* ins, int64_t npix, int Q) {
  const int b = blockIdx.y;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  int s = 0, se = 0, in = 0;
  if (n_keep[b] > 0) {
    const int k = lab_map[(int64_t)b * npix + pix];
    const int id = seg_id[b * Q + k];
    if (id > 0) {
      s = id;
      se = seg_label[b * Q + k] + 1;  // +1: 0 is background (model.py:282-284)
      in = id;
    }
  }
  seg[(int64_t)b * npix + pix] = s;
  sem[(int64_t)b * npix + pix] = se;
  ins[(int64_t)b * npix + pix] = in;
}

// out [(T*H*W), nq, C] for ONE batch item; acc = device list of accepted k (indices into kept_idx)
__global__ __launch_bounds__(256) void pp_qcl_kernel(const float* p256, const float* probs, const int32_t* kept_idx, const int32_t* acc, int nq,
                                                     float* out, int b, int T, int H, int W, int MS, int Q, int C) {
  // A workgroup owns 256 consecutive (pixel, accepted query) pairs = one contiguous run of 256 x C output floats.  Phase 1: one thread
  // per pair does the index arithmetic and the bilinear mask sample once.  Phase 2: all threads write the run, consecutive lanes to
  // consecutive floats.  (A thread per pair writing its own 21 floats left every store strided: 380 GB/s on the 176 MB volume; a thread
  // per element redid the 64-bit index arithmetic per float.)
  __shared__ float s_mp[256];
  __shared__ int s_q[256];
  const unsigned npairs = (unsigned)((int64_t)T * H * W * nq);
  const unsigned pair0 = blockIdx.x * 256u, pair = pair0 + threadIdx.x;
  float mp = 0.f;
  int q = 0;
  if (pair < npairs) {
    const unsigned pix = pair / (unsigned)nq;
    const int j = (int)(pair - pix * (unsigned)nq);
    const unsigned row = pix / (unsigned)W;
    const int x = (int)(pix - row * (unsigned)W);
    const unsigned t = row / (unsigned)H;
    const int y = (int)(row - t * (unsigned)H);
    int y0, y1, x0, x1;
    float ly, lx;
    src_idx(y, MS, H, y0, y1, ly);
    src_idx(x, MS, W, x0, x1, lx);
    const int k = acc[b * Q + j];
    q = kept_idx[b * Q + k];
    mp = sample256(p256 + (((int64_t)b * T + t) * Q + k) * MS * MS, MS, y0, y1, x0, x1, ly, lx);
  }
  s_mp[threadIdx.x] = mp;
  s_q[threadIdx.x] = q;
  __syncthreads();
  const unsigned nloc = min(256u, npairs - pair0) * (unsigned)C;
  float* o = out + (size_t)pair0 * C;
  const float* pb = probs + (int64_t)b * Q * C;
  for (unsigned e = threadIdx.x; e < nloc; e += 256u) {
    const unsigned pl = e / (unsigned)C, c = e - pl * (unsigned)C;
    o[e] = pb[s_q[pl] * C + c] * s_mp[pl];
  }
}

inline dim3 g1(int64_t n, int blk = 256) { return dim3((unsigned)cdiv64(n, blk)); }

}  // namespace

extern "C" int siu3r_panoptic_stage1(const float* class_logits, const float* mask_logits_cl, float* probs, float* scores,
                                     int32_t* labels, int32_t* kept_idx, int32_t* n_keep, float* p256,
                                     int32_t* lab_map, int32_t* area, int32_t* orig, int32_t* seg_id, int32_t* seg_label,
                                     int32_t* seg_fused, float* seg_score, int32_t* acc_list, int32_t* n_acc,
                                     int32_t* seg, int32_t* sem, int32_t* ins, int B, int T, int Q, int C, int IH,
                                     int IW, int H, int W, int mask_size, float threshold, float mask_threshold,
                                     float overlap, uint32_t fuse_mask, void* stream) {
  SIU3R_CHECK(class_logits && mask_logits_cl && probs && p256 && lab_map && seg, "panoptic_stage1: null pointer");
  SIU3R_CHECK(Q <= 128 && C <= 64, "panoptic_stage1: supports up to 128 queries / 64 classes (Q=%d C=%d)", Q, C);
  SIU3R_CHECK((int64_t)B * T * mask_size * mask_size * Q < 0x7fffffffll && (int64_t)T * H * W < 0x7fffffffll,
              "panoptic_stage1: mask volume / pixel count must stay below 2^31");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pp_class_kernel, dim3(B), dim3(64), 0, s, class_logits, probs, scores, labels, kept_idx, n_keep, Q, C, threshold, area, orig);
  const int64_t nvol = (int64_t)B * T * mask_size * mask_size * Q;
  const int64_t npix = (int64_t)T * H * W;
  const dim3 ag((unsigned)cdiv64(npix, 256), B);
  hipLaunchKernelGGL(pp_mask256_kernel, dim3((unsigned)cdiv64((int64_t)mask_size * mask_size, 256), Q, B * T), dim3(256), 0, s, mask_logits_cl, kept_idx, n_keep, p256, T, IH, IW, mask_size, Q);
  hipLaunchKernelGGL(pp_argmax_kernel, ag, dim3(256), 0, s, p256, scores, kept_idx, n_keep, lab_map, area, orig, T, H, W, mask_size, Q, mask_threshold);
  hipLaunchKernelGGL(pp_accept_kernel, g1(B, 64), dim3(64), 0, s, area, orig, kept_idx, n_keep, labels, scores, seg_id, seg_label, seg_fused, seg_score, acc_list, n_acc, Q, overlap, fuse_mask, B);
  hipLaunchKernelGGL(pp_write_kernel, dim3((unsigned)cdiv64(npix, 256), B), dim3(256), 0, s, lab_map, seg_id, seg_label, n_keep, seg, sem, ins, npix, Q);
  SIU3R_LAUNCH_CHECK("siu3r_panoptic_stage1");
  return 0;
}

extern "C" int siu3r_panoptic_qcl(const float* p256, const float* probs, const int32_t* kept_idx, const int32_t* acc_list,
                                  int nq, float* out, int b, int T, int H, int W, int mask_size, int Q, int C,
                                  void* stream) {
  SIU3R_CHECK(p256 && probs && kept_idx && acc_list && out && nq > 0, "panoptic_qcl: bad arguments");
  SIU3R_CHECK((int64_t)T * H * W * nq * C < 0x7fffffffll, "panoptic_qcl: the volume must stay below 2^31 elements");
  hipLaunchKernelGGL(pp_qcl_kernel, g1((int64_t)T * H * W * nq), dim3(256), 0, (hipStream_t)stream, p256, probs, kept_idx, acc_list, nq, out, b, T, H, W, mask_size, Q, C);
  SIU3R_LAUNCH_CHECK("siu3r_panoptic_qcl");
  return 0;
}

// ======================= query-class-logit lifting (reference src/pipeline.py:137-193) ==========================
namespace {

// One WAVE per pixel (64 consecutive pixels per wave): a pixel's q x C logits are contiguous (channel-last), so the lanes read them
// coalesced and reduce with shuffles; lane k keeps pixel k's result and the 64 results leave with coalesced stores.  (A thread per pixel
// walked its 400-2500 bytes alone: every load of a wave touched 64 different cache lines -- 5 ms for 6 views x 512^2 x 105 channels
// instead of the 0.25 ms the bytes cost.)  The reference takes, per class, the first query attaining the maximum, then the first class (in
// rolled order: void first) attaining the maximum of those (pipeline.py:141-150): the winner is the smallest key = rolled class * q + query
// among the elements equal to the global maximum.
__global__ void lift_pixel_kernel(const float* qc, int64_t npix, int q, int C, float thr, int64_t* sem_id, int64_t* ins_id,
                                  int32_t* first_pix) {
  const int lane = threadIdx.x & 63;
  const int64_t base = ((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)) * 64;  // first pixel of this wave
  if (base >= npix) return;
  const int n = q * C;
  int my_sem = 0, my_q = 0;
  const int cnt = (int)((npix - base) < 64 ? (npix - base) : 64);
  for (int k = 0; k < cnt; ++k) {
    const float* v = qc + (base + k) * (int64_t)n;
    float best = -INFINITY;
    int bkey = 0x7fffffff;
    for (int e = lane; e < n; e += 64) {
      const float x = v[e];
      const int qi = e / C, orig = e - qi * C;
      const int cr = orig == C - 1 ? 0 : orig + 1;  // rolled class index: 0 <- void (last), k <- k-1   (:144-149)
      const int key = cr * q + qi;
      if (x > best || (x == best && key < bkey)) {
        best = x;
        bkey = key;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int ok = __shfl_xor(bkey, o);
      if (ob > best || (ob == best && ok < bkey)) {
        best = ob;
        bkey = ok;
      }
    }
    int sem = bkey / q;
    int qidx = bkey - sem * q + 1;
    if (best < thr) sem = 0;                // (:161-162)
    if (sem == 0) qidx = 0;                     // (:163)
    if (lane == k) {
      my_sem = sem;
      my_q = qidx;
    }
  }
  if (lane < cnt) {
    sem_id[base + lane] = my_sem;
    ins_id[base + lane] = my_q;
    // first pixel owned by each query: one same-address atomic per pixel is ~2.4 ns each (3 ms for 1.5 M pixels); a plain look at the
    // current minimum first (possibly stale: then the atomic is merely redundant) leaves only the record-setting pixels
    if (my_q > 0 && (int32_t)(base + lane) < ((volatile int32_t*)first_pix)[my_q - 1]) atomicMin(&first_pix[my_q - 1], (int32_t)(base + lane));
  }
}

__global__ void lift_label_kernel(const int64_t* sem_id, const int32_t* first_pix, int32_t* q_label, int q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= q) return;
  q_label[i] = first_pix[i] == 0x7fffffff ? -1 : (int32_t)sem_id[first_pix[i]];  // sem_id of the first owned pixel (:166-170)
}

__global__ void lift_fuse_kernel(const int64_t* sem_id, int64_t* ins_id, int64_t npix, int num_queries, uint32_t stuff_mask) {
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int s = (int)sem_id[pix];
  if (s >= 1 && s <= 32 && ((stuff_mask >> (s - 1)) & 1u)) ins_id[pix] = num_queries + (s - 1) + 1;  // (:180-184)
}

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

extern "C" int siu3r_lift_ids(const float* qc, int V, int H, int W, int q, int C, float sem_threshold, int num_queries,
                              uint32_t stuff_mask, int64_t* sem_id, int64_t* ins_id, int32_t* first_pix, int32_t* q_label,
                              void* stream) {
  SIU3R_CHECK(qc && sem_id && ins_id && first_pix && q_label && q > 0 && C > 1, "lift_ids: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)V * H * W;
  SIU3R_CHECK(npix < 0x7fffffff, "lift_ids: too many pixels");
  hipLaunchKernelGGL(fill_i32_kernel, g1(q), dim3(256), 0, s, first_pix, q, 0x7fffffff);
  hipLaunchKernelGGL(lift_pixel_kernel, g1(((npix + 63) / 64) * 64), dim3(256), 0, s, qc, npix, q, C, sem_threshold, sem_id, ins_id, first_pix);
  hipLaunchKernelGGL(lift_label_kernel, g1(q), dim3(256), 0, s, sem_id, first_pix, q_label, q);
  hipLaunchKernelGGL(lift_fuse_kernel, g1(npix), dim3(256), 0, s, sem_id, ins_id, npix, num_queries, stuff_mask);
  SIU3R_LAUNCH_CHECK("siu3r_lift_ids");
  return 0;
}
