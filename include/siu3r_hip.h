/*
 * siu3r_hip.h -- C ABI of libsiu3r_hip.so, the MI355X (gfx950) hot path of SIU3R inference.
 *
 * Conventions:
 *   - every entry point returns 0 on success, non-zero on error; siu3r_last_error() returns a
 *     thread-local message.  The Python wrappers raise RuntimeError, mirroring the reference's
 *     TORCH_CHECK behaviour (reference: src/models/croco/curope/curope.cpp:54-59, kernels.cu:91-94).
 *   - all pointers are raw DEVICE pointers unless stated; no entry point allocates, synchronises
 *     or owns memory; kernels are enqueued on `stream` (a hipStream_t passed as void*).
 *   - dtype codes: SIU3R_BF16 = 0, SIU3R_F32 = 1 everywhere; siu3r_rope2d (seam 1) also takes SIU3R_F16 = 2 and SIU3R_F64 = 3,
 *     the types the reference's kernel dispatches (kernels.cu:101).
 *   - activations are channel-last ("NHWC" / token-major) everywhere.
 *
 * Each group cites the reference interface it replaces.
 */
#ifndef SIU3R_HIP_H
#define SIU3R_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SIU3R_BF16 0
#define SIU3R_F32 1
#define SIU3R_F16 2
#define SIU3R_F64 3

const char* siu3r_last_error(void);
#define SIU3R_ABI_VERSION 10 /* 10: siu3r_stem7x7_x3 (the Gaussian heads' stem as one dedicated kernel), siu3r_proj_rows_x3 (the heads' last 1 x 1 convolutions as row streams); 9: siu3r_raster_project_c2w + siu3r_raster_cam.k2_near / k2_far (the reference renderer's own pose tensors, consumed on the device), siu3r_raster_tune (replaces the SIU3R_FEAT_FORM / SIU3R_FEAT_NP environment switches; the shared-batch matrix-core composite is gone: no workspace = 32-channel kernel); 8: LPIPS (siu3r_maxpool2x2s2, siu3r_lpips_layer); 7: device-side poses for the gsplat seam (siu3r_raster_project_dp, siu3r_sh_eval_dp, siu3r_blend_background_dp), precomputed K2 colours (sh_degree < 0), all-channel list composite; 6: pre-split bf16x3 activations (siu3r_gemm_params.c_x3 / a_x3, siu3r_gemm_plan_t.a_x3_ok / c_x3_ok), siu3r_attn_params.kv_bxor / kv_x3, siu3r_gemm_params.c_x3_col0; 5: siu3r_raster_sort takes stage 1's counters (culled Gaussians leave the sort), NaN-poisoned views on entry overflow, siu3r_gemm_tune key 4; 4: siu3r_gemm_plan, split-K workspace capacities + self-resetting tickets, tile_cfg; 3: view-batched sort-free rasterizer */
int siu3r_abi_version(void);

/* ---- seam 1: curope.rope_2d(tokens, positions, base, fwd)
 * reference: src/models/croco/curope/curope.cpp:49-65 (dispatch), kernels.cu:17-108 (kernel).
 * tokens [B,N,H,D] in place (stride_d == 1), positions int64 [B,N,2] contiguous, D % 4 == 0.    */
int siu3r_rope2d(void* tokens, int dtype, int B, int N, int H, int D, int64_t stride_b,
                 int64_t stride_n, int64_t stride_h, const int64_t* positions, float base, float fwd,
                 void* stream);

/* ---- GEMM / implicit-GEMM convolution on MFMA (bf16 operands, fp32 accumulate).
 * Replaces every nn.Linear / nn.Conv2d / nn.ConvTranspose2d(k==stride) on the path
 * (reference: croco/blocks.py:58-79,94-112; heads/dpt_block.py; vit_adapter.py; video_seg_decoder.py).
 * C[M,N] = epilogue(A[M,K] * W[N,K]^T).  W is pre-packed bf16 [N, Kpad] (Kpad % 64 == 0, zero padded);
 * w_lo (may be NULL) holds the bf16 residual of the fp32 weight and enables the 3-pass
 * "bf16x3" mode (A must then be fp32): C = Ahi*Whi + Ahi*Wlo + Alo*Whi, ~fp32 accuracy. */
typedef struct {
  const void* a;          /* activations */
  const void* w_hi;       /* bf16 [N,Kpad] */
  const void* w_lo;       /* bf16 [N,Kpad] or NULL */
  void* c;                /* output */
  const float* bias;      /* [N] (or [Cout] for out_mode 1) or NULL */
  const void* residual;   /* added after activation, same indexing as c, or NULL */
  int32_t m, n, k, kpad;
  int64_t lda, ldc, ldr;  /* row strides in elements (dense A / C / residual) */
  int32_t a_dtype, c_dtype, r_dtype;
  int32_t act;            /* 0 none, 1 exact-erf GELU, 2 ReLU */
  int32_t relu_in;        /* apply ReLU to A while loading */
  /* batching: blockIdx.z selects the batch; strides in elements */
  int32_t batch;
  int64_t sa, sw, sc, sr;
  /* A addressing: 0 dense rows; 1 NHWC conv gather; 2 NCHW fp32 16x16 patchify (patch-embed) */
  int32_t a_mode;
  int32_t ih, iw, cin, kh, kw, stride, pad, oh, ow; /* conv geometry (a_mode 1,2) */
  /* output addressing: 0 row-major; 1 conv-transpose (kernel==stride) pixel shuffle:
     n = (ky*up+kx)*cout + co, row m=(b,iy,ix) -> out[b, iy*up+ky, ix*up+kx, co] */
  int32_t out_mode, up, cout;
  /* fused epilogue extra: bilinear x2 (align_corners=True) upsample-add of a low-res NHWC map
     (GS head: feat_up(path_1) + ReLU(conv7x7(img)), reference dpt_gs_head.py:160-162) */
  const void* up_src;     /* [B, oh/2, ow/2, n] or NULL */
  int32_t up_dtype;
  /* fused RoPE2D epilogue (reference croco/blocks.py:101-103 + curope kernels.cu:17-82) for the QKV / projq /
     projk GEMMs: output columns [0, rope_ncols) are heads of dim 64 laid out [u_Y v_Y u_X v_X]; row (z, m) is
     the token with position rope_pos[(z*m_rows + m)*2 + {y,x}]; tables are fp32 [max_pos, 16]. NULL = off. */
  const float* rope_cos;
  const float* rope_sin;
  const int64_t* rope_pos;
  int32_t rope_ncols;
  int32_t map_gx, map_rm, map_rn; /* filled by the launcher (XCD-aware tile map); callers leave them 0 */
  uint64_t* trace;                /* tuning aid, normally NULL: 8 shader-clock stamps per workgroup (tools/gemm_trace.py) */
  const void* w_x3;               /* bf16x3 on the LDS-DMA path: both planes interleaved per 32-deep K tile, bf16 [N, Kpad/32, 2, 32]
                                     (hi 32 | lo 32); NULL = use w_hi / w_lo with the register-staged kernel */
  /* two-level batching (bmod > 0): blockIdx.z = zo * bmod + zi; element offsets A: zo*sa + zi*sa_i, C: zo*sc + zi*sc_i, residual:
     zo*sr + zi*sr_i; the weight set is zi (offset zi*sw); bias / ln_c1 / ln_c2 advance by zi*sbias floats.  This is how the two
     decoder sides (dec_blocks / dec_blocks2, reference backbone_croco.py:231-255) run as ONE launch: zi = side, zo = batch item,
     and a side's cross-attention memory is the OTHER view (a negative sa_i).  bmod == 0: single level (z*sa, z*sw, z*sc, z*sr). */
  int32_t bmod;
  int64_t sa_i, sc_i, sr_i, sbias;
  /* LayerNorm folded into this GEMM (reference croco/blocks.py:127-130,186-191: x + f(LN(x))): A holds the UN-normalised rows, the
     packed weight is W diag(gamma), and the epilogue computes rstd_m * (acc - mean_m * ln_c1[n]) + ln_c2[n] with
     ln_c1[n] = sum_k W'[n,k], ln_c2[n] = sum_k beta_k W[n,k] + b[n] (bias must be NULL).  mean / rstd of row m come from ln_stats:
     ln_tiles (mean, M2) partials of 64 columns each (the last one of K - 64*(ln_tiles-1)), written by the GEMM that produced A
     (stats_out below), row (zo, zi, m) at float2 index ((zo*ln_sz + zi*ln_sz_i + m*ln_ldm) * ln_tiles). */
  const float* ln_stats;
  const float* ln_c1;
  const float* ln_c2;
  int32_t ln_tiles;
  float ln_eps;
  int64_t ln_ldm, ln_sz, ln_sz_i;
  /* row statistics of THIS GEMM's output (out_mode 0, after bias / activation / residual): per 64-column tile the mean and the
     centred sum of squares (Welford partials; merged by the consumer), float2 index ((zo*st_sz + zi*st_sz_i + m*st_ldm) * tiles_n + tile_n) */
  float* stats_out;
  int64_t st_ldm, st_sz, st_sz_i;
  void* c_aux;                    /* optional bf16 copy of the output (same indexing as c): the next GEMM's A operand in bf16 mode */
  /* split-K over workgroups: blockIdx.y = K slice; every slice writes its fp32 partial tile to sk_ws [tiles, splitk, BM * BN] and draws
     a ticket from sk_cnt [tiles] (int32, ZERO before the first launch that uses them); the last arriver of a tile sums the slabs in
     slice order (deterministic), runs the epilogue and resets the ticket, so one zero-initialised counter array serves any number
     of launches ON ONE STREAM.  For launches with few tiles and a long K (at batch 1 most of the decoder / head / Mask2Former GEMMs).
     splitk: 0 = the launcher decides (siu3r_gemm_plan; 1 when no workspace is attached), 1 = never, > 1 = this many slices (the
     workspace must hold siu3r_gemm_plan's ws_floats / counters for it).  sk_ws_floats / sk_cnt_n: capacities of the workspace. */
  int32_t splitk;
  float* sk_ws;
  int32_t* sk_cnt;
  int64_t sk_ws_floats;
  int32_t sk_cnt_n;
  int32_t tile_cfg;               /* 0 = the launcher decides; SIU3R_TILE_* forces a kernel family / tile (tools, tests) */
  int32_t m_main;                 /* internal (launcher): rows covered by the tiled kernel when a skinny launch multiplies the last <= 32 rows */
  int32_t sk_gx;                  /* internal (launcher): > 0 = the ping-pong launch carries the remainder rows itself, as workgroups blockIdx.x >= sk_gx */
  /* Pre-split bf16x3 activations.  A bf16x3 product multiplies hi = the upper 16 bits of every fp32 activation and lo = bf16(a - hi);
     the ping-pong kernels normally derive both inside their K loop, which costs 12-18 % of a launch.  c_x3 != NULL: the epilogue ALSO
     writes the (fp32, post-activation, post-residual) output as those two planes, interleaved per 32 columns like w_x3 -- row m at the
     byte offset of row m of c, then [n / 32][hi 32 | lo 32] bf16: a buffer of the size and strides of the fp32 output; c may then be
     NULL (planes only).  a_x3 != 0: `a` points at such planes (dense A only, k % 64 == 0, lda and the batch strides multiples of 32):
     same products in the same order as with the fp32 operand, hence bit-identical results.  Both need a ping-pong plan and the common
     output form (siu3r_gemm_plan_t.a_x3_ok / c_x3_ok say whether the plan of THIS block can consume / emit planes; siu3r_gemm fails
     loudly otherwise). */
  void* c_x3;
  int32_t a_x3;
  int32_t c_x3_col0;              /* planes are written for output columns >= c_x3_col0 only (a multiple of 64; 0 = all).  With c_x3 == c the
                                     buffer is MIXED: fp32 in columns [0, c_x3_col0), planes behind them (the q | k | v projection: q stays
                                     fp32, k and v are pre-split for the attention kernel) */
} siu3r_gemm_params;
#define SIU3R_TILE_AUTO 0
#define SIU3R_TILE_128x64 -1   /* the 128 x 64 LDS-DMA / register-staged kernels (gemm_dma.hip, gemm.hip) */
#define SIU3R_TILE_PP_256x256 1 /* 8-wave ping-pong kernels (gemm_pp.hip) */
#define SIU3R_TILE_PP_256x128 2
#define SIU3R_TILE_PP_128x128 3
int siu3r_gemm(const siu3r_gemm_params* p, void* stream);
/* What siu3r_gemm will launch for these parameters (same decision function): tile family, tile, split-K slices, whether the last
 * rows go to the skinny kernel, the split-K workspace it needs (fp32 elements / int32 counters) and the kernel's name as rocprofv3
 * prints it.  Callers that attach a workspace of at least this size get the split; smaller workspaces reduce splitk. */
typedef struct {
  int32_t tile_cfg, bm, bn, splitk, skinny_rows, counters;
  int64_t ws_floats;
  char kernel[160];
  int32_t a_x3_ok, c_x3_ok; /* this plan can read A as pre-split planes (a_x3) / write the output as planes (c_x3) */
} siu3r_gemm_plan_t;
int siu3r_gemm_plan(const siu3r_gemm_params* p, siu3r_gemm_plan_t* out);
/* tuning aid (tools/, tests): key 0 = process-wide default of siu3r_gemm_params.tile_cfg (SIU3R_TILE_*; also env SIU3R_GEMM_PP), key 1 =
 * 1 disables the skinny remainder-row launch, key 2 = 1 disables split-K, key 3 = 1 ignores the table of measured tile choices
 * (csrc/gemm_tuned.h; also env SIU3R_GEMM_NO_TUNED) so that every problem is priced by the cost model, key 4 = 1 sends the remainder
 * rows to the skinny launch whenever it is applicable (tests).  Not for concurrent use. */
int siu3r_gemm_tune(int key, int value);

/* ---- LayerNorm over the last dim (fp32 in, act-dtype out); reference: nn.LayerNorm call sites
 * croco/blocks.py:127-191 (eps 1e-6), video_seg_decoder.py:945-1018 (eps 1e-5). */
int siu3r_layernorm(const float* x, void* y, int y_dtype, const float* gamma, const float* beta,
                    int64_t rows, int C, int64_t ldx, int64_t ldy, float eps, void* stream);
/* same, plus a second bf16 copy y2 of the result (row stride ldy2): the fp32 output remains the residual stream, the bf16 one is
 * the next GEMM's A operand */
int siu3r_layernorm2(const float* x, void* y, int y_dtype, void* y2_bf16, const float* gamma, const float* beta, int64_t rows, int C,
                     int64_t ldx, int64_t ldy, int64_t ldy2, float eps, void* stream);

/* ---- fused attention (flash style, online softmax), head_dim 64 or 32.
 * Replaces Attention/CrossAttention (croco/blocks.py:94-112,149-169, incl. RoPE2D on q,k) and
 * the Mask2Former decoder attentions (video_seg_decoder.py:782-912, nn.MultiheadAttention 946-983).
 * q,k,v: element strides (batch, token, head); head_dim contiguous.  out [B,Nq,H*D] contiguous.
 * rope_cos/sin: fp32 [max_pos, D/4] tables or NULL; qpos/kpos int64 [B,N,2] (y,x).
 * mask: uint8 [B,Nq,mask_ld] (1 = blocked; row stride mask_ld >= Nk, multiple of 64), shared by all heads, or NULL. split3 = bf16x3 mode (fp32 io). */
typedef struct {
  const void *q, *k, *v;
  void* out;
  int32_t dtype;          /* dtype of q,k,v,out */
  int32_t B, H, Nq, Nk, D;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh;
  float scale;
  const float *rope_cos, *rope_sin;
  int32_t rope_max_pos;
  const int64_t *qpos, *kpos;
  const uint8_t* mask;
  int32_t split3;
  int64_t mask_ld;        /* bytes per mask row (>= Nk, multiple of 64; padding bytes are ignored) */
  float* ws;            /* split-KV workspace, fp32 [B, H, splits, ceil(Nq/128)*128, D + 4] (partial O, running max, partial sum, 2 pad);
                           NULL = no split */
  int32_t splits;       /* > 1: the keys are cut into `splits` ranges of whole 64-key tiles, one workgroup per (query tile, range),
                           combined by a second kernel.  Fast paths only (bf16, or fp32 with split3; no RoPE-on-load): few query tiles
                           against many keys (Mask2Former's 100 queries x 512..8192 keys) */
  int32_t kv_bxor;      /* batch item b reads the K / V of batch item b ^ kv_bxor (0: its own).  The two decoder sides of a pair are batch items
                           2b and 2b + 1 and a side's cross-attention memory is projected from the OTHER side's rows: with the K / V
                           projection merged into the launch that also makes q / k / v of the row's own side, the memory of side g lies in
                           the row block of side 1 - g (kv_bxor = 1).  B must be a multiple of kv_bxor + 1 (power of two). */
  int32_t kv_x3;        /* bf16x3 (fp32 tensors, split3): k and v point at PRE-SPLIT planes -- per token and head the 64 dims as two 128-byte
                           segments [hi 32 | lo 32] bf16, as a projection GEMM writes them with c_x3 / c_x3_col0 -- same strides as the
                           fp32 tensors.  Served by the pipelined kernel only (siu3r_attention_kv_x3_ok); anything else is an error. */
} siu3r_attn_params;
int siu3r_attention(const siu3r_attn_params* p, void* stream);
/* 1 if siu3r_attention would accept these parameters with kv_x3 = 1 (head_dim 64, no mask / RoPE-on-load / key split, >= 64 keys,
 * fp32 + split3, segment-aligned K / V rows), else 0 */
int siu3r_attention_kv_x3_ok(const siu3r_attn_params* p);

/* ---- element-wise / gather kernels (see DESIGN.md for the HBM roofline of each) ---------- */
/* y = a + b (b broadcast over rows when b_rows < rows: row r uses b[r % b_rows]) */
int siu3r_add(const float* a, const float* b, float* y, int64_t rows, int64_t b_rows, int C, void* stream);
/* image [N,3,H,W] fp32 (NCHW) -> [N,H,W,cpad] channel-last, zero padded channels: cpad = 8 (bf16 or fp32) or 4 (fp32: one 16-byte
 * pixel per gather chunk of the bf16x3 convolution) */
int siu3r_pack_image_nhwc(const float* img, void* out, int out_dtype, int N, int H, int W, int cpad, void* stream);
/* bilinear resize NHWC; align_corners as in F.interpolate; y = affine(resize(x) + addend)
 * (heads/dpt_block.py:230-235 align=1; vit_adapter.py:429-433, video_seg_decoder.py:2173-2178 align=0) */
int siu3r_resize_bilinear(const void* x, int x_dtype, void* y, int y_dtype, const void* addend, int add_dtype,
                          const float* ch_scale, const float* ch_shift, int N, int IH, int IW, int OH, int OW,
                          int C, int align_corners, void* stream);
/* the same on batch-strided inputs: x [N,IH,IW,C] / addend [N,OH,OW,C] whose batch items lie x_batch_stride / addend_batch_stride
 * elements apart (token maps that are views of a longer per-item sequence: vit_adapter.py:393-433 reads them in place) */
int siu3r_resize_bilinear_strided(const void* x, int x_dtype, void* y, int y_dtype, const void* addend, int add_dtype,
                                  const float* ch_scale, const float* ch_shift, int N, int IH, int IW, int OH, int OW,
                                  int C, int align_corners, int64_t x_batch_stride, int64_t addend_batch_stride, void* stream);
/* y = x*scale[c] + shift[c] (+ addend) ; eval-mode BatchNorm folded (vit_adapter.py:436-440) */
int siu3r_affine_add(const void* x, int x_dtype, const void* addend, int add_dtype, void* y, int y_dtype,
                     const float* ch_scale, const float* ch_shift, int64_t rows, int C, void* stream);
/* rows = batch items x rows_per_batch; batch items of x / addend lie *_batch_stride elements apart; y is dense */
int siu3r_affine_add_strided(const void* x, int x_dtype, const void* addend, int add_dtype, void* y, int y_dtype,
                             const float* ch_scale, const float* ch_shift, int64_t rows, int C, int64_t rows_per_batch,
                             int64_t x_batch_stride, int64_t addend_batch_stride, void* stream);
/* 3x3 stride-2 pad-1 max pool, NHWC (vit_adapter.py:226) */
int siu3r_maxpool3x3s2(const void* x, void* y, int dtype, int N, int IH, int IW, int C, void* stream);
/* 2x2 stride-2 max pool (floor), NHWC: the pooling of the VGG16 feature stack under the evaluator's LPIPS (evaluator.py:55-57, 263) */
int siu3r_maxpool2x2s2(const void* x, void* y, int dtype, int N, int IH, int IW, int C, void* stream);
/* one feature tap of LPIPS (torchmetrics LearnedPerceptualImagePatchSimilarity("vgg"); third-party, absent from the reference tree):
 * dist[p] = sum_c w[c] * (f0[p][c] / sqrt(eps + |f0[p]|^2) - f1[p][c] / sqrt(eps + |f1[p]|^2))^2 for f0, f1 [npix, C] fp32, w [C] */
int siu3r_lpips_layer(const float* f0, const float* f1, const float* w, float* dist, int64_t npix, int C, float eps, void* stream);
/* The Gaussian heads' stem, bf16x3: out = ReLU(conv7x7(img, pad 3) + bias) + up_x2_bilinear_align_corners(up_src)
 * (reference src/models/heads/dpt_gs_head.py:71-77 input_merger = Conv2d(3, 256, 7, 1, 3) + ReLU, :158-162 feat_up(path_1) + direct_img_feat).
 * img [B, G, H, W, 4] fp32 (RGB + one zero channel); G weight sets (head g of every batch item): wfrag = MFMA B fragments of the
 * [256, 3, 7, 7] weights, [G][8][14][2][64] x 8 bf16 (siu3r_amd/ops.py pack_stem7: K ordered (ky, kx in 0..7, c in 0..3), hi = bf16(w),
 * lo = bf16(w - hi)); bias [G, 256] or NULL; up_src [B, G, H/2, W/2, 256] fp32 or NULL; out [B, G, H, W, 256]: fp32 values, or
 * (planes_out != 0) the pre-split planes a ping-pong GEMM reads as its A operand (siu3r_gemm_params.a_x3: per pixel and 32 channels one
 * 128-byte line [hi 32 | lo 32]).  H and W multiples of 16.  Same bf16x3 arithmetic as siu3r_gemm's convolution, another K order. */
int siu3r_stem7x7_x3(const float* img, const void* wfrag, const float* bias, const float* up_src, float* out, int B, int G, int H, int W,
                     int planes_out, void* stream);
/* The last 1 x 1 convolution of a DPT head as a row stream, bf16x3: out[z, m, 0..N) = x[z, m, :] W_g^T + b_g, g = z % G
 * (reference src/models/heads/dpt_block.py:384-391 gs_params head, Conv2d(256, 83, 1); :357-369 regression head, Conv2d(128, 3 + conf, 1)).
 * x [Z, M, K] fp32 with contiguous rows; wfrag = MFMA B fragments of the G weight matrices [N, K], [G][ceil(N/32)][K/16][2][64] x 8 bf16
 * (siu3r_amd/ops.py pack_proj); bias [G, 32 ceil(N/32)] zero padded, or NULL; out [Z, M, N] fp32 with row stride ldc floats and
 * out_z_stride floats between consecutive z (0 = M * ldc).
 * Instantiations: K = 256 with 65 <= N <= 96, K = 128 with N <= 32.  Same bf16x3 products as siu3r_gemm, another K order. */
int siu3r_proj_rows_x3(const float* x, const void* wfrag, const float* bias, float* out, int Z, int G, int64_t M, int K, int N, int ldc,
                       int64_t out_z_stride, void* stream);
/* depth-wise 3x3 + bias + GELU over the 3 token scales of the adapter ConvFFN (vit_adapter.py:16-59) */
int siu3r_dwconv3x3_gelu(const void* x, void* y, int dtype, const float* w9c, const float* bias, int B, int H,
                         int W, int C, void* stream);
/* multi-scale deformable attention sampling core (vit_adapter/blocks.py:217-267):
 * value [B,S,heads,d]; offs_aw fp32 [B,Q,heads*L*P*3] = per row [offsets (h,L,P,2) | logits (h,L,P)];
 * ref fp32 [Q,L,2]; shapes int32 [L,2] (h,w) host pointer; out [B,Q,heads*d]. */
int siu3r_msdeform_sample(const void* value, int v_dtype, const float* offs_aw, const float* ref,
                          const int32_t* shapes_host, void* out, int out_dtype, int B, int S, int Q, int heads,
                          int d, int L, int P, void* stream);
/* GroupNorm(32) on NHWC (video_seg_decoder.py:2002-2050): two kernels, stats then apply
 * y = relu?(gn(x)) + addend? ; stats is a [N,groups,2] fp32 workspace */
int siu3r_groupnorm(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                    float* stats_ws, const void* addend, int add_dtype, int relu, int N, int HW, int C,
                    int groups, float eps, void* stream);
/* pts3d = xyz/|xyz| * expm1(|xyz|)  (heads/postprocess.py:45-61), in place on [n,3] fp32 */
int siu3r_pts3d_exp(float* xyz, int64_t n, void* stream);
/* UnifiedGaussianAdapter.forward (gaussian_adapter.py:81-110): raw [n,83] -> fields (all fp32) */
int siu3r_gaussian_adapter(const void* raw, int raw_dtype, float* opacities, float* scales, float* rotations,
                           float* harmonics, float* covariances, int64_t n, void* stream);
/* Mask2Former attention mask (video_seg_decoder.py:1461-1478 + 1306-1308): mask logits
 * [B,T,IH,IW,Q] (channel-last) -> uint8 [B,Q,out_ld] (first T*OH*OW bytes of each row), 1 = blocked; rows fully
 * blocked are cleared.  row_counts_ws: B*Q int32 of scratch (zeroed by the call; holds a "row has an open key" flag). */
int siu3r_m2f_attn_mask(const float* mask_logits, uint8_t* out, int32_t* row_counts_ws, int B, int T, int IH,
                        int IW, int OH, int OW, int Q, int64_t out_ld, void* stream);

/* ---- Gaussian splat rasterizer (tile-binned, all views of a call in every launch).  Replaces the reference's two
 * un-vendored CUDA dependencies at their call sites: GaussianRasterizer(settings)(means3D, ..., cov3D_precomp, ...) ->
 * (image, radii, depth, opacity, n_touched) (reference src/models/cuda_splatting.py:90-118; mode 0) and
 * gsplat.rasterization(means, covars, opacities, colors[N,C], viewmats, Ks, width, height, near_plane, far_plane) ->
 * (colors, alphas, meta) (reference src/models/gaussian_renderer.py:92-106; mode 1).  The reference loops over the views in
 * Python (cuda_splatting.py:82-121); here V cameras go in as one HOST array (copied to `cams_dev` on the stream) and
 * blockIdx.y is the view.  All constants of the published algorithms are explicit parameters.
 * Stages: project -> sort (one stable depth radix sort per view) -> bin (depth-ordered coarse bins of cb x cb tiles) ->
 * composite_rgb (mode 0; walks the bins, no per-tile lists) | tile_lists + composite_feat (mode 1). */
typedef struct {
  int32_t mode;        /* 0 = K2 (3DGS family), 1 = K3 (gsplat family) */
  int32_t width, height;
  float w2c[16];       /* world->camera, row-major (column-vector convention) */
  float proj[16];      /* K2: full projection P = Proj * W2C, row-major */
  float tanfovx, tanfovy;
  float campos[3];
  float bg[3];
  int32_t sh_degree;   /* K2: 0..4; -1 = `colors` are precomputed [G,3] colours, blended as given (no SH, no +0.5, no clamp) */
  int32_t sh_band4;    /* K2: evaluate SH coefficients 16..24 (open question of the fork; default 0) */
  float k2_znear_cull; /* 0.2 */
  float fx, fy, cx, cy; /* K3: pixel-unit intrinsics */
  float near_plane, far_plane; /* K3; near_plane must be > 0 (depth keys are the float bit patterns) */
  float eps2d;         /* 0.3 */
  float radius_clip;
  float extent_sigma;  /* 3.33 */
  int32_t opacity_aware_extent;
  float alpha_min;     /* 1/255 */
  float alpha_max;     /* 0.99 (K2) / 0.999 (K3) */
  float t_min;         /* 1e-4 */
  float dilation;      /* K2 low-pass 0.3 */
  int32_t nt_post_blend; /* K2 n_touched: count a pixel when the transmittance AFTER blending the Gaussian is > 0.5
                            (1, the MonoGS fork's `test_T > 0.5f`) or the one before it (0) */
  float k2_near, k2_far; /* K2, siu3r_raster_project_c2w only: the planes of the [0, 1]-depth projection matrix the device derives
                            (cuda_splatting.py:16-43); ignored when `proj` is given by the host */
} siu3r_raster_cam;
/* frame geometry / workspace sizes: out8 = {gw, gh, T = tiles, cb = coarse-bin edge in tiles, NB = coarse bins,
 * nchunks_sort (columns of rs_hist), nchunks_bin (columns of bin_hist), sizeof(siu3r_raster_cam)} */
int siu3r_raster_geometry(int width, int height, int64_t G, int32_t* out8);
/* stage 1: project G Gaussians (means [G,3]; cov: cov_stride 6 = upper-triangular (xx,xy,xz,yy,yz,zz), 9 = row-major 3x3;
 * opacities [G]; colors: mode 0 SH coefficients, sh_planar 0 = [G,channels,3] (the layout of cuda_splatting.py:65), 1 = [G,3,25]
 * (Gaussians.harmonics as stored), mode 1 unused; all shared by the V views) for every view.  cams_host: V structs in HOST memory,
 * copied to cams_dev (device, V * sizeof(siu3r_raster_cam) bytes) on the stream.  Outputs, all [V, G, ...]: rec fp32 [.,12] = {mean2d x, y,
 * depth, 0 | conic a, b, c, opacity | r, g, b, 0} (16-byte aligned), radii [.,2] i32, rect [.,4] i32 (tile rect [min,max)),
 * tiles_touched i32, keys u32 (depth bits; 0xffffffff = culled).  stats: u64 [V,4] = {visible Gaussians, tile pairs D, coarse
 * entries E (stage 3), flags: bit 0 entries overflowed cap_e, bit 1 tile lists overflowed cap_d}; zeroed here. */
int siu3r_raster_project(const siu3r_raster_cam* cams_host, int V, void* cams_dev, int64_t G, const float* means, const float* cov,
                         int cov_stride, const float* opacities, const float* colors, int channels, int sh_planar, float* rec,
                         int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys, uint64_t* stats, void* stream);
/* the same with the POSE in device memory, gsplat family (mode 1) only -- what gsplat.rasterization receives (reference
 * src/models/gaussian_renderer.py:92-106, viewer.py:319-335): viewmats_dev [V,4,4] world->camera and Ks_dev [V,3,3] pixel-unit
 * intrinsics, row-major fp32 device tensors; they overwrite w2c / fx / fy / cx / cy of the uploaded camera blocks on the stream, so that
 * the host never reads a pose back (no synchronisation).  cams_host carries everything else (frame size, planes, thresholds). */
int siu3r_raster_project_dp(const siu3r_raster_cam* cams_host, int V, void* cams_dev, const float* viewmats_dev, const float* Ks_dev, int64_t G,
                            const float* means, const float* cov, int cov_stride, const float* opacities, const float* colors, int channels,
                            int sh_planar, float* rec, int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys, uint64_t* stats,
                            void* stream);
/* the same with the pose AS THE REFERENCE'S RENDERER HOLDS IT (SplattingCUDA.forward, src/models/gaussian_renderer.py:29-74; render_cuda,
 * cuda_splatting.py:46-121), both families: c2w_dev [V,4,4] camera-to-world extrinsics and Kn_dev [V,3,3] NORMALISED intrinsics, row-major
 * fp32 device tensors.  A one-thread-per-view kernel on the stream derives w2c = inverse(c2w with its translation * t_scale), campos and,
 * mode 0: fov from the K^-1 edge rays (utils/projection.py:247-261), tanfovx / tanfovy, proj = Proj(k2_near, k2_far, fov) * w2c; mode 1:
 * fx, fy, cx, cy = K * (width, height) -- in fp64, rounded once -- and overwrites those fields of the uploaded blocks: no pose value is
 * read on the host (the reference's .inverse() / get_fov run on its device too).  A singular c2w gives a NaN pose (nothing passes the depth test). */
int siu3r_raster_project_c2w(const siu3r_raster_cam* cams_host, int V, void* cams_dev, const float* c2w_dev, const float* Kn_dev, float t_scale,
                             int64_t G, const float* means, const float* cov, int cov_stride, const float* opacities, const float* colors,
                             int channels, int sh_planar, float* rec, int32_t* radii, int32_t* rect, int32_t* tiles_touched, uint32_t* keys,
                             uint64_t* stats, void* stream);
/* stage 2: per view, stable LSD radix sort (4 x 8 bits) of keys_a [V,G] with the Gaussian index as payload; keys_b / ids_a / ids_b
 * [V,G] ping-pong buffers; rs_hist i32 [V,256,nchunks_sort], rs_tot i32 [V,256]; stats: stage 1's counters (device).  Culled Gaussians
 * (key 0xffffffff) leave the sort in its first pass: the result is the (depth, id)-ordered list of the n = stats[v][0] VISIBLE Gaussians in
 * keys_a / ids_a [v][0, n); what lies behind is unspecified (stage 3 reads n from stats as well). */
int siu3r_raster_sort(int V, int64_t G, uint32_t* keys_a, uint32_t* keys_b, int32_t* ids_a, int32_t* ids_b, int32_t* rs_hist,
                      int32_t* rs_tot, const uint64_t* stats, void* stream);
/* stage 3: depth-ordered coarse bins.  bin_hist i32 [V,NB,nchunks_bin], bin_tot i32 [V,NB], bin_start i32 [V,NB+1] (out),
 * entries: 8-byte records [V,cap_e] (Gaussian id, rect clipped to the bin).  Entries beyond cap_e are dropped and flagged in
 * stats (the true count is stats[v][2]): size by a bound, enqueue the whole frame, check once afterwards. */
int siu3r_raster_bin(const siu3r_raster_cam* cams_host, int V, int64_t G, const uint32_t* keys, const int32_t* ids, const int32_t* rect,
                     int32_t* bin_hist, int32_t* bin_tot, int32_t* bin_start, void* entries, int64_t cap_e, uint64_t* stats, void* stream);
/* stage 4 (mode 0): image [V,3,H,W], depth [V,H,W], accumulated opacity [V,H,W], n_touched [V,G] i32 (NULL = not wanted).
 * A view whose entries overflowed cap_e (bin_start[v][NB] > cap_e) gets NaN in every output pixel: never a subtly wrong image, so the
 * caller may read stats late (asynchronously) and repeat the call with a larger cap_e. */
int siu3r_raster_composite_rgb(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* bin_start,
                               const void* entries, int64_t cap_e, const float* rec, float* image, float* out_depth, float* out_alpha,
                               int32_t* n_touched, void* stream);
/* per-tile Gaussian lists, front to back (wave ballot + prefix popcount over the coarse bins): tile_count i32 [V,T] workspace,
 * tile_start i32 [V,T+2] ([0..T] clamped to cap_d, [T+1] = true pair count), ids i32 [V,cap_d] */
int siu3r_raster_tile_lists(const siu3r_raster_cam* cams_host, int V, const int32_t* bin_start, const void* entries, int64_t cap_e,
                            int32_t* tile_count, int32_t* tile_start, int32_t* ids, int64_t cap_d, uint64_t* stats, void* stream);
/* stage 4 (mode 1): feats [G,channels] -> out [V,H,W,channels] (+ alphas [V,H,W]) over the tile lists, 32 channels per pass (alpha and
 * transmittance re-evaluated per pass).  A pixel only ever sees the feature rows of entries that reach it with alpha >= 1/255.
 * An empty scene (G == 0; then rec / feats / ids may be null) renders zero maps. */
int siu3r_raster_composite_feat(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* tile_start,
                                const int32_t* ids, int64_t cap_d, const float* rec, const float* feats, int channels, float* out,
                                float* out_alpha, void* stream);
/* the same with a WORKSPACE (device, ws_bytes >= siu3r_raster_composite_feat_ws_bytes(width, height, V, cap_d), 4-byte aligned): for
 * channels >= 32 (arrays below 4 GiB) the tile lists are first cut per 8 x 8 pixel quadrant (a conservative footprint test; 16 * cap_d +
 * 16 * T bytes per view) and every wave of the composite walks its own quadrant's list with wave-private LDS staging, all channels in
 * one pass per 192-channel chunk, the blend as rank-2 v_mfma_f32_32x32x2_f32 updates (exact f32, accumulating in list order) -- no
 * workgroup barrier in the kernel.  Identical bits to siu3r_raster_composite_feat FOR FINITE FEATURES (which it falls back to without a
 * workspace, below 32 channels, or after siu3r_raster_tune(0, 1)).  Restriction: every pixel of a quadrant takes part in every entry of
 * the quadrant's list with weight 0 where the entry does not reach it, and list tails are padded with Gaussian 0's row, so a NON-FINITE
 * feature value (0 * inf = NaN) in any listed Gaussian -- or in Gaussian 0 -- poisons pixels the 32-channel kernel leaves untouched:
 * callers that may hold non-finite features select the 32-channel kernel (tests/test_raster_gpu.py pins both behaviours). */
int64_t siu3r_raster_composite_feat_ws_bytes(int width, int height, int V, int64_t cap_d);
int siu3r_raster_composite_feat_ws(const siu3r_raster_cam* cams_host, int V, const void* cams_dev, int64_t G, const int32_t* tile_start,
                                   const int32_t* ids, int64_t cap_d, const float* rec, const float* feats, int channels, float* out,
                                   float* out_alpha, void* ws, int64_t ws_bytes, void* stream);
/* tuning / A-B switches of the rasterizer (not for concurrent use).  key 0: 1 = the 32-channel kernel for every N-channel composite, 0 =
 * default (matrix-core form where it applies); key 1: accumulator blocks per chunk of the matrix-core form (1 .. 6, default 6) */
int siu3r_raster_tune(int key, int value);
/* x *= s in place (the reference rescales the scene x10 in place, src/models/gaussian_renderer.py:43-46) */
int siu3r_scale_inplace(float* x, int64_t n, float s, void* stream);
/* query-class-logit lifting (reference src/pipeline.py:137-193): rendered [V,H,W,q*C] -> sem_id, ins_id int64 [V,H,W];
 * first_pix [q] i32 workspace/out (first pixel owned by each query, or INT_MAX), q_label [q] i32 out */
int siu3r_lift_ids(const float* qc, int V, int H, int W, int q, int C, float sem_threshold, int num_queries,
                   uint32_t stuff_mask, int64_t* sem_id, int64_t* ins_id, int32_t* first_pix, int32_t* q_label,
                   void* stream);

/* fp32 [rows, k] (row stride ldx) -> bf16 hi plane [rows,kpad] (+ optional lo = bf16(x - hi)), zero padded:
 * weight / operand pre-packing for siu3r_gemm.  x3 (optional, kpad % 32 == 0): both planes interleaved per 32-deep K tile,
 * [rows][kpad / 32][hi 32 | lo 32] = siu3r_gemm_params.w_x3.  hi may be null when x3 is given. */
int siu3r_split_bf16(const float* x, void* hi, void* lo, void* x3, int64_t rows, int k, int kpad, int64_t ldx, void* stream);

/* ---- panoptic post-process on device (integer outputs).  Replaces
 * VideoMask2FormerImageProcessor.post_process_panoptic_segmentation
 * (reference src/models/mask2former/image_processing_video_mask2former.py:1238-1481) and the label scatter of
 * SIU3RModel.post_process_gaussians (reference src/models/model.py:267-294).
 * class_logits [B,Q,C] fp32; mask_logits_cl [B,T,IH,IW,Q] fp32 (channel-last).  All other pointers are outputs /
 * workspaces: probs [B,Q,C], scores [B,Q], labels/kept_idx [B,Q] i32, n_keep [B] i32, p256 [B,T,Q,ms,ms] fp32 (the ms x ms probability planes of the KEPT queries of an item: plane k < n_keep[b] belongs to query kept_idx[b,k]; the rest is not written),
 * lab_map [B,T,H,W] i32, area/orig [B,Q] i32, per-kept-query table seg_id/seg_label/seg_fused [B,Q] i32 + seg_score
 * [B,Q] fp32, acc_list [B,Q] / n_acc [B] i32, and the maps seg/sem/ins [B,T,H,W] i32.  fuse_mask bit c = class c fuses. */
int siu3r_panoptic_stage1(const float* class_logits, const float* mask_logits_cl, float* probs, float* scores,
                          int32_t* labels, int32_t* kept_idx, int32_t* n_keep, float* p256, int32_t* lab_map,
                          int32_t* area, int32_t* orig, int32_t* seg_id, int32_t* seg_label, int32_t* seg_fused,
                          float* seg_score, int32_t* acc_list, int32_t* n_acc, int32_t* seg, int32_t* sem, int32_t* ins,
                          int B, int T, int Q, int C, int IH, int IW, int H, int W, int mask_size, float threshold,
                          float mask_threshold, float overlap, uint32_t fuse_mask, void* stream);
/* query_class_logits of batch item b in Gaussian-major layout out[(t,y,x), j, c] (model.py:261-263) */
int siu3r_panoptic_qcl(const float* p256, const float* probs, const int32_t* kept_idx, const int32_t* acc_list, int nq,
                       float* out, int b, int T, int H, int W, int mask_size, int Q, int C, void* stream);

/* ---- viewer-semantics render helpers (reference viewer.py:301-336: gsplat.rasterization(means, quats, exp(scales), sigmoid(opacities),
 * colors = cat(sh0, shN), sh_degree, backgrounds = 1); gsplat is un-vendored: published algorithm, see oracle/raster_ref.c) ------------ */
/* quats (w,x,y,z, normalised here) + scales -> the 6 upper-triangular covariance entries consumed by siu3r_raster_bin */
int siu3r_quat_scale_to_cov6(const float* quats_wxyz, const float* scales, float* cov6, int64_t G, void* stream);
/* view-dependent colour: rgb[G,3] = max(SH_degree(normalise(mean - campos)) . sh[G,ncoef,3] + 0.5, 0); campos3_host is a HOST pointer */
int siu3r_sh_eval(const float* means, const float* campos3_host, const float* sh, int ncoef, int degree, float* rgb, int64_t G, void* stream);
/* colors[pixels, channels] += (1 - alpha[pixels]) * bg[channels]; bg_host is a HOST pointer, channels <= 3 */
int siu3r_blend_background(float* colors, const float* alpha, const float* bg_host, int channels, int64_t pixels, void* stream);
/* the two helpers with their small operand in DEVICE memory (no host round trip of a pose / background tensor): campos3_dev = 3 floats,
 * bg_dev = `channels` floats (any channel count) */
int siu3r_sh_eval_dp(const float* means, const float* campos3_dev, const float* sh, int ncoef, int degree, float* rgb, int64_t G, void* stream);
int siu3r_blend_background_dp(float* colors, const float* alpha, const float* bg_dev, int channels, int64_t pixels, void* stream);

#ifdef __cplusplus
}
#endif
#endif
