/*
 * probpose_hip.h -- C ABI of libprobpose_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the ProbPose forward + decode hot path.
 * The reference (zir-vision/ProbPose_pytorch) has no FFI layer: its boundary
 * is the Python class API (probpose.model / backbone / head / codec).  The
 * Python host in probpose_pytorch_amd/ keeps that class API and binds these
 * entry points with ctypes; every entry point below cites the reference
 * interface (file:line, relative to the reference checkout) it replaces.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All device buffers are
 *     caller-owned (torch tensors' data_ptr()); nothing is allocated here
 *     (where an entry point needs scratch, a *_workspace_bytes query sizes it).
 *   - every launch takes an explicit hipStream_t (as void*); calls are
 *     asynchronous and capturable into a hipGraph (no sync, no malloc).
 *   - return value: 0 = ok, <0 = error; pp_last_error() gives the message
 *     (thread-local).  No exceptions cross the ABI.
 *   - dtype enum: PP_F32 = 0 (fp32 storage, exact-fp32 MFMA), PP_BF16 = 1
 *     (bf16 storage, bf16 MFMA, fp32 accumulate), PP_FP8 = 2 (OCP e4m3 storage,
 *     block-scaled fp8 MFMA, fp32 accumulate; pp_gemm / pp_layernorm_fp8 /
 *     pp_attention_fp8out only).
 */
#ifndef PROBPOSE_HIP_H
#define PROBPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_F32 0
#define PP_BF16 1
#define PP_FP8 2                        /* OCP e4m3 storage (gfx950 fp8 MFMA), fp32 accumulate: pp_gemm / pp_layernorm_fp8 only */

#define PP_MAX_RADIUS 9                 /* heatmap.py:178-179: s<=3.0 -> ceil(3s)<=9 */
#define PP_MAX_TAPS (2 * PP_MAX_RADIUS + 1)

/* epilogue flags of pp_gemm (bit-or) */
#define PP_EPI_BIAS 1                   /* + bias[n]                                  */
#define PP_EPI_GELU 2                   /* GELU (timm Mlp act): exact erf in f32 mode; bf16 mode: fitted form, abs err <= 3e-5 */
#define PP_EPI_RELU 4                   /* ReLU (head.py:120)                         */
#define PP_EPI_RESIDUAL 8               /* out_f32 = residual_f32 + (acc + bias)      */
#define PP_EPI_OUT_F32 16               /* store fp32 regardless of the storage dtype */
#define PP_EPI_ROWBIAS 32               /* + rowbias[(m % rowbias_period) * ldc + n]  (pos_embed) */
#define PP_EPI_ROWSTATS 128              /* RETIRED (refused): round-1 LayerNorm fusion, producer side */
#define PP_EPI_LNFOLD 256                /* RETIRED (refused): round-1 LayerNorm fusion, consumer side.  The fusion cost
                                           +11..16 us per fused GEMM against the 12 us LayerNorm launch it replaced. */
#define PP_EPI_OUT_FP8 512              /* fp8 GEMM only: store e4m3(value * out_scale) instead of bf16 (the next fp8 GEMM's A) */
#define PP_EPI_NOCLAMP 1024             /* with PP_EPI_HEATMAP: store v / temperature unclamped (the Sparsemax path, head.py:526-528) */
#define PP_EPI_FUSE_FINAL 2048           /* bf16, N = 256 (tile 9): the epilogue also applies the final 1x1 heatmap layer
                                           (head.py:525-532) to the ReLU'd tile: C is the f32 NCHW heat buffer [B, hm_K, hm_HW],
                                           final_w [hm_K, 256] bf16, final_b [hm_K] f32, hm_K <= 32; the 256-channel map is
                                           never stored.  out_rowmap gives each GEMM row its pixel index b * hm_HW + hw. */
#define PP_EPI_HEADMAJOR 4096            /* bf16 qkv projection (timm attn.qkv): C is written [3][heads][M][head_dim] instead of
                                           [M][3 * heads * head_dim]; heads = hm_K, head_dim = hm_HW (fields reused), N = 3 * hm_K *
                                           hm_HW.  Read by pp_attention_headmajor. */
#define PP_EPI_HEATMAP 64               /* head.py:526-532: f32 NCHW store of clamp(v / temperature, 0, 1):
                                           C[((r / hm_HW) * hm_K + n) * hm_HW + r % hm_HW], r = output row */

int pp_version(void);
const char *pp_last_error(void);
/* 1 when a gfx950 device is visible to this process, else 0 (never throws). */
int pp_device_ok(void);

/* ------------------------------------------------------------------------
 * Fused decode.  Replaces, in one launch per batch (every map that fits LDS, at batch sizes up to ~1 536 maps of 64x48 /
 * ~512 of 96x72; larger 64x48 / 96x72 batches take two launches -- one wave per map, then the few flat maps it hands
 * over -- and maps beyond LDS, e.g. 256x256, three launches through a global-memory image):
 *   Codec.decode            probpose/codec.py:249-263
 *   ProbMap.decode          probpose/codec.py:214-239
 *   get_heatmap_expected_value + _get_subpixel_maximums
 *                           probpose/heatmap.py:291-395, :114-167
 *   the D2H copy of util.to_numpy   probpose/util.py:6-12
 * (OKS kernels of heatmap.py:170-194 arrive pre-factored as normalised 1-D
 * taps; the 2-D kernel is exactly their outer product.)
 *
 * heatmaps [B,K,H,W] f32 contiguous.  prob/vis/oks/err: [B*K] f32 or NULL.
 * taps [K][PP_MAX_TAPS] f64 (entry j = offset j - radius[k]), radius [K].
 * den_x/den_y = heatmap_size-1 ([W-1,H-1] of codec.py:237), in_w/in_h = input_size.
 * Outputs (each may be NULL): kpts f64 [B,K,2] (input-image space),
 * scores f32 [B,K] (raw heatmap at the integer peak), locs f32 [B,K,2]
 * (heatmap space, = get_heatmap_expected_value's locs), aux f32 [3,B,K]
 * (prob,vis,oks passthrough), err f64 [B,K] (= err / sqrt(H^2+W^2)),
 * conv f32 [B,K,H,W] (return_heatmap=True), packed f64 [B,K,7] = (kpt x, kpt y,
 * score, prob, vis, oks, err) per keypoint: the record the multi-GPU all-gather ships.
 * workspace: pp_decode_workspace_bytes() bytes, 4-byte aligned: a (B*K + 2)-int hand-over list for 64x48 / 96x72 maps
 * (the wave-per-map path; NULL selects the workgroup-per-map kernels instead) which the caller ZEROES ONCE when it
 * allocates it (every call returns its two counters to zero itself: no memset or reset launch) and must not share
 * between calls that may run concurrently; 0 for other maps that fit in LDS; a float64 + float32 image of the batch for
 * maps that do not.
 * flags: 0 = the default form per map size and batch size (64x48 / 96x72: the all-pixel kernel up to ~1 536 / ~512
 * maps, the wave-per-map kernel above); PP_DECODE_NO_WAVE / PP_DECODE_SCREEN / PP_DECODE_ALL_PIXEL select the
 * other implementations (A/B measurements and the equivalence tests: all forms return identical numbers).
 * ---------------------------------------------------------------------- */
#define PP_DECODE_NO_WAVE 1             /* not the wave-per-map kernel (64x48 / 96x72 maps)                  */
#define PP_DECODE_SCREEN 2              /* the workgroup-per-map screened kernel on every map that fits LDS  */
#define PP_DECODE_ALL_PIXEL 4           /* float64 convolution of every pixel (the round-1 kernel)          */
#define PP_DECODE_WAVE 16               /* the wave-per-map path at any batch size (default: above two rounds of the
                                           all-pixel kernel's resident workgroups, ~1 536 maps of 64x48)      */
size_t pp_decode_workspace_bytes(int B, int K, int H, int W);
int pp_decode_f32(const float *heatmaps, const float *prob, const float *vis,
                  const float *oks, const float *err, int B, int K, int H, int W,
                  const double *taps, const int *radius, double den_x, double den_y,
                  double in_w, double in_h, double *out_kpts, float *out_scores,
                  float *out_locs, float *out_aux, double *out_err, float *out_conv,
                  double *out_packed, void *workspace, int flags, void *stream);

/* ------------------------------------------------------------------------
 * Dense contraction on MFMA:  C[M,N] = epilogue(A[M,Kd] * W[N,Kd]^T).
 * Replaces torch.nn.Linear / Conv2d-as-GEMM call sites of the backbone
 * (timm VisionTransformer via probpose/backbone.py:26-39: patch_embed.proj,
 * attn.qkv, attn.proj, mlp.fc1, mlp.fc2) and of the head (head.py:227-233
 * final_layer, :262-285 aux convs, :459-469 deconvs after im2col-free row
 * gather).  A rows are addressed through an optional gather table so 3x3
 * convolutions and the four stride-2 deconvolution parities run as implicit
 * GEMMs with no im2col buffer:
 *   A(m, k) = Abase[ rowoff[(k / seg_len) * M + m] + (k % seg_len) ]   (rowoff<0 -> 0)
 * with rowoff == NULL meaning A(m,k) = Abase[m*lda + k].
 * batch > 1 runs independent problems (strideA/W/C/bias in elements).
 * dtype selects storage+MFMA type for A, W and (unless PP_EPI_OUT_F32) C.
 * ---------------------------------------------------------------------- */
typedef struct pp_gemm_args {
  const void *A; const void *W; void *C;
  const float *bias;            /* [N] f32 or NULL                      */
  const float *residual;        /* [M,ldc] f32 (PP_EPI_RESIDUAL)        */
  const float *rowbias;         /* [rowbias_period, ldc] f32            */
  const int32_t *rowoff;        /* [segs*M] element offsets or NULL     */
  const int32_t *out_rowmap;    /* [M] output row of GEMM row m, or NULL (identity);
                                   the stride-2 deconvolution parities scatter through it */
  int M, N, Kd;
  int lda, ldw, ldc;
  int seg_len;                  /* K-segment length for the gather      */
  int rowbias_period;
  int batch;
  long long strideA, strideW, strideC, strideBias, strideRowoff, strideRowmap;
  int dtype;                    /* PP_F32 | PP_BF16 | PP_FP8 (A, W e4m3 bytes; C bf16, or f32 / e4m3 by flag;
                                   K % 128 == 0; colsum = [N] f32 dequantisation scale per output column:
                                   C = act((A8 W8^T)[m,n] * colsum[n] + bias[n]) (+ residual); tiles 2, 3) */
  int epilogue;                 /* PP_EPI_* flags                       */
  int hm_K, hm_HW;              /* PP_EPI_HEATMAP geometry              */
  float hm_temperature;         /* head.py:107 (0.5)                    */
  void *C2;                     /* reserved (retired LayerNorm fusion): must be NULL / 0; kept so the struct layout
                                   of round 1 binaries and bindings stays valid                          */
  int ldc2;
  float *stats_out;             /* reserved */
  const float *stats_in;        /* reserved */
  int stats_parts;              /* reserved */
  const float *colsum;          /* PP_FP8: [N] f32 column scales (batch stride strideBias)               */
  float ln_eps;                 /* reserved */
  int tile;                     /* 0 = auto (fewest rounds of resident workgroups), 1 = 128x128,
                                   2 = 192x96 (4 waves, 2 LDS stages), 3 = 192x192, 4 = 192x128 (8 waves, 3 stages), 5 = 384x128 (8 waves, 2 stages),
                                   6 = 192x192 wave-specialised (8 MFMA + 4 DMA waves, 3 stages),
                                   7 = 192x384, 8 = 256x256, 9 = 192x256 (8 waves, 2 stages; bf16),
                                   10 = 192x192 with the two wave quartets half a K-tile apart (one loads fragments and
                                   issues DMA while the other runs MFMAs; bf16 / fp8 plain layers),
                                   14 = 192x192 as TWO 4-wave workgroups per CU (96x96 wave tiles, 32-deep K-tiles, 72 KB
                                   of LDS each; plain bf16 layers with bias / GELU / ReLU / f32 residual).
                                   13 = persistent 192x192 stream (one workgroup per CU walks its tiles as one stream of
                                   K-tiles, the next tile's first K-tiles land under the epilogue; plain bf16 -> bf16
                                   layers with bias / GELU / ReLU, e.g. qkv and fc1).
                                   18 / 19 / 20 = four-wave stream forms (pp_gemm_quad.hip): 256x192 / 192x288 / 192x256
                                   tiles, one wave per SIMD with a 128x96 / 96x144 / 96x128 wave tile, one workgroup per CU
                                   walking its tiles as one stream of 32-deep K-tiles, a finished tile's rows stored from
                                   the next tile's K-loop; plain bf16 -> bf16 layers with bias / GELU / ReLU (and the
                                   head-major qkv layout), M and N whole numbers of tiles, K >= 512 (qkv, fc1).
                                   11, 12: round-2 experiments, removed (refused); 15 - 17: the per-launch four-wave forms,
                                   lab builds only (refused by the shipped library). */
  float out_scale;              /* PP_EPI_OUT_FP8: 1 / (scale of the fp8 output tensor) */
  int splitk;                   /* 0 / 1 = off.  S > 1: the launch computes S partial products per batch entry, split s
                                   over the K range [s * Kd, (s + 1) * Kd) (Kd = the PER-SPLIT depth): operands advance
                                   by the *_k strides below, the f32 partials go to C + s * strideC_k and are summed by
                                   the consumer (pp_maxpool_relu_sum).  Only with epilogue == PP_EPI_OUT_F32 (no bias,
                                   activation or residual); gather segments must not straddle a split (Kd % seg_len == 0).
                                   Used for the long-K, few-row aux convolutions (head.py:255-405 stages 2, 3). */
  long long strideA_k, strideW_k, strideC_k, strideRowoff_k;   /* elements per split step */
  const void *final_w;          /* PP_EPI_FUSE_FINAL */
  const float *final_b;
} pp_gemm_args;
int pp_gemm(const pp_gemm_args *args, void *stream);

/* LayerNorm over the last dim (timm Block.norm1/norm2/final norm, eps 1e-6).
 * x [rows,C] f32 (the fp32 residual stream) -> out [rows,C] in `dtype`. */
int pp_layernorm(const float *x, const float *gamma, const float *beta, float eps,
                 int rows, int C, void *out, int dtype, void *stream);
/* Same, output quantised to OCP e4m3: out[r,c] = e4m3(LN(x)[r,c] * inv_scale) (static per-tensor scale). */
int pp_layernorm_fp8(const float *x, const float *gamma, const float *beta, float eps,
                     int rows, int C, unsigned char *out, float inv_scale, void *stream);

/* Multi-head self-attention (timm Attention.forward: softmax(q k^T * hd^-1/2) v).
 * qkv [B*N, 3*heads*hd] laid out [3][heads][hd] along the row (timm's
 * reshape(B,N,3,heads,hd)); out [B*N, heads*hd]. */
int pp_attention(const void *qkv, void *out, int B, int N, int heads, int hd,
                 int dtype, void *stream);
/* Same attention on a HEAD-MAJOR bf16 qkv, [3][heads][B*N][hd] as the qkv GEMM writes it with PP_EPI_HEADMAJOR (one
 * head's rows contiguous: whole cache lines for head_dim 80 / 32); out [B*N, heads*hd] row-major as above.
 * The streaming MFMA kernel, head_dim 32 / 64 / 80, any N. */
int pp_attention_headmajor(const void *qkv, void *out, int B, int N, int heads, int hd, void *stream);
/* Same on bf16 qkv, output quantised to OCP e4m3: out[., c] = e4m3(o * inv_scale) (fp8 mode: the A operand of
 * the fp8 proj GEMM).  MFMA kernels only (head_dim 32 / 64 / 80). */
int pp_attention_fp8out(const void *qkv, unsigned char *out, int B, int N, int heads, int hd,
                        float inv_scale, void *stream);

/* Patch im2col + cast: x [B,3,H,W] f32 NCHW -> A [B*gh*gw, 3*p*p] (k = c*p*p + py*p + px),
 * the A operand of patch_embed.proj as a GEMM (timm PatchEmbed, stride = patch). */
int pp_patchify(const float *x, void *out, int B, int H, int W, int patch, int dtype,
                void *stream);

/* Split-K consumer of the aux convolutions: x = nsplit f32 partial maps [nsplit][B, h, w, C] (split stride
 * `split_stride` elements), bias [C] f32 -> out [B, h/kh, w/kw, C] in `dtype` = ReLU(MaxPool(sum_s x_s + bias))
 * (head.py:271-276 behind a conv whose K was split over workgroups). */
int pp_maxpool_relu_sum(const float *x, int nsplit, long long split_stride, const float *bias, void *out, int B, int h,
                        int w, int C, int kh, int kw, int dtype, void *stream);

/* MaxPool(kh,kw stride kh,kw) + ReLU on channels-last rows (head.py:271-276):
 * x [B, h, w, C] -> out [B, h/kh, w/kw, C]. */
int pp_maxpool_relu(const void *x, void *out, int B, int h, int w, int C, int kh, int kw,
                    int dtype, void *stream);

/* Final 1x1 conv + /temperature + clamp(0,1), channels-last rows -> NCHW float32 (head.py:525-532),
 * as a streaming (HBM-bound) kernel for small K:  x [B*HW, Cin], w [K, Cin], bias [K] ->
 * heat [B, K, HW] f32 = clamp((x w^T + bias) / temperature, 0, 1).  (K*Cin*sizeof + 64 rows must fit
 * LDS; larger final layers / k > 1 kernels go through pp_gemm with PP_EPI_HEATMAP.) */
int pp_final_heatmap(const void *x, const void *w, const float *bias, float *out, int B, int HW,
                     int Cin, int K, float temperature, int dtype, void *stream);

/* Same contraction without the clamp: logits = (x w^T + bias) / temperature, the input of the Sparsemax
 * normalisation (head.py:526-528 with normalize != None). */
int pp_final_logits(const void *x, const void *w, const float *bias, float *out, int B, int HW,
                    int Cin, int K, float temperature, int dtype, void *stream);

/* Sparsemax over the last axis of x [rows, n] f32 (in place), then * scale and clamp(0, 1): head.py:237-245
 * (normalize_layer = Sparsemax(dim=-1), third-party sparsemax==0.1.9: restated from the published algorithm,
 * Martins & Astudillo 2016 -- parity unpinned) + head.py:529-531 (x * normalize, clamp).  One workgroup per row. */
int pp_sparsemax_rows(float *x, long long rows, int n, float scale, void *stream);

/* ArgMaxProbMap.decode (codec.py:515-543): raw arg-max of every heatmap (heatmap.py:13-52 get_heatmap_maximum)
 * refined by DARK-UDP (codec.py:315-375: Gaussian blur codec.py:284-313, clip, log, 3x3 Hessian step), rescaled to
 * input pixels.  heatmaps [B,K,H,W] f32; taps_host = the ksize float32 coefficients of cv2.getGaussianKernel(ksize, 0)
 * (host memory, copied into the launch); outputs kpts [B,K,2] f64, scores [B,K] f32 (raw maxima), locs [B,K,2] f32
 * (integer arg-max, (-1,-1) where the maximum is <= 0).  cv2 is not importable here: parity unpinned. */
size_t pp_dark_decode_lds_bytes(int H, int W, int ksize);
int pp_dark_decode_f32(const float *heatmaps, int B, int K, int H, int W, const float *taps_host, int ksize,
                       double in_w, double in_h, double *out_kpts, float *out_scores, float *out_locs, void *stream);

/* Aux tail: 1x1 conv C->K on pooled 1x1 features + Sigmoid/ReLU (head.py:277-286,:391-400).
 * x [4 branches][B, C] -> out [4][B,K] f32 (branches 0..2 sigmoid, 3 relu). */
int pp_aux_tail(const void *x, const void *w, const float *bias, float *out, int B, int C,
                int K, int dtype, void *stream);

/* (B,N,C) tokens -> (B,C,gh,gw) f32, the permute of backbone.py:40. */
int pp_tokens_to_nchw(const void *x, float *out, int B, int N, int C, int dtype, void *stream);
/* (B,C,h,w) f32 NCHW -> (B,h*w,C) channels-last rows in `dtype` (head entry when
 * ProbMapHead.forward is called on a foreign NCHW feature map). */
int pp_nchw_to_tokens(const float *x, void *out, int B, int C, int HW, int dtype, void *stream);

/* ------------------------------------------------------------------------
 * Front end: person boxes of one RGB frame -> network input crops.  Replaces
 *   dataset.py:71-90   scale_box: image.crop(box).resize(image_size, PIL.Image.LANCZOS)
 *   inference.py:74-82 image.resize(input_size, LANCZOS), v2.ToImage(), v2.ToDtype(float32, scale=True)
 * (the arithmetic is Pillow's: Image.crop zero-pads outside the frame; ImagingResample, 8 bits per
 * channel, Lanczos-3, two passes with a uint8 intermediate, 22-bit fixed-point coefficients).
 * Bit-exact against Pillow.
 *
 * boxes_xyxy [n][4] int32: (x0, y0, x1, y1) AFTER Image.crop's rounding (int(round(v)) in Python).
 * pp_frontend_plan_bytes / pp_frontend_plan_build run on the HOST (no GPU needed): they compute, with
 * Pillow's own double-precision expressions, the per-box bounds / coefficient tables and the
 * workgroup table into `plan` (host memory); the caller copies `plan` to the device.
 * pp_frontend_crop_resize: image uint8 HWC RGB on the device (img_stride = bytes per row), plan_dev =
 * the device copy of the plan, out [n][3][out_h][out_w] f32 in [0,1] (= f32(u8) * f32(1/255)).
 * ---------------------------------------------------------------------- */
long long pp_frontend_plan_bytes(int n_boxes, const int *boxes_xyxy, int out_w, int out_h);   /* < 0: error */
int pp_frontend_plan_build(int n_boxes, const int *boxes_xyxy, int out_w, int out_h, void *plan,
                           int *n_blocks_out, long long *lds_bytes_out);
int pp_frontend_crop_resize(const unsigned char *image, int img_w, int img_h, long long img_stride,
                            const void *plan_dev, int n_boxes, int n_blocks, long long lds_bytes,
                            int out_w, int out_h, float *out, void *stream);

/* ------------------------------------------------------------------------
 * Training targets: the OKS probability maps of ProbMap.encode, batched over crops.  Replaces
 *   generate_probmaps   probpose/codec.py:11-70  (called from ProbMap.encode, codec.py:176-182)
 * kpts_hm [B,K,2] f32: keypoints in HEATMAP pixels (= keypoints / scale_factor, codec.py:178);
 * visible [B,K] f32; two_s [K] f64 = 2 * s_k with s_k the per-keypoint (or overriding) variance of
 * codec.py:61-66, computed by the caller in float64.  Outputs: heatmaps [B,K,H,W] f32 (zero for
 * visible < 0.5), weights [B,K] f32 (= visible for skipped keypoints, else map.max() > 0).
 * ---------------------------------------------------------------------- */
int pp_encode_probmaps(const float *kpts_hm, const float *visible, const double *two_s, int B, int K, int H,
                       int W, float *heatmaps, float *weights, void *stream);

/* ------------------------------------------------------------------------
 * Evaluation metric: PCK over a batch of keypoint pairs in one pass.  Replaces the per-instance host code
 *   _calc_distances        probpose/heatmap.py:55-89   (normalised distances, -1 for masked-out pairs)
 *   _distance_acc          probpose/heatmap.py:92-111  (fraction below the threshold)
 *   keypoint_pck_accuracy  probpose/loss.py:825-866    (called from pose_pck_accuracy, loss.py:767-822)
 * pred / gt [N,K,2] float32 (coord_f64 = 0) or float64 (coord_f64 = 1): the difference is formed in that type;
 * mask [N,K] u8; norm [N,2] f64 with the reference's "<= 0 -> 1e6" substitution already applied, skip[n] = 1 for an
 * instance whose normalisation factor held an exact 0 (heatmap.py:78-82); thr: the threshold as the comparison sees it
 * (a Python float compared with a float32 array is rounded to float32 by numpy 2).  Outputs: counts [2][K] int32 =
 * (pairs below thr, valid pairs) per keypoint (zeroed here by a memset node), dist [K,N] f32 or NULL.
 * ---------------------------------------------------------------------- */
/* get_heatmap_maximum (probpose/heatmap.py:13-52) for `maps` = B*K contiguous H x W float32 maps of any size: locs
 * [maps,2] f32 = (x, y) of the first maximum in row-major order (a NaN counts as the maximum, like np.argmax),
 * (-1,-1) where the maximum is <= 0; vals [maps] f32 = the maximum. */
int pp_heatmap_argmax(const float *heatmaps, long long maps, int H, int W, float *locs, float *vals, void *stream);
int pp_pck_counts(const void *pred, const void *gt, int coord_f64, const unsigned char *mask, const double *norm,
                  const unsigned char *skip, double thr, int N, int K, int *counts, float *dist, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PROBPOSE_HIP_H */
