// Fused ProbPose decode for gfx950: OKS-Gaussian convolution (scipy 'reflect'
// semantics) -> first-index argmax -> 1-D quadratic sub-pixel shift per axis ->
// raw-score gather -> rescale to input pixels, one workgroup per (crop,keypoint)
// map, the map read from HBM exactly once.
//
// Arithmetic follows the reference bit-for-bit where it is float32 and to the
// last double ulp where the reference accumulates in float64:
//   * scipy.ndimage.convolve accumulates in double and rounds the sum to
//     float32 (probpose/heatmap.py:361-364).  The OKS kernel is an isotropic
//     Gaussian normalised over a square window (heatmap.py:184-189), i.e. the
//     outer product of two normalised 1-D Gaussians, so the kernel runs as a
//     row pass + column pass, both accumulated in float64 and rounded to
//     float32 once, exactly where the reference rounds.
//   * argmax = np.argmax (first maximum in row-major order; NaN wins)
//     (heatmap.py:366-369).
//   * sub-pixel = float32 finite differences on the float32 convolved map
//     (heatmap.py:114-167); compiled with -ffp-contract=off so no FMA fuses
//     what numpy evaluates as separate float32 operations.
//   * rescale in float64: locs / (size-1) * input_size (codec.py:237).
#include "pp_common.h"

namespace pp {

// scipy 'reflect' (half-sample symmetric, d c b a | a b c d | d c b a), any offset.
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i >= 0 && i < n) return i;
  int p = 2 * n;
  int m = i % p;
  if (m < 0) m += p;
  return m < n ? m : p - 1 - m;
}

// single reflection, valid while -n <= i < 2n (always true for |offset| <= radius <= n): no division
__device__ __forceinline__ int reflect_once(int i, int n) {
  return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i);
}

struct Best {
  float v;
  int i;
};

// a strictly better than b under np.argmax semantics
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) {
  bool an = av != av, bn = bv != bv;
  if (an || bn) {
    if (an && bn) return ai < bi;
    return an;
  }
  return av > bv || (av == bv && ai < bi);
}

__device__ __forceinline__ Best block_argmax(Best mine, Best *red /* [waves] in LDS */) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_down(mine.v, o, 64);
    int oi = __shfl_down(mine.i, o, 64);
    if (better(ov, oi, mine.v, mine.i)) {
      mine.v = ov;
      mine.i = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) red[wave] = mine;
  __syncthreads();
  Best r = red[0];
  for (int w = 1; w < nw; ++w)
    if (better(red[w].v, red[w].i, r.v, r.i)) r = red[w];
  return r;
}

struct DecodeOut {
  double *kpts;
  float *scores;
  float *locs;
  float *aux;
  double *err;
};

// Everything after the argmax; one thread.  c(y,x) reads the float32 convolved map.
template <typename ConvAt>
__device__ __forceinline__ void finalize(int map, int B, int K, int H, int W, int best_idx,
                                         ConvAt c, const float *__restrict__ raw_map,
                                         const float *prob, const float *vis, const float *oks,
                                         const float *err, double den_x, double den_y,
                                         double in_w, double in_h, DecodeOut o) {
  const int x = best_idx % W, y = best_idx / W;
  float fx = (float)x, fy = (float)y;
  if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {  // heatmap.py:120-125
    const float cc = c(y, x), xp = c(y, x + 1), xm = c(y, x - 1), yp = c(y + 1, x),
                ym = c(y - 1, x);
    const float dx = (xp - xm) / 2.0f;
    const float dy = (yp - ym) / 2.0f;
    float dxx = (xp + xm) - 2.0f * cc;
    float dyy = (yp + ym) - 2.0f * cc;
    if (!(dxx != 0.0f)) dxx = 1e-6f;  // np.where(dxx != 0, dxx, 1e-6): NaN != 0 is true
    if (!(dyy != 0.0f)) dyy = 1e-6f;
    fx = fx + (-dx / dxx);
    fy = fy + (-dy / dyy);
  }
  if (o.locs) {
    o.locs[2 * map + 0] = fx;
    o.locs[2 * map + 1] = fy;
  }
  if (o.kpts) {
    o.kpts[2 * map + 0] = (double)fx / den_x * in_w;  // codec.py:237
    o.kpts[2 * map + 1] = (double)fy / den_y * in_h;
  }
  if (o.scores) o.scores[map] = raw_map[y * W + x];  // heatmap.py:375-379
  const int BK = B * K;
  if (o.aux) {
    if (prob) o.aux[map] = prob[map];
    if (vis) o.aux[BK + map] = vis[map];
    if (oks) o.aux[2 * BK + map] = oks[map];
  }
  if (o.err && err)  // codec.py:260-261: float32 / np.float64 scalar -> float64
    o.err[map] = (double)err[map] / sqrt((double)(H * H + W * W));
}

// ---------------------------------------------------------------------------
// LDS-resident path: 12 B of LDS per pixel (f32 map, then reused for the f32
// convolved map; f64 row-pass intermediate).
// ---------------------------------------------------------------------------
constexpr int DEC_THREADS = 256;

// Row pass + column pass for a compile-time radius R.  Each thread produces 4 adjacent outputs per
// item from one shared window of 4 + 2R inputs (3.5x fewer LDS reads than one output per thread at
// R = 9) with 4 independent float64 FMA chains; per output the taps are still accumulated in
// ascending order, so the result does not depend on the blocking.
template <int R>
__device__ __forceinline__ void conv_passes(float *__restrict__ buf, double *__restrict__ tmp, int H,
                                            int W, const double *__restrict__ wk,
                                            float *__restrict__ out_conv_map, float &best_v,
                                            int &best_i, bool &have) {
  constexpr int T = 2 * R + 1, WIN = T + 3;
  const int tid = threadIdx.x;
  double w[T];
#pragma unroll
  for (int j = 0; j < T; ++j) w[j] = wk[j];  // block-uniform -> scalar registers

  const int W4 = (W + 3) >> 2;
  for (int item = tid; item < H * W4; item += DEC_THREADS) {
    const int y = item / W4, x0 = (item - y * W4) * 4;
    const float *row = buf + y * W;
    double v[WIN];
    if (x0 >= R && x0 + 3 + R < W) {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = (double)row[x0 - R + j];
    } else if (W >= R + 3) {   // one reflection suffices (offsets reach at most R + 3 past an edge)
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = (double)row[reflect_once(min(x0 - R + j, 2 * W - 1), W)];
    } else {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = (double)row[reflect_idx(x0 - R + j, W)];
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      a0 = fma(w[j], v[j], a0);
      a1 = fma(w[j], v[j + 1], a1);
      a2 = fma(w[j], v[j + 2], a2);
      a3 = fma(w[j], v[j + 3], a3);
    }
    double *o = tmp + y * W + x0;
    o[0] = a0;
    if (x0 + 1 < W) o[1] = a1;
    if (x0 + 2 < W) o[2] = a2;
    if (x0 + 3 < W) o[3] = a3;
  }
  __syncthreads();

  const int H4 = (H + 3) >> 2;
  for (int item = tid; item < H4 * W; item += DEC_THREADS) {
    const int y4 = item / W, x = item - y4 * W, y0 = y4 * 4;
    double v[WIN];
    if (y0 >= R && y0 + 3 + R < H) {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = tmp[(y0 - R + j) * W + x];
    } else if (H >= R + 3) {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = tmp[reflect_once(min(y0 - R + j, 2 * H - 1), H) * W + x];
    } else {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = tmp[reflect_idx(y0 - R + j, H) * W + x];
    }
    double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < T; ++j) {
      a[0] = fma(w[j], v[j], a[0]);
      a[1] = fma(w[j], v[j + 1], a[1]);
      a[2] = fma(w[j], v[j + 2], a[2]);
      a[3] = fma(w[j], v[j + 3], a[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (y0 + i < H) {
        const int p = (y0 + i) * W + x;
        const float cv = (float)a[i];
        buf[p] = cv;  // the raw copy is dead after the row pass
        if (out_conv_map) out_conv_map[p] = cv;
        if (!have || better(cv, p, best_v, best_i)) {
          best_v = cv;
          best_i = p;
          have = true;
        }
      }
    }
  }
}

__global__ __launch_bounds__(DEC_THREADS) void decode_lds_kernel(
    const float *__restrict__ heatmaps, const float *prob, const float *vis, const float *oks,
    const float *err, int B, int K, int H, int W, const double *__restrict__ taps,
    const int *__restrict__ radius, double den_x, double den_y, double in_w, double in_h,
    DecodeOut o, float *__restrict__ out_conv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HW = H * W;
  double *tmp = reinterpret_cast<double *>(smem);               // [HW] f64
  float *buf = reinterpret_cast<float *>(smem + (size_t)HW * 8);  // [HW] f32: raw, then conv
  __shared__ Best red[DEC_THREADS / 64];

  const int map = blockIdx.x;
  const int k = map % K;
  const float *__restrict__ src = heatmaps + (size_t)map * HW;
  const int tid = threadIdx.x;

  // 1. HBM -> LDS, 16 B per lane when the map allows it
  if ((HW & 3) == 0) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(buf);
    for (int p = tid; p < (HW >> 2); p += DEC_THREADS) d4[p] = s4[p];
  } else {
    for (int p = tid; p < HW; p += DEC_THREADS) buf[p] = src[p];
  }
  const int r = __builtin_amdgcn_readfirstlane(radius[k]);
  const double *wk = taps + k * PP_MAX_TAPS;
  __syncthreads();

  // 2+3. separable float64 convolution -> float32 map in LDS + per-thread running argmax
  Best mine;
  mine.v = -__builtin_inff();
  mine.i = 0x7fffffff;
  bool have = false;
  float *ocm = out_conv ? out_conv + (size_t)map * HW : nullptr;
  switch (r) {  // block-uniform
    case 2: conv_passes<2>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 3: conv_passes<3>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 4: conv_passes<4>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 5: conv_passes<5>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 6: conv_passes<6>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 7: conv_passes<7>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    case 8: conv_passes<8>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
    default: conv_passes<9>(buf, tmp, H, W, wk, ocm, mine.v, mine.i, have); break;
  }
  if (!have) {  // more threads than pixels: never wins
    mine.v = -__builtin_inff();
    mine.i = 0x7fffffff;
  }
  __syncthreads();
  const Best b = block_argmax(mine, red);
  if (tid == 0) {
    auto at = [&](int yy, int xx) { return buf[yy * W + xx]; };
    finalize(map, B, K, H, W, b.i, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
  }
}

// ---------------------------------------------------------------------------
// Large-map path (map does not fit in LDS, e.g. the reference's own 256x256
// test, tests/test_heatmap.py:6): three passes through a global workspace.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_rowpass_kernel(
    const float *__restrict__ heatmaps, double *__restrict__ tmp, int K, int H, int W,
    const double *__restrict__ taps, const int *__restrict__ radius) {
  const int map = blockIdx.x, k = map % K, HW = H * W;
  const int r = radius[k];
  const double *w = taps + k * PP_MAX_TAPS;
  const float *src = heatmaps + (size_t)map * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) {
    const int y = p / W, x = p - y * W;
    double acc = 0.0;
    for (int j = 0; j <= 2 * r; ++j)
      acc = fma(w[j], (double)src[y * W + reflect_idx(x - r + j, W)], acc);
    tmp[(size_t)map * HW + p] = acc;
  }
}

__global__ __launch_bounds__(256) void decode_colpass_kernel(
    const double *__restrict__ tmp, float *__restrict__ conv, int K, int H, int W,
    const double *__restrict__ taps, const int *__restrict__ radius) {
  const int map = blockIdx.x, k = map % K, HW = H * W;
  const int r = radius[k];
  const double *w = taps + k * PP_MAX_TAPS;
  const double *src = tmp + (size_t)map * HW;
  for (int p = blockIdx.y * 256 + threadIdx.x; p < HW; p += gridDim.y * 256) {
    const int y = p / W, x = p - y * W;
    double acc = 0.0;
    for (int j = 0; j <= 2 * r; ++j) acc = fma(w[j], src[reflect_idx(y - r + j, H) * W + x], acc);
    conv[(size_t)map * HW + p] = (float)acc;
  }
}

__global__ __launch_bounds__(DEC_THREADS) void decode_argmax_kernel(
    const float *__restrict__ heatmaps, const float *__restrict__ conv, const float *prob,
    const float *vis, const float *oks, const float *err, int B, int K, int H, int W,
    double den_x, double den_y, double in_w, double in_h, DecodeOut o) {
  __shared__ Best red[DEC_THREADS / 64];
  const int map = blockIdx.x, HW = H * W, tid = threadIdx.x;
  const float *c = conv + (size_t)map * HW;
  Best mine;
  mine.v = -__builtin_inff();
  mine.i = 0x7fffffff;
  bool have = false;
  for (int p = tid; p < HW; p += DEC_THREADS) {
    const float cv = c[p];
    if (!have || better(cv, p, mine.v, mine.i)) {
      mine.v = cv;
      mine.i = p;
      have = true;
    }
  }
  const Best b = block_argmax(mine, red);
  if (tid == 0) {
    auto at = [&](int yy, int xx) { return c[yy * W + xx]; };
    finalize(map, B, K, H, W, b.i, at, heatmaps + (size_t)map * HW, prob, vis, oks, err, den_x,
             den_y, in_w, in_h, o);
  }
}

constexpr size_t LDS_LIMIT = 160 * 1024 - 256;

static bool fits_lds(int H, int W) { return (size_t)H * W * 12 <= LDS_LIMIT; }

}  // namespace pp

extern "C" size_t pp_decode_workspace_bytes(int B, int K, int H, int W) {
  if (pp::fits_lds(H, W)) return 0;
  return (size_t)B * K * H * W * (sizeof(double) + sizeof(float));
}

extern "C" int pp_decode_f32(const float *heatmaps, const float *prob, const float *vis,
                             const float *oks, const float *err, int B, int K, int H, int W,
                             const double *taps, const int *radius, double den_x, double den_y,
                             double in_w, double in_h, double *out_kpts, float *out_scores,
                             float *out_locs, float *out_aux, double *out_err, float *out_conv,
                             void *workspace, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, "pp_decode_f32: bad shape B=%d K=%d H=%d W=%d", B, K,
             H, W);
  if (B == 0) return 0;  // empty batch: nothing to launch (buffers may be null)
  PP_REQUIRE(heatmaps && taps && radius, "pp_decode_f32: null input");
  PP_REQUIRE((long long)H * W < (1ll << 30), "pp_decode_f32: map too large");
  hipStream_t s = (hipStream_t)stream;
  DecodeOut o{out_kpts, out_scores, out_locs, out_aux, out_err};
  const int maps = B * K;
  if (fits_lds(H, W)) {
    const size_t lds = (size_t)H * W * 12;
    static thread_local size_t attr_set = 0;
    if (lds > 64 * 1024 && lds > attr_set) {
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(decode_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
      attr_set = LDS_LIMIT;
    }
    hipLaunchKernelGGL(decode_lds_kernel, dim3(maps), dim3(DEC_THREADS), lds, s, heatmaps, prob, vis,
                       oks, err, B, K, H, W, taps, radius, den_x, den_y, in_w, in_h, o, out_conv);
    PP_CHECK_LAUNCH("decode_lds_kernel");
    return 0;
  }
  PP_REQUIRE(workspace, "pp_decode_f32: map %dx%d needs a workspace (pp_decode_workspace_bytes)", H,
             W);
  double *tmp = reinterpret_cast<double *>(workspace);
  float *conv = out_conv ? out_conv : reinterpret_cast<float *>(tmp + (size_t)maps * H * W);
  const int bx = cdiv((long long)H * W, 256 * 4) > 65535 ? 65535 : cdiv((long long)H * W, 256 * 4);
  hipLaunchKernelGGL(decode_rowpass_kernel, dim3(maps, bx), dim3(256), 0, s, heatmaps, tmp, K, H, W,
                     taps, radius);
  PP_CHECK_LAUNCH("decode_rowpass_kernel");
  hipLaunchKernelGGL(decode_colpass_kernel, dim3(maps, bx), dim3(256), 0, s, tmp, conv, K, H, W, taps,
                     radius);
  PP_CHECK_LAUNCH("decode_colpass_kernel");
  hipLaunchKernelGGL(decode_argmax_kernel, dim3(maps), dim3(DEC_THREADS), 0, s, heatmaps, conv, prob,
                     vis, oks, err, B, K, H, W, den_x, den_y, in_w, in_h, o);
  PP_CHECK_LAUNCH("decode_argmax_kernel");
  return 0;
}
