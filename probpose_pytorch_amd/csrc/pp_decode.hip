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
  double *packed;   // [B*K][7] f64: x, y (input space), score, prob, vis, oks, err: the record the all-gather ships
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
  if (o.packed) {   // every field exactly as above, widened to f64 (lossless for the f32 ones)
    double *r = o.packed + (size_t)map * 7;
    r[0] = (double)fx / den_x * in_w;
    r[1] = (double)fy / den_y * in_h;
    r[2] = (double)raw_map[y * W + x];
    r[3] = prob ? (double)prob[map] : 0.0;
    r[4] = vis ? (double)vis[map] : 0.0;
    r[5] = oks ? (double)oks[map] : 0.0;
    r[6] = err ? (double)err[map] / sqrt((double)(H * H + W * W)) : 0.0;
  }
}

// ---------------------------------------------------------------------------
// LDS-resident path: 8 B (f64 row-pass intermediate) + 4 B (f32 map with a reflected 12-column halo,
// later reused for the f32 convolved map) of LDS per pixel.
// ---------------------------------------------------------------------------
#ifndef PP_DEC_THREADS
#define PP_DEC_THREADS 512   /* measured: 256 -> 512 threads per map: 64x48 maps +5 %, 96x72 maps +29 %; 1024: slower on 64x48 */
#endif
constexpr int DEC_THREADS = PP_DEC_THREADS;

// Diagnostic build only (-DPP_DEC_STAMPS): wave-0 phase cycle counts are written through out_conv.
#ifdef PP_DEC_STAMPS
__device__ __forceinline__ unsigned long long dec_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define DEC_STAMP(v) const unsigned long long v = dec_stamp()
#else
#define DEC_STAMP(v)
#endif

// Row pass + column pass for a compile-time radius R.  Each thread produces 4 adjacent outputs per
// item from one shared window of 4 + 2R inputs (3.5x fewer LDS reads than one output per thread at
// R = 9) with 4 independent float64 FMA chains; per output the taps are still accumulated in
// ascending order, so the result does not depend on the blocking.
constexpr int DEC_HALO = 12;  // reflected halo columns on each side of a row in LDS: 4 * ceil(9 / 4)

// Row pass + column pass for a compile-time radius R.  Each thread produces 4 adjacent outputs per
// item from one shared window of 4 + 2R inputs with 4 independent float64 FMA chains; per output the
// taps are accumulated in ascending order, so the result does not depend on the blocking.
// The raw map sits in LDS with a reflected halo (rawp, row stride WP), so every row window is read
// with 16-B aligned, lane-consecutive ds_read_b128: the stride-4 scalar reads this replaces were an
// 8-way LDS bank conflict and made the row pass LDS-bound.
template <int R>
__device__ __forceinline__ void conv_passes(const float *__restrict__ rawp, int WP, float *__restrict__ buf,
                                            double *__restrict__ tmp, int H, int W,
                                            const double (&wk)[PP_MAX_TAPS], float *__restrict__ out_conv_map,
                                            float &best_v, int &best_i, bool &have,
                                            unsigned long long *dbg = nullptr) {
  constexpr int T = 2 * R + 1, WIN = T + 3;
  constexpr int CR = (R + 3) / 4, NCH = 2 * CR + 1, SKIP = 4 * CR - R;  // aligned chunks around the window
  const int tid = threadIdx.x;
  double w[T];
#pragma unroll
  for (int j = 0; j < T; ++j) w[j] = wk[j];  // block-uniform -> scalar registers

  // one division per thread up front, then (row, column-group) advances incrementally by DEC_THREADS items
  const int W4 = (W + 3) >> 2;
  const int step_y = DEC_THREADS / W4, step_x = DEC_THREADS - step_y * W4;
  {
    int y = tid / W4, xg = tid - y * W4;
    for (; y < H;) {
      const int x0 = xg * 4;
      const float4 *rp = reinterpret_cast<const float4 *>(rawp + y * WP + x0 + DEC_HALO - 4 * CR);
      float c[NCH * 4];
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        const float4 t = rp[q];
        c[4 * q + 0] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
      }
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        a0 = fma(w[j], (double)c[SKIP + j], a0);
        a1 = fma(w[j], (double)c[SKIP + j + 1], a1);
        a2 = fma(w[j], (double)c[SKIP + j + 2], a2);
        a3 = fma(w[j], (double)c[SKIP + j + 3], a3);
        // keep the four chains interleaved: hipcc otherwise runs them one after another and every
        // v_fmac_f64 then waits out the previous one's latency (3-4 waves per SIMD cannot hide it)
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
      double *o = tmp + y * W + x0;
      if (x0 + 3 < W) {
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
      } else {
        o[0] = a0;
        if (x0 + 1 < W) o[1] = a1;
        if (x0 + 2 < W) o[2] = a2;
      }
      y += step_y;
      xg += step_x;
      if (xg >= W4) {
        xg -= W4;
        ++y;
      }
    }
  }
#ifdef PP_DEC_STAMPS
  if (dbg) dbg[0] = dec_stamp();
#endif
  __syncthreads();
#ifdef PP_DEC_STAMPS
  if (dbg) dbg[1] = dec_stamp();
#endif

  const int H4 = (H + 3) >> 2;
  const int cstep_y = DEC_THREADS / W, cstep_x = DEC_THREADS - cstep_y * W;
  const bool one_reflection_y = H >= R + 3;
  {
    int y4 = tid / W, x = tid - y4 * W;
    for (; y4 < H4;) {
      const int y0 = y4 * 4;
      double v[WIN];
      if (y0 >= R && y0 + 3 + R < H) {
#pragma unroll
        for (int j = 0; j < WIN; ++j) v[j] = tmp[(y0 - R + j) * W + x];
      } else if (one_reflection_y) {
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
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (y0 + i < H) {
          const int p = (y0 + i) * W + x;
          const float cv = (float)a[i];
          buf[p] = cv;  // the raw copy is dead after the row pass
          if (out_conv_map) out_conv_map[p] = cv;
          // rows ascend, so within a thread a later pixel always has the larger flat index: a strict
          // '>' keeps the first maximum; NaN handling stays in better()
          if (!have || better(cv, p, best_v, best_i)) {
            best_v = cv;
            best_i = p;
            have = true;
          }
        }
      }
      y4 += cstep_y;
      x += cstep_x;
      if (x >= W) {
        x -= W;
        ++y4;
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
  const int WP = 4 * ((W + 3) >> 2) + 2 * DEC_HALO;              // padded row: [halo | W (+pad to 4) | halo]
  double *tmp = reinterpret_cast<double *>(smem);               // [HW] f64 row-pass result
  float *rawp = reinterpret_cast<float *>(smem + (size_t)HW * 8);  // [H][WP] f32 raw map + reflected halo
  float *buf = rawp;                                             // [HW] f32 convolved map (raw is dead by then)
  __shared__ Best red[DEC_THREADS / 64];

  const int map = blockIdx.x;
  const int k = map % K;
  const float *__restrict__ src = heatmaps + (size_t)map * HW;
  const int tid = threadIdx.x;

  DEC_STAMP(t0);
  // taps and radius are block-uniform: scalar loads issued now, their latency hides under the map load
  const int r = __builtin_amdgcn_readfirstlane(radius[k]);
  double wk[PP_MAX_TAPS];
#pragma unroll
  for (int j = 0; j < PP_MAX_TAPS; ++j) wk[j] = taps[k * PP_MAX_TAPS + j];
  // 1. HBM -> LDS.  Interior: 16 B per lane when rows are 16-B aligned; halo cells: scipy 'reflect'
  // of the same rows (their cache lines are already on the way).
  if ((W & 3) == 0) {
    const int W4 = W >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    const int sy = DEC_THREADS / W4, sx = DEC_THREADS - sy * W4;
    int y = tid / W4, xq = tid - y * W4;
    for (int p = tid; p < H * W4; p += DEC_THREADS) {
      *reinterpret_cast<float4 *>(rawp + y * WP + DEC_HALO + 4 * xq) = s4[p];
      y += sy;
      xq += sx;
      if (xq >= W4) {
        xq -= W4;
        ++y;
      }
    }
  } else {
    for (int p = tid; p < HW; p += DEC_THREADS) {
      const int y = p / W, x = p - y * W;
      rawp[y * WP + DEC_HALO + x] = src[p];
    }
  }
  {
    const int npad = WP - W;  // halo + alignment cells per row, 24..27: one 32-lane group per row
    const int q = tid & 31;
    if (q < npad) {
      const int col = q < DEC_HALO ? q : q + W;       // padded column index
      const int sx = reflect_idx(col - DEC_HALO, W);
      for (int y = tid >> 5; y < H; y += DEC_THREADS / 32) rawp[y * WP + col] = src[y * W + sx];
    }
  }
  __syncthreads();
  DEC_STAMP(t1);

  // 2+3. separable float64 convolution -> float32 map in LDS + per-thread running argmax
  Best mine;
  mine.v = -__builtin_inff();
  mine.i = 0x7fffffff;
  bool have = false;
  float *ocm = out_conv ? out_conv + (size_t)map * HW : nullptr;
  unsigned long long *dbgp = nullptr;
#ifdef PP_DEC_STAMPS
  unsigned long long dbg_st[2] = {0, 0};
  dbgp = dbg_st;
#endif
  switch (r) {  // block-uniform
    case 2: conv_passes<2>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 3: conv_passes<3>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 4: conv_passes<4>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 5: conv_passes<5>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 6: conv_passes<6>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 7: conv_passes<7>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 8: conv_passes<8>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    default: conv_passes<9>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
  }
  if (!have) {  // more threads than pixels: never wins
    mine.v = -__builtin_inff();
    mine.i = 0x7fffffff;
  }
  DEC_STAMP(t2);
  __syncthreads();
  const Best b = block_argmax(mine, red);
  DEC_STAMP(t3);
  if (tid == 0) {
    auto at = [&](int yy, int xx) { return buf[yy * W + xx]; };
    finalize(map, B, K, H, W, b.i, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
  }
#ifdef PP_DEC_STAMPS
  {
    DEC_STAMP(t4);
    if (tid == 0 && out_conv == nullptr) {
      unsigned long long *d = reinterpret_cast<unsigned long long *>(o.locs + (size_t)2 * B * K) + (size_t)map * 8;
      d[0] = t1 - t0; d[1] = t2 - t1; d[2] = t3 - t2; d[3] = t4 - t3; d[4] = t4 - t0; d[5] = t0; d[6] = r;
      d[7] = ((dbg_st[0] - t1) << 32) | ((dbg_st[1] - dbg_st[0]) & 0xffffffffull);
    }
  }
#endif
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

static size_t lds_bytes(int H, int W) {
  return (size_t)H * W * 8 + (size_t)H * (4 * ((W + 3) / 4) + 2 * DEC_HALO) * 4;
}
static bool fits_lds(int H, int W) { return lds_bytes(H, W) <= LDS_LIMIT; }

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
                             double *out_packed, void *workspace, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, "pp_decode_f32: bad shape B=%d K=%d H=%d W=%d", B, K,
             H, W);
  if (B == 0) return 0;  // empty batch: nothing to launch (buffers may be null)
  PP_REQUIRE(heatmaps && taps && radius, "pp_decode_f32: null input");
  PP_REQUIRE((long long)H * W < (1ll << 30), "pp_decode_f32: map too large");
  hipStream_t s = (hipStream_t)stream;
  DecodeOut o{out_kpts, out_scores, out_locs, out_aux, out_err, out_packed};
  const int maps = B * K;
  if (fits_lds(H, W)) {
    const size_t lds = lds_bytes(H, W);
    static thread_local unsigned long long attr_mask = 0;
    int dev_ = 0;
    if (lds > 64 * 1024 && attr_needed(attr_mask, dev_))
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(decode_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
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
