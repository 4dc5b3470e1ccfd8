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
#include <algorithm>

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
template <int R, int NT>   // NT = threads of the workgroup
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

  // one division per thread up front, then (row, column-group) advances incrementally by NT items
  const int W4 = (W + 3) >> 2;
  const int step_y = NT / W4, step_x = NT - step_y * W4;
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
  const int cstep_y = NT / W, cstep_x = NT - cstep_y * W;
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

template <int NT>
__device__ __forceinline__ void decode_lds_map(
    const int map, char *smem, Best *red, const float *__restrict__ heatmaps, const float *prob, const float *vis,
    const float *oks, const float *err, int B, int K, int H, int W, const double *__restrict__ taps,
    const int *__restrict__ radius, double den_x, double den_y, double in_w, double in_h,
    DecodeOut o, float *__restrict__ out_conv) {
  const int HW = H * W;
  const int WP = 4 * ((W + 3) >> 2) + 2 * DEC_HALO;              // padded row: [halo | W (+pad to 4) | halo]
  double *tmp = reinterpret_cast<double *>(smem);               // [HW] f64 row-pass result
  float *rawp = reinterpret_cast<float *>(smem + (size_t)HW * 8);  // [H][WP] f32 raw map + reflected halo
  float *buf = rawp;                                             // [HW] f32 convolved map (raw is dead by then)

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
    const int sy = NT / W4, sx = NT - sy * W4;
    int y = tid / W4, xq = tid - y * W4;
    for (int p = tid; p < H * W4; p += NT) {
      *reinterpret_cast<float4 *>(rawp + y * WP + DEC_HALO + 4 * xq) = s4[p];
      y += sy;
      xq += sx;
      if (xq >= W4) {
        xq -= W4;
        ++y;
      }
    }
  } else {
    for (int p = tid; p < HW; p += NT) {
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
      for (int y = tid >> 5; y < H; y += NT / 32) rawp[y * WP + col] = src[y * W + sx];
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
    case 2: conv_passes<2, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 3: conv_passes<3, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 4: conv_passes<4, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 5: conv_passes<5, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 6: conv_passes<6, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 7: conv_passes<7, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    case 8: conv_passes<8, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
    default: conv_passes<9, NT>(rawp, WP, buf, tmp, H, W, wk, ocm, mine.v, mine.i, have, dbgp); break;
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

__global__ __launch_bounds__(DEC_THREADS) void decode_lds_kernel(
    const float *__restrict__ heatmaps, const float *prob, const float *vis, const float *oks,
    const float *err, int B, int K, int H, int W, const double *__restrict__ taps,
    const int *__restrict__ radius, double den_x, double den_y, double in_w, double in_h,
    DecodeOut o, float *__restrict__ out_conv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ Best red[DEC_THREADS / 64];
  decode_lds_map<DEC_THREADS>(blockIdx.x, smem, red, heatmaps, prob, vis, oks, err, B, K, H, W, taps, radius, den_x,
                              den_y, in_w, in_h, o, out_conv);
}

// ---------------------------------------------------------------------------
// Screened path (the default when the convolved maps themselves are not requested): the result of the decode is
// decided by a handful of pixels -- the arg-max of the convolved map and its four neighbours -- so the float64
// convolution is evaluated ONLY there.
//   1. the map goes HBM -> LDS once (f32, reflected halo), A = max |x| comes with it;
//   2. the separable convolution runs over every pixel in float32 (row pass into a 4 B/pixel buffer, column pass
//      on the fly), giving c32 with |c32 - c*| <= 41 u A of the reference value c* (u = 2^-24; the kernel is
//      normalised, so every partial sum is bounded by A: (T + 1) u A per pass with T <= 19 taps, + u A for the
//      reference's own rounding to float32);
//   3. every pixel with c32 >= max(c32) - 128 u A is a candidate (contains every arg-max of c*: c* of a maximum is
//      >= c*max, so its c32 >= c*max - 41 u A >= max(c32) - 82 u A);
//   4. one wave per candidate evaluates c* exactly as the reference does -- row sums and the column sum as float64
//      FMA chains over ascending taps, rounded to float32 once (bit-identical to decode_lds_kernel and, on every
//      golden, to scipy) -- then the first-index arg-max over the candidates' exact values, the exact values of its
//      four neighbours, and the float32 sub-pixel arithmetic of finalize().
// LDS: 8 B per pixel (+ halo) instead of 12, 256 threads per map instead of 512: 5 workgroups per CU on 64x48 maps
// (was 3), 2 on 96x72 (was 1).  Maps with non-finite values, or flat maps with more candidates than the list holds,
// take the exact evaluation for every pixel that passes the predicate (slow, rare, still exact).
// ---------------------------------------------------------------------------
constexpr int DF_THREADS = 256;
constexpr int DF_MAXCAND = 512;


typedef float f32x2 __attribute__((ext_vector_type(2)));

// Float32 screening passes for a compile-time radius.  Row pass: 4 adjacent outputs per item from one aligned window
// (two packed-f32 FMA chains); column pass: 4 vertically adjacent outputs per item.  The column-pass results stay in
// registers (keep[][]), so the candidate test after the block-wide maximum needs no second pass.
template <int R, int DF_KEEP>
__device__ __forceinline__ void screen_passes(const float *__restrict__ rawp, int WP, float *__restrict__ tmp32, int H,
                                              int W, const float (&w)[PP_MAX_TAPS], float (&keep)[DF_KEEP][4],
                                              float &best_v) {
  constexpr int T = 2 * R + 1, WIN = T + 3;
  constexpr int CR = (R + 3) / 4, NCH = 2 * CR + 1, SKIP = 4 * CR - R;
  const int tid = threadIdx.x;
  const int W4 = (W + 3) >> 2;
  for (int it = tid; it < H * W4; it += DF_THREADS) {
    const int y = it / W4, x0 = (it - y * W4) * 4;
    const float4 *rp = reinterpret_cast<const float4 *>(rawp + y * WP + x0 + DEC_HALO - 4 * CR);
    float c[NCH * 4];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const float4 t = rp[q];
      c[4 * q + 0] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
    }
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const f32x2 wj = {w[j], w[j]};
      a01 = __builtin_elementwise_fma(wj, (f32x2){c[SKIP + j], c[SKIP + j + 1]}, a01);
      a23 = __builtin_elementwise_fma(wj, (f32x2){c[SKIP + j + 2], c[SKIP + j + 3]}, a23);
    }
    float *o = tmp32 + y * W + x0;
    o[0] = a01.x;
    if (x0 + 1 < W) o[1] = a01.y;
    if (x0 + 2 < W) o[2] = a23.x;
    if (x0 + 3 < W) o[3] = a23.y;
  }
  __syncthreads();
  const int H4 = (H + 3) >> 2;
#pragma unroll
  for (int u = 0; u < DF_KEEP; ++u) {
    const int it = tid + u * DF_THREADS;
    keep[u][0] = keep[u][1] = keep[u][2] = keep[u][3] = -__builtin_inff();
    if (it >= H4 * W) continue;
    const int y4 = it / W, x = it - y4 * W, y0 = y4 * 4;
    float v[WIN];
    if (y0 >= R && y0 + 3 + R < H) {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = tmp32[(y0 - R + j) * W + x];
    } else {
#pragma unroll
      for (int j = 0; j < WIN; ++j) v[j] = tmp32[reflect_idx(y0 - R + j, H) * W + x];
    }
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const f32x2 wj = {w[j], w[j]};
      a01 = __builtin_elementwise_fma(wj, (f32x2){v[j], v[j + 1]}, a01);
      a23 = __builtin_elementwise_fma(wj, (f32x2){v[j + 2], v[j + 3]}, a23);
    }
    const float a[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (y0 + i >= H) continue;
      keep[u][i] = a[i];
      best_v = fmaxf(best_v, a[i]);
    }
  }
}

// exact value c*(y, x) by one wave: lanes 0..T-1 each run one row's float64 chain, lane 0 the column chain
__device__ __forceinline__ float exact_conv_at(const float *__restrict__ rawp, int WP, int H, int W, int R,
                                               const double *__restrict__ wk, int y, int x, double *__restrict__ scratch) {
  const int lane = threadIdx.x & 63, T = 2 * R + 1;
  if (lane < T) {
    const int yy = reflect_idx(y - R + lane, H);
    const float *rp = rawp + yy * WP + DEC_HALO + x - R;       // halo columns hold the reflected values (R <= 12)
    double t = 0.0;
    for (int j = 0; j < T; ++j) t = fma(wk[j], (double)rp[j], t);
    scratch[lane] = t;
  }
  __builtin_amdgcn_wave_barrier();
  double c = 0.0;
  for (int i = 0; i < T; ++i) c = fma(wk[i], scratch[i], c);    // every lane computes it: uniform result
  __builtin_amdgcn_wave_barrier();
  return (float)c;
}

// DF_KEEP: column-pass items a thread keeps in registers (4 outputs each): 3 covers 64x48 maps (92 VGPRs: five
// workgroups per CU, one round for the 1 088 maps of a bs-64 K=17 batch), 7 covers 96x72.
template <int DF_KEEP>
__global__ __launch_bounds__(DF_THREADS, DF_KEEP <= 3 ? 5 : 4) void decode_screen_kernel(
    const float *__restrict__ heatmaps, const float *prob, const float *vis, const float *oks, const float *err, int B,
    int K, int H, int W, const double *__restrict__ taps, const int *__restrict__ radius, double den_x, double den_y,
    double in_w, double in_h, DecodeOut o) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = DF_THREADS / 64;
  const int HW = H * W;
  const int WP = 4 * ((W + 3) >> 2) + 2 * DEC_HALO;
  float *rawp = reinterpret_cast<float *>(smem);                              // [H][WP]
  float *tmp32 = rawp + (size_t)H * WP;                                       // [HW]
  int *cand = reinterpret_cast<int *>(tmp32 + HW);                            // [DF_MAXCAND]
  __shared__ double wk[PP_MAX_TAPS];
  __shared__ double scratch[NW][PP_MAX_TAPS + 1];
  __shared__ float redf[NW], redmin[NW], redmax[NW];
  __shared__ Best red[NW];
  __shared__ int ncand_s;
  __shared__ float nbr[5];

  const int map = blockIdx.x, k = map % K, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float *__restrict__ src = heatmaps + (size_t)map * HW;
#ifdef PP_DSC_STAMPS
  unsigned long long st[8];
#define DSC(i_) st[i_] = __builtin_amdgcn_s_memtime()
#else
#define DSC(i_)
#endif
  DSC(0);
  const int r = __builtin_amdgcn_readfirstlane(radius[k]);
  if (tid < PP_MAX_TAPS) wk[tid] = taps[k * PP_MAX_TAPS + tid];
  if (tid == 0) ncand_s = 0;
  // 1. HBM -> LDS (+ reflected halo), A = max |x| (NaN / inf make A non-finite)
  float amax = 0.f, vmin = __builtin_inff(), vmax = -__builtin_inff();
  bool bad = false;
  if ((W & 3) == 0) {
    const int W4 = W >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (int p = tid; p < H * W4; p += DF_THREADS) {
      const int y = p / W4, xq = p - y * W4;
      const float4 v = s4[p];
      *reinterpret_cast<float4 *>(rawp + y * WP + DEC_HALO + 4 * xq) = v;
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      vmin = fminf(fminf(vmin, fminf(v.x, v.y)), fminf(v.z, v.w));
      vmax = fmaxf(fmaxf(vmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
      bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
    }
  } else {
    for (int p = tid; p < HW; p += DF_THREADS) {
      const int y = p / W, x = p - y * W;
      const float v = src[p];
      rawp[y * WP + DEC_HALO + x] = v;
      amax = fmaxf(amax, fabsf(v));
      vmin = fminf(vmin, v);
      vmax = fmaxf(vmax, v);
      bad |= !(fabsf(v) <= 3.0e38f);
    }
  }
  {
    const int npad = WP - W;
    const int q = tid & 31;
    if (q < npad) {
      const int col = q < DEC_HALO ? q : q + W;
      const int sx = reflect_idx(col - DEC_HALO, W);
      for (int y = tid >> 5; y < H; y += DF_THREADS / 32) rawp[y * WP + col] = src[y * W + sx];
    }
  }
  amax = wave_max(amax);
  vmax = wave_max(vmax);
  vmin = -wave_max(-vmin);
  const unsigned long long badm = __ballot(bad);
  if (lane == 0) {
    redf[wave] = badm ? __builtin_inff() : amax;
    redmin[wave] = vmin;
    redmax[wave] = vmax;
  }
  __syncthreads();
  float A = redf[0], gmin = redmin[0], gmax = redmax[0];
#pragma unroll
  for (int w_ = 1; w_ < NW; ++w_) {
    A = fmaxf(A, redf[w_]);
    gmin = fminf(gmin, redmin[w_]);
    gmax = fmaxf(gmax, redmax[w_]);
  }
  const bool finite = A <= 3.0e38f;
  DSC(1);
  if (finite && gmin == gmax) {
    // constant map (e.g. an all-zero channel after the clamp): every pixel's float64 chain sees the same inputs, so
    // every convolved value is the same number and np.argmax returns index 0 -- a border pixel, no sub-pixel step
    if (tid == 0) {
      auto at = [&](int, int) { return 0.f; };
      finalize(map, B, K, H, W, 0, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
    }
    return;
  }

  // 2 + 3. float32 screening
  float thr = -__builtin_inff();        // non-finite map: every pixel is a candidate
  const int H4 = (H + 3) >> 2;
  const bool keepable = H4 * W <= DF_KEEP * DF_THREADS;
  bool listed = false;                  // candidates are in cand[0 .. ncand_s)
  if (finite) {
    float w32[PP_MAX_TAPS];
#pragma unroll
    for (int j = 0; j < PP_MAX_TAPS; ++j) w32[j] = (float)taps[k * PP_MAX_TAPS + j];
    float best = -__builtin_inff();
    float keep[DF_KEEP][4];
#define PP_SCREEN(R_) screen_passes<R_, DF_KEEP>(rawp, WP, tmp32, H, W, w32, keep, best)
    switch (r) {
      case 2: PP_SCREEN(2); break;
      case 3: PP_SCREEN(3); break;
      case 4: PP_SCREEN(4); break;
      case 5: PP_SCREEN(5); break;
      case 6: PP_SCREEN(6); break;
      case 7: PP_SCREEN(7); break;
      case 8: PP_SCREEN(8); break;
      default: PP_SCREEN(9); break;
    }
#undef PP_SCREEN
    best = wave_max(best);
    __syncthreads();                     // redf readers above are done
    if (lane == 0) redf[wave] = best;
    __syncthreads();
    float m32 = redf[0];
#pragma unroll
    for (int w_ = 1; w_ < NW; ++w_) m32 = fmaxf(m32, redf[w_]);
    thr = m32 - 128.0f * 5.9604645e-08f * A;
    DSC(2);
    if (keepable) {
      listed = true;
#pragma unroll
      for (int u = 0; u < DF_KEEP; ++u) {
        const int it = tid + u * DF_THREADS;
        const int y4 = it / W, x = it - y4 * W;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (keep[u][i] >= thr) {       // rows beyond H and items beyond the map hold -inf
            const int slot = atomicAdd(&ncand_s, 1);
            if (slot < DF_MAXCAND) cand[slot] = (y4 * 4 + i) * W + x;
          }
      }
    }
  }
  __syncthreads();
  const int nc = (finite && listed) ? ncand_s : DF_MAXCAND + 1;
  DSC(3);

  // 4. exact values of the candidates, first-index arg-max (np.argmax semantics incl. NaN)
  Best mine;
  mine.v = -__builtin_inff();
  mine.i = 0x7fffffff;
  bool have = false;
  auto consider = [&](int p) {
    const float cv = exact_conv_at(rawp, WP, H, W, r, wk, p / W, p % W, scratch[wave]);
    if (!have || better(cv, p, mine.v, mine.i)) {
      mine.v = cv;
      mine.i = p;
      have = true;
    }
  };
  if (nc <= DF_MAXCAND) {
    for (int ci = wave; ci < nc; ci += NW) consider(cand[ci]);
  } else if (finite) {
    // more candidates than the list holds (flat map): every pixel whose float32 value passes the threshold.  The
    // float32 column sum is re-evaluated per pixel by the whole wave redundantly (uniform), then the exact value.
    for (int p = wave; p < HW; p += NW) {
      const int y = p / W, x = p - y * W;
      float a = 0.f;
      for (int j = 0; j <= 2 * r; ++j) a = fmaf((float)wk[j], tmp32[reflect_idx(y - r + j, H) * W + x], a);
      if (a >= thr) consider(p);
    }
  } else {
    for (int p = wave; p < HW; p += NW) consider(p);
  }
  if (!have || lane != 0) {
    // every lane of a wave holds the same (mine); keep one copy per wave for the reduction
    if (lane != 0 || !have) {
      mine.v = -__builtin_inff();
      mine.i = 0x7fffffff;
    }
  }
  DSC(4);
  __syncthreads();
  // block_argmax's better() treats a NaN value as winning; the "no candidate" sentinel (-inf, INT_MAX) never wins
  const Best b = block_argmax(mine, red);
  // exact neighbours of the winner (interior only), one wave each
  const int bx = b.i % W, by = b.i / W;
  const bool interior = bx > 0 && bx < W - 1 && by > 0 && by < H - 1;
  if (interior) {
    const int dyx[5][2] = {{0, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}};
    for (int q = wave; q < 5; q += NW) {
      const float cv = exact_conv_at(rawp, WP, H, W, r, wk, by + dyx[q][0], bx + dyx[q][1], scratch[wave]);
      if (lane == 0) nbr[q] = cv;
    }
  }
  __syncthreads();
  DSC(5);
  if (tid == 0) {
    auto at = [&](int yy, int xx) {
      return yy == by ? (xx == bx ? nbr[0] : (xx == bx + 1 ? nbr[1] : nbr[2])) : (yy == by + 1 ? nbr[3] : nbr[4]);
    };
    finalize(map, B, K, H, W, b.i, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
  }
#ifdef PP_DSC_STAMPS
  DSC(6);
  if (tid == 0) {
    unsigned long long *d = reinterpret_cast<unsigned long long *>(o.locs + (size_t)2 * B * K) + (size_t)map * 8;
    d[0] = st[1] - st[0]; d[1] = st[2] - st[1]; d[2] = st[3] - st[2]; d[3] = st[4] - st[3]; d[4] = st[5] - st[4];
    d[5] = st[6] - st[0]; d[6] = r; d[7] = nc;
  }
#endif
#undef DSC
}

// ---------------------------------------------------------------------------
// Wave-per-map path (64x48 maps: every 256x192 model; 96x72: the 384x288 models).  Same idea as the screened path
// above -- float32 screening, the float64 chains only at the candidates -- but ONE WAVE owns a map from the load to
// the result, so there is no workgroup barrier, no shared reduction and no inter-wave hand-off anywhere: 12 maps
// (64x48) or 5 (96x72) are in flight per CU, each an independent instruction stream, and the float32 passes run out
// of registers:
//   1. the map goes HBM -> LDS once with lane-linear 16-byte loads, A = max |x| comes with it;
//   2. row pass: lane = row (NRI rounds of HI = H / NRI rows), its W inputs in registers, reflection resolved at
//      compile time;
//   3. the intermediate crosses the lanes through the SAME LDS buffer, transposed: round i writes a [W][HI + pad]
//      image over the raw rows it has just consumed (conflict-free strides both ways);
//   4. column pass: lane = column (NCI rounds of WI = W / NCI columns), its H inputs in registers; the float32
//      results go back over the column they came from, the lane keeps only its maximum;
//   5. wave max -> threshold -> the columns that reach it are scanned by the whole wave (one row per lane) ->
//      candidates -> exact float64 value of each candidate by T lanes (one row chain each, the map re-read from L2) +
//      the column chain; then the four neighbours of the winner in one more round of 3T + 2 row chains; finalize()
//      on lane 0.
// Non-finite maps and flat maps (clamped plateaus: more than DWV_MAXCAND candidates) are not settled by a wave: it
// appends the map to a list in the workspace and the all-pixel float64 kernel (decode_lds_list_kernel, 512 threads per
// map) that follows on the stream decodes exactly those maps; the list resets itself there (no reset launch).
// Round 3 first tried to keep that work INSIDE the wave kernel's launch (helper workgroups polling an in-launch work list,
// the last screening workgroup draining it).  It was correct and it lost: (1) `sc1` loads of the list counters went
// stale across XCDs (helpers that had polled the empty list never saw pushes), so every read had to become an atomic
// read-modify-write; (2) 64 - 256 pollers on three counter words (a word serves ~88 atomics per us chip-wide) plus one
// arrival atomic per screening workgroup cost the launch 5 - 10 % at B = 1024; (3) helpers are the last block ids, so they
// start when the screening is nearly over, and a 256-thread workgroup decodes a flat map in 12 us where the 512-thread
// list kernel needs 8: on the bench model's heatmaps (5 % flat maps) 588 us per 256 crops against 69 for the all-pixel
// kernel alone.  An EARLY flat-map test on the raw map ("n pixels at the maximum", n = 12 or (2r + 1)^2) handed a third of
// those maps to the slow path (a saturated blob rarely holds a whole all-maximum kernel window).  All of it was removed.
// (A wave-local exact path for flat maps -- row chains shared through an LDS ring -- was built in round 2 and measured
// at 150 us per 64-crop batch of random-weight heatmaps: one wave's float64 chains are latency-bound.)
// ---------------------------------------------------------------------------
constexpr int DWV_MAXCAND = 2;           // more candidates than this: the map goes to the all-pixel kernel's list
                                         // (a candidate costs a wave ~1.5 us of serial float64 chains; 94 % of maps have 1)

constexpr __host__ __device__ int reflect_c(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i); }

template <int H_, int W_> struct WaveGeom {
  static constexpr int H = H_, W = W_, HW = H * W;
  static constexpr int NRI = (H + 63) / 64, HI = H / NRI;        // row-pass rounds, rows per round
  static constexpr int NCI = (W + 63) / 64, WI = W / NCI;        // column-pass rounds, columns per round
  static constexpr int RS = W + 4;                               // raw row stride (floats): 16-B rows, conflict-free b128
  static constexpr int TPAD = NRI == 1 ? 4 : 2;                  // transposed column stride HI + TPAD
  static constexpr int TS = HI + TPAD;
  static constexpr int SEG = HI * RS;                            // raw rows of one round = home of its transposed image
  static constexpr int BUF = H * RS;
  static constexpr int NLD = HW / 256;                           // 16-byte loads per lane
  static_assert(H % NRI == 0 && W % NCI == 0 && W % 4 == 0 && HI % 4 == 0 && HW % 256 == 0, "wave-per-map geometry");
  static_assert(W * TS <= SEG, "a round's transposed image fits over the raw rows it replaces");
  static_assert(H <= 128 && W <= 128, "two rounds at most are meant");
};

// pair i of a register-resident line of 2 * N2 values, i possibly outside [0, N2): scipy 'reflect' maps element e < 0
// to -e - 1 and e >= N to 2N - 1 - e, which for the pair (2i, 2i + 1) is the SWAPPED pair -i - 1 (resp. 2 N2 - 1 - i)
template <int N2>
__device__ __forceinline__ f32x2 pair_r(const f32x2 (&a)[N2], int i) {
  if (i >= 0 && i < N2) return a[i];
  const int k = i < 0 ? -i - 1 : 2 * N2 - 1 - i;
  return (f32x2){a[k].y, a[k].x};
}

// One float32 pass over a line of 2 * N2 values held as aligned pairs, two outputs per instruction: the taps with an
// even offset accumulate the output pair (2m, 2m + 1), the taps with an odd offset the pair (2m - 1, 2m), so that
// EVERY operand is an aligned register pair of the line (no shifted copy); an output is the sum of its two halves.
// (The float32 result differs from the one-chain order by rounding only: it screens, it does not decide.)
template <int N2, int R, typename Emit>
__device__ __forceinline__ void pass_pairs(const f32x2 (&a)[N2], const f32x2 (&w2)[2 * R + 1], Emit emit) {
  constexpr int T = 2 * R + 1;
  auto odd_pair = [&](int m) __attribute__((always_inline)) {      // outputs (2m - 1, 2m)
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j)
      if ((j - R) & 1) acc = __builtin_elementwise_fma(w2[j], pair_r<N2>(a, m + (j - R - 1) / 2), acc);
    return acc;
  };
  f32x2 o_cur = odd_pair(0);
#pragma unroll
  for (int m = 0; m < N2; ++m) {
    f32x2 e = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < T; ++j)
      if (!((j - R) & 1)) e = __builtin_elementwise_fma(w2[j], pair_r<N2>(a, m + (j - R) / 2), e);
    const f32x2 o_nxt = odd_pair(m + 1);
    emit(2 * m, e.x + o_cur.y);
    emit(2 * m + 1, e.y + o_nxt.x);
    o_cur = o_nxt;
  }
}

// steps 2-4 for a compile-time radius; returns the lane's maximum over its columns (lanes without a column: -inf);
// afterwards the transposed image holds the float32 convolved value of every pixel
template <typename G, int R>
__device__ __forceinline__ float wave_passes(float *__restrict__ buf, const double *__restrict__ wk) {
  constexpr int T = 2 * R + 1, H = G::H, W = G::W;
  const int lane = threadIdx.x & 63;
  f32x2 w2[T];
#pragma unroll
  for (int j = 0; j < T; ++j) {
    const float wj = (float)wk[j];
    w2[j] = (f32x2){wj, wj};
  }
#pragma unroll
  for (int ri = 0; ri < G::NRI; ++ri) {
    f32x2 in[W / 2];
    const int y = ri * G::HI + (lane < G::HI ? lane : G::HI - 1);
#pragma unroll
    for (int q = 0; q < W / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(buf + y * G::RS + 4 * q);
      in[2 * q] = (f32x2){t.x, t.y};
      in[2 * q + 1] = (f32x2){t.z, t.w};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the whole wave has its rows: this round's raw rows are free
    float *tb = buf + ri * G::SEG;
    // lanes beyond the round's rows hold a copy of its last row and write that row's values again: harmless, and
    // it keeps the stores free of per-output branches
    const int yl = lane < G::HI ? lane : G::HI - 1;
    pass_pairs<W / 2, R>(in, w2, [&](int x, float a) __attribute__((always_inline)) { tb[x * G::TS + yl] = a; });
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // transposed intermediate is in LDS
  float lm = -__builtin_inff();
#pragma unroll
  for (int ci = 0; ci < G::NCI; ++ci) {
    f32x2 col[H / 2];
    const bool act = lane < G::WI;
    const int x = ci * G::WI + (act ? lane : G::WI - 1);
#pragma unroll
    for (int ri = 0; ri < G::NRI; ++ri) {
      const float *tp = buf + ri * G::SEG + x * G::TS;
      if constexpr (G::TPAD == 4) {
#pragma unroll
        for (int q = 0; q < G::HI / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4 *>(tp + 4 * q);
          col[ri * G::HI / 2 + 2 * q] = (f32x2){t.x, t.y};
          col[ri * G::HI / 2 + 2 * q + 1] = (f32x2){t.z, t.w};
        }
      } else {
#pragma unroll
        for (int q = 0; q < G::HI / 2; ++q) {
          const float2 t = *reinterpret_cast<const float2 *>(tp + 2 * q);
          col[ri * G::HI / 2 + q] = (f32x2){t.x, t.y};
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float lmc = -__builtin_inff();
    pass_pairs<H / 2, R>(col, w2, [&](int y, float a) __attribute__((always_inline)) {
      buf[(y / G::HI) * G::SEG + x * G::TS + (y % G::HI)] = a;         // back over the column it came from (lanes
      lmc = fmaxf(lmc, a);                                             // without a column repeat the last one)
    });
    if (act) lm = fmaxf(lm, lmc);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return lm;
}

// one float64 row chain of the reference's separable evaluation, read from global (L2-hot) memory.  The taps sit in
// registers and the loop is unrolled to the largest kernel with a uniform guard, so that the loads are issued together
// and only the FMAs form the chain (a runtime-length loop over taps in memory costs a round trip per step).
// NT: compile-time bound on the taps (7 / 13 / 19: the kernel sizes of radius <= 3 / <= 6 / <= 9), so that a small
// kernel does not pay for 19 guarded steps.
template <int NT>
__device__ __forceinline__ double dwv_row_chain(const float *__restrict__ src, int H, int W, int R, int T,
                                                const double (&wkr)[PP_MAX_TAPS], int yy, int x) {
  const float *rp = src + reflect_once(yy, H) * W;
  float v[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) v[i] = rp[reflect_once(x - R + (i < T ? i : 0), W)];   // no branch: the loads batch
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const double tn = fma(wkr[i], (double)v[i], t);
    t = i < T ? tn : t;                                   // steps beyond the kernel are computed and dropped
  }
  return t;
}
// column chain over the row-chain results held by lanes base .. base + T - 1
template <int NT>
__device__ __forceinline__ double dwv_col_chain(double t, int base, int T, const double (&wkr)[PP_MAX_TAPS]) {
  double c = 0.0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const double cn = fma(wkr[j], __shfl(t, (base + j) & 63, 64), c);
    c = j < T ? cn : c;
  }
  return c;
}

// hand-over list in the workspace: [0] = count, [1] = list-kernel workgroups that have read the count (its reset),
// [2 ..] = map indices.  Pushed by single lanes of the wave kernel, consumed by decode_lds_list_kernel, the next launch
// on the stream (a kernel boundary: plain device-scope atomics and plain loads suffice).
// (`maps` bounds the slot: a caller that broke the contract -- a workspace that was never zeroed -- gets wrong results for
// the dropped maps, never an out-of-bounds write.)
__device__ __forceinline__ void wl_push(int *ws, int map, int maps) {
  const int i = atomicAdd(ws, 1);
  if ((unsigned)i < (unsigned)maps) ws[2 + i] = map;
}

template <int H, int W>
__device__ __forceinline__ void wave_decode_map(
    const int map, float *__restrict__ buf, const float *__restrict__ heatmaps, const float *prob, const float *vis,
    const float *oks, const float *err, int B, int K, const double *__restrict__ taps, const int *__restrict__ radius,
    double den_x, double den_y, double in_w, double in_h, DecodeOut o, int *__restrict__ ws) {
  using G = WaveGeom<H, W>;
  constexpr int HW = G::HW;
  const int lane = threadIdx.x & 63;
  const int k = map % K;
  const float *__restrict__ src = heatmaps + (size_t)map * HW;
  const int r = __builtin_amdgcn_readfirstlane(radius[k]);
  const double *__restrict__ wk = taps + (size_t)k * PP_MAX_TAPS;

#ifdef PP_DWV_STAMPS
  unsigned long long st[8];
#define DWS(i_) st[i_] = __builtin_amdgcn_s_memtime()
#else
#define DWS(i_)
#endif
  DWS(0);
  // 1. HBM -> LDS (row stride RS), A = max |x| (NaN / inf make the map "not finite")
  float amax = 0.f, vmin = __builtin_inff(), vmax = -__builtin_inff();
  bool bad = false;
  {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    constexpr int CH = 9;                              // loads in flight per lane
#pragma unroll
    for (int i0 = 0; i0 < G::NLD; i0 += CH) {
      float4 v[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (i0 + i < G::NLD) v[i] = s4[(i0 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (i0 + i >= G::NLD) continue;
        const int e = 4 * ((i0 + i) * 64 + lane), y = e / W, x = e - y * W;
        *reinterpret_cast<float4 *>(buf + y * G::RS + x) = v[i];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i].x), fabsf(v[i].y))), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        vmin = fminf(fminf(vmin, fminf(v[i].x, v[i].y)), fminf(v[i].z, v[i].w));
        vmax = fmaxf(fmaxf(vmax, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
        bad |= !(fabsf(v[i].x) <= 3.0e38f) | !(fabsf(v[i].y) <= 3.0e38f) | !(fabsf(v[i].z) <= 3.0e38f) |
               !(fabsf(v[i].w) <= 3.0e38f);
      }
    }
  }
  const float A = wave_max(amax), gmax = wave_max(vmax), gmin = -wave_max(-vmin);
  const bool finite = __ballot(bad) == 0ull;
  if (finite && gmin == gmax) {
    // constant map: every convolved value is the same number, np.argmax returns index 0 (a border pixel)
    if (lane == 0) {
      auto at = [&](int, int) { return 0.f; };
      finalize(map, B, K, H, W, 0, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
    }
    return;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the map is in LDS (this wave's own buffer)
  DWS(1);
  int cand_r[(DWV_MAXCAND + 63) / 64];                       // candidate ci lives in lane ci % 64 (register, not LDS:
  cand_r[0] = 0;                                             // the buffer still holds the convolved map while they are found)
  int ncand = DWV_MAXCAND + 1;                               // "every pixel" unless the screen says less
  if (finite) {
    float lm;
    switch (r) {
      case 2: lm = wave_passes<G, 2>(buf, wk); break;
      case 3: lm = wave_passes<G, 3>(buf, wk); break;
      case 4: lm = wave_passes<G, 4>(buf, wk); break;
      case 5: lm = wave_passes<G, 5>(buf, wk); break;
      case 6: lm = wave_passes<G, 6>(buf, wk); break;
      case 7: lm = wave_passes<G, 7>(buf, wk); break;
      case 8: lm = wave_passes<G, 8>(buf, wk); break;
      default: lm = wave_passes<G, 9>(buf, wk); break;
    }
    const float m32 = wave_max(lm);
    const float thr = m32 - 128.0f * 5.9604645e-08f * A;     // see the error budget of the screened path above
    DWS(2);
    // columns whose maximum reaches the threshold (usually one) are scanned by the whole wave, one row per lane.
    // A lane holds the maximum over its NCI columns, so every column of a flagged lane is scanned.
    ncand = 0;
    unsigned long long flagged = __ballot(lm >= thr);
    while (flagged) {
      const int fl = __builtin_ctzll(flagged);
      flagged &= flagged - 1;
#pragma unroll
      for (int ci = 0; ci < G::NCI; ++ci) {
        const int x = ci * G::WI + fl;
#pragma unroll
        for (int y0 = 0; y0 < H; y0 += 64) {
          const int y = y0 + lane;
          const bool pred = y < H && buf[(y / G::HI) * G::SEG + x * G::TS + (y % G::HI)] >= thr;
          const unsigned long long m = __ballot(pred);
          // candidate number ncand + q is kept by lane ncand + q: that lane fetches it from the q-th set lane of m
          const int cnt = __builtin_popcountll(m);
          if (cnt && ncand + cnt <= DWV_MAXCAND) {
            // slot s = ncand + rank (< 64): lane s takes the value from the rank-th set lane of m
            const int want = lane - ncand;                    // rank this lane's slot asks for
            int srcl = 0;
            if (want >= 0 && want < cnt) {
              unsigned long long mm = m;
              for (int q = 0; q < want; ++q) mm &= mm - 1;
              srcl = __builtin_ctzll(mm);
            }
            const int got = __shfl(y * W + x, srcl, 64);
            if (want >= 0 && want < cnt) cand_r[0] = got;
          }
          ncand += cnt;
        }
      }
    }
  }

  DWS(3);
  if (ncand > DWV_MAXCAND) {
    // Flat map (a clamped plateau: hundreds of exact ties and near-ties) or non-finite map (every pixel counts): not
    // worth a wave's serial chains.  It goes on the work list of the all-pixel float64 decode (whole workgroups).
    if (lane == 0) wl_push(ws, map, B * K);
    return;
  }
  // 5. exact values, first-index arg-max (np.argmax semantics incl. NaN).  Row chains on lanes 0..T-1, handed to the
  // column chain by lane shuffles (no LDS: the buffer still holds the screened map).
  const int T = 2 * r + 1;
  double wkr[PP_MAX_TAPS];
#pragma unroll
  for (int j = 0; j < PP_MAX_TAPS; ++j) wkr[j] = wk[j];
  auto exact_at = [&](int y, int x) __attribute__((always_inline)) {
    double t = 0.0, c;
    if (T <= 7) {
      if (lane < T) t = dwv_row_chain<7>(src, H, W, r, T, wkr, y - r + lane, x);
      c = dwv_col_chain<7>(t, 0, T, wkr);
    } else if (T <= 13) {
      if (lane < T) t = dwv_row_chain<13>(src, H, W, r, T, wkr, y - r + lane, x);
      c = dwv_col_chain<13>(t, 0, T, wkr);
    } else {
      if (lane < T) t = dwv_row_chain<PP_MAX_TAPS>(src, H, W, r, T, wkr, y - r + lane, x);
      c = dwv_col_chain<PP_MAX_TAPS>(t, 0, T, wkr);
    }
    return (float)c;                       // the same number on every lane
  };
  Best best;
  best.v = -__builtin_inff();
  best.i = 0x7fffffff;
  bool have = false;
  auto consider = [&](int p, float cv) __attribute__((always_inline)) {
    if (!have || better(cv, p, best.v, best.i)) {
      best.v = cv;
      best.i = p;
      have = true;
    }
  };
  // One candidate (94 % of maps): it IS the arg-max, and its own exact value comes out of the neighbours' round below
  // (the centre column's row chains are part of it) instead of a round of its own.
  const bool single = ncand == 1;
  if (single) {
    best.i = __shfl(cand_r[0], 0, 64);
    have = true;
  } else {
    for (int ci = 0; ci < ncand; ++ci) {
      const int p = __shfl(cand_r[0], ci, 64);
      consider(p, exact_at(p / W, p % W));
    }
  }
  DWS(4);
  // exact neighbours of the winner (interior only): lanes [0, T) column x + 1, [T, 2T) column x - 1, [2T, 3T + 2)
  // column x over rows y - 1 - r .. y + 1 + r; then four (five) column chains on lanes 0..3 (4)
  const int bx = best.i % W, by = best.i / W;
  const bool interior = bx > 0 && bx < W - 1 && by > 0 && by < H - 1;
  float nb = 0.f;                 // lane q < 4: value at (x + 1), (x - 1), (y + 1), (y - 1)
  if (interior) {
    double t = 0.0;
    if (lane < 3 * T + 2) {
      const int grp = lane < T ? 0 : (lane < 2 * T ? 1 : 2);
      const int idx = lane - grp * T;
      const int xx = grp == 0 ? bx + 1 : (grp == 1 ? bx - 1 : bx);
      const int yy = grp == 2 ? by - 1 - r + idx : by - r + idx;
      if (T <= 7) t = dwv_row_chain<7>(src, H, W, r, T, wkr, yy, xx);
      else if (T <= 13) t = dwv_row_chain<13>(src, H, W, r, T, wkr, yy, xx);
      else t = dwv_row_chain<PP_MAX_TAPS>(src, H, W, r, T, wkr, yy, xx);
    }
    // lane 4: the centre itself (rows y - r .. y + r of the third group)
    const int base = lane == 0 ? 0 : (lane == 1 ? T : (lane == 2 ? 2 * T + 2 : (lane == 3 ? 2 * T : 2 * T + 1)));
    const double c = T <= 7 ? dwv_col_chain<7>(t, base, T, wkr)          // lanes >= 5: unused values
                            : (T <= 13 ? dwv_col_chain<13>(t, base, T, wkr) : dwv_col_chain<PP_MAX_TAPS>(t, base, T, wkr));
    nb = (float)c;
    if (single) best.v = __shfl(nb, 4, 64);
  } else if (single) {
    best.v = exact_at(by, bx);     // border arg-max: no neighbours, the value alone
  }
  const float n_xp = __shfl(nb, 0, 64), n_xm = __shfl(nb, 1, 64), n_yp = __shfl(nb, 2, 64), n_ym = __shfl(nb, 3, 64);
  if (lane == 0) {
    auto at = [&](int yy, int xx) {
      return yy == by ? (xx == bx ? best.v : (xx == bx + 1 ? n_xp : n_xm)) : (yy == by + 1 ? n_yp : n_ym);
    };
    finalize(map, B, K, H, W, best.i, at, src, prob, vis, oks, err, den_x, den_y, in_w, in_h, o);
  }
#ifdef PP_DWV_STAMPS
  DWS(5);
  if (lane == 0) {   // diagnostic build: phase cycles behind the locs array (tools/dwv_stamps.py sizes it)
    unsigned long long *d = reinterpret_cast<unsigned long long *>(o.locs + (size_t)2 * B * K) + (size_t)map * 8;
    d[0] = st[1] - st[0]; d[1] = st[2] - st[1]; d[2] = st[3] - st[2]; d[3] = st[4] - st[3]; d[4] = st[5] - st[4];
    d[5] = st[5] - st[0]; d[6] = r; d[7] = ncand;
  }
#endif
#undef DWS
}

// NWV waves (= maps) per workgroup; MINW = waves per SIMD the register budget is cut for.
template <int H, int W, int NWV, int MINW>
__global__ __launch_bounds__(NWV * 64, MINW) void decode_wave_kernel(
    const float *__restrict__ heatmaps, const float *prob, const float *vis, const float *oks, const float *err, int B,
    int K, const double *__restrict__ taps, const int *__restrict__ radius, double den_x, double den_y, double in_w,
    double in_h, DecodeOut o, int *__restrict__ ws) {
  using G = WaveGeom<H, W>;
  __shared__ __attribute__((aligned(16))) float lds[NWV][G::BUF];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int map = blockIdx.x * NWV + wave;
  if (map >= B * K) return;                          // whole waves leave; nothing synchronises across waves
  wave_decode_map<H, W>(map, lds[wave], heatmaps, prob, vis, oks, err, B, K, taps, radius, den_x, den_y, in_w, in_h, o, ws);
}

// The all-pixel float64 decode over the LIST the wave kernel left (the launch before this one on the stream).  The grid
// is fixed (graph-capturable); workgroups stride over the list and leave at once when it is empty.  The list resets
// itself: every workgroup reads the count, then arrives on a second counter; the last to arrive -- every workgroup has
// read the count by then -- zeroes both.  The caller zeroes the workspace once when it allocates it; no memset node, no
// reset kernel (round 2 spent 5 us per batch on one).
__global__ __launch_bounds__(DEC_THREADS) void decode_lds_list_kernel(
    int *__restrict__ list, const float *__restrict__ heatmaps, const float *prob, const float *vis,
    const float *oks, const float *err, int B, int K, int H, int W, const double *__restrict__ taps,
    const int *__restrict__ radius, double den_x, double den_y, double in_w, double in_h, DecodeOut o) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ Best red[DEC_THREADS / 64];
  __shared__ int s_n;
  if (threadIdx.x == 0) {
    s_n = __hip_atomic_load(list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (atomicAdd(list + 1, 1) == (int)gridDim.x - 1) {      // everybody has read the count
      __hip_atomic_store(list, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(list + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const int n = min(max(s_n, 0), B * K);              // (bounded whatever the workspace held: see wl_push)
  for (int b = blockIdx.x; b < n; b += gridDim.x) {
    decode_lds_map<DEC_THREADS>(min(max(list[2 + b], 0), B * K - 1), smem, red, heatmaps, prob, vis, oks, err, B, K, H, W, taps, radius, den_x,
                                den_y, in_w, in_h, o, nullptr);
    __syncthreads();                     // the next map reuses the LDS image and the reduction slots
  }
}

static size_t screen_lds_bytes(int H, int W) {
  return ((size_t)H * (4 * ((W + 3) / 4) + 2 * DEC_HALO) + (size_t)H * W + DF_MAXCAND) * 4;
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
static bool wave_geometry(int H, int W) { return (H == 64 && W == 48) || (H == 96 && W == 72); }
}  // namespace pp

extern "C" size_t pp_decode_workspace_bytes(int B, int K, int H, int W) {
  if (pp::wave_geometry(H, W)) return ((size_t)B * K + 2) * sizeof(int);   // hand-over list of the wave-per-map path (zeroed once by the caller)
  if (pp::fits_lds(H, W)) return 0;
  return (size_t)B * K * H * W * (sizeof(double) + sizeof(float));
}

extern "C" int pp_decode_f32(const float *heatmaps, const float *prob, const float *vis,
                             const float *oks, const float *err, int B, int K, int H, int W,
                             const double *taps, const int *radius, double den_x, double den_y,
                             double in_w, double in_h, double *out_kpts, float *out_scores,
                             float *out_locs, float *out_aux, double *out_err, float *out_conv,
                             double *out_packed, void *workspace, int flags, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, "pp_decode_f32: bad shape B=%d K=%d H=%d W=%d", B, K,
             H, W);
  if (B == 0) return 0;  // empty batch: nothing to launch (buffers may be null)
  PP_REQUIRE(heatmaps && taps && radius, "pp_decode_f32: null input");
  PP_REQUIRE((long long)H * W < (1ll << 30), "pp_decode_f32: map too large");
  hipStream_t s = (hipStream_t)stream;
  DecodeOut o{out_kpts, out_scores, out_locs, out_aux, out_err, out_packed};
  const int maps = B * K;
#ifndef PP_DEC_STAMPS
  constexpr size_t SCREEN_DYN_LIMIT = 156 * 1024;    // the kernel also holds ~1.5 KB of static LDS
  // Measured (tools/decode_ab.py, one process, interleaved): 96x72 maps 699 vs 884 us per 128 x 133 maps (-21 %); 64x48
  // maps 24.3 vs 22.9 us at B = 64 and 236 vs 234 us at B = 1024 (a tie: both forms are bound by the latency of their
  // short barrier-separated phases, not by HBM, LDS or the FMA rate).  The screened form is the default where it
  // wins (maps larger than 4096 pixels); flag PP_DECODE_SCREEN forces it everywhere, PP_DECODE_ALL_PIXEL never.
  // wave-per-map kernel: 64x48 maps (256x192 models) and 96x72 maps (384x288), when the caller passed the (zeroed)
  // workspace pp_decode_workspace_bytes asks for (without one: the workgroup-per-map kernels below)
  PP_REQUIRE((flags & ~(PP_DECODE_NO_WAVE | PP_DECODE_SCREEN | PP_DECODE_ALL_PIXEL | PP_DECODE_WAVE)) == 0,
             "pp_decode_f32: bad flags %d", flags);
  const bool exact_all = (flags & PP_DECODE_ALL_PIXEL) != 0;
  // Batches of at most two rounds of the all-pixel kernel's resident workgroups (3 per CU at 64x48, 1 at 96x72) take that
  // kernel: ONE launch, 22 - 24 us at bs 64 x 17 whatever the maps hold, where the wave kernel + its list kernel need
  // 20 - 25 us on peaked maps and ~27 on the bench model's plateau-ridden ones.  Above that the wave kernel wins by up to
  // 2.4x (B = 1024: 94 - 99 vs 232 us) and the second launch is noise.  PP_DECODE_WAVE forces the wave path at any size.
  static int ncu_ = 0;                       // (one device class per process; a wrong count only moves the cross-over)
  if (ncu_ == 0) {
    int dev_w = 0, n = 0;
    if (hipGetDevice(&dev_w) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_w) != hipSuccess || n <= 0)
      n = 256;
    ncu_ = n;
  }
  const long long small_batch = 2ll * ncu_ * (H == 64 ? 3 : 1);
  const bool wave_wanted = (flags & PP_DECODE_WAVE) || maps > small_batch;
  if (!out_conv && !(flags & PP_DECODE_NO_WAVE) && wave_wanted && !exact_all && workspace && ((uintptr_t)heatmaps & 15) == 0 &&
      ((uintptr_t)workspace & 3) == 0 && wave_geometry(H, W)) {
    int *ws = reinterpret_cast<int *>(workspace);       // [0] count, [1] arrivals of the list kernel, [2 ..] maps
    if (H == 64)
      hipLaunchKernelGGL((decode_wave_kernel<64, 48, 4, 3>), dim3((unsigned)cdiv(maps, 4)), dim3(256), 0, s, heatmaps, prob,
                         vis, oks, err, B, K, taps, radius, den_x, den_y, in_w, in_h, o, ws);
    else
      hipLaunchKernelGGL((decode_wave_kernel<96, 72, 1, 2>), dim3((unsigned)maps), dim3(64), 0, s, heatmaps, prob, vis, oks,
                         err, B, K, taps, radius, den_x, den_y, in_w, in_h, o, ws);
    PP_CHECK_LAUNCH("decode_wave_kernel");
    const size_t lds = lds_bytes(H, W);
    static thread_local unsigned long long attr_mask3 = 0;
    int dev3 = 0;
    if (lds > 64 * 1024 && attr_needed(attr_mask3, dev3))
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(decode_lds_list_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_LIMIT));
    hipLaunchKernelGGL(decode_lds_list_kernel, dim3((unsigned)(maps < 256 ? maps : 256)), dim3(DEC_THREADS), lds, s, ws,
                       heatmaps, prob, vis, oks, err, B, K, H, W, taps, radius, den_x, den_y, in_w, in_h, o);
    PP_CHECK_LAUNCH("decode_lds_list_kernel");
    return 0;
  }
  const bool want_screen = (flags & PP_DECODE_SCREEN) || (long long)H * W > 4096;
  if (!out_conv && want_screen && !exact_all && screen_lds_bytes(H, W) <= SCREEN_DYN_LIMIT) {
    const size_t lds = screen_lds_bytes(H, W);
    const int items = ((H + 3) / 4) * W;
    if (items <= 3 * DF_THREADS) {
      hipLaunchKernelGGL(decode_screen_kernel<3>, dim3(maps), dim3(DF_THREADS), lds, s, heatmaps, prob, vis, oks, err, B,
                         K, H, W, taps, radius, den_x, den_y, in_w, in_h, o);
    } else {
      static thread_local unsigned long long attr_mask2 = 0;
      int dev2 = 0;
      if (lds > 48 * 1024 && attr_needed(attr_mask2, dev2))
        PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(decode_screen_kernel<7>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCREEN_DYN_LIMIT));
      hipLaunchKernelGGL(decode_screen_kernel<7>, dim3(maps), dim3(DF_THREADS), lds, s, heatmaps, prob, vis, oks, err, B,
                         K, H, W, taps, radius, den_x, den_y, in_w, in_h, o);
    }
    PP_CHECK_LAUNCH("decode_screen_kernel");
    return 0;
  }
#endif
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
