// ArgMaxProbMap.decode (reference codec.py:515-543): raw arg-max of every heatmap (heatmap.py:13-52,
// get_heatmap_maximum) refined by DARK-UDP (codec.py:315-375, refine_keypoints_dark_udp):
//
//   blur   = GaussianBlur(zero-pad(map, b), (k, k), sigma from k) cropped back          (codec.py:284-313, cv2)
//   blur  *= max(map) / (max(blur) + 1e-12)
//   L      = log(clip(blur, 1e-3, 50))
//   at the arg-max (x, y) of the RAW map, with L edge-padded by one pixel:
//     dx = (L[y, x+1] - L[y, x-1]) / 2, dy likewise, dxx, dyy, dxy finite differences,
//     (x, y) -= pinv([[dxx, dxy], [dxy, dyy]] + eps_f32 I) (dx, dy)                       (float64 solve)
//   keypoint = (x, y) / (W - 1, H - 1) * input_size                                       (float64)
//
// One workgroup per (crop, keypoint) map, the map read from HBM once (HBM-bound: 4 B per pixel in, 20 B out):
// the zero-padded map sits in LDS, the separable blur runs as a row pass and a column pass in float32 (the
// padding width equals the kernel radius, so the blur's own border mode never reaches the cropped result), the
// two maxima and the first-index arg-max come from wave shuffles + LDS, and one lane does the 3x3 stencil.
// cv2 is not importable in this environment and no reference fixture covers this decoder: PARITY UNPINNED.  The
// float32 summation order restated here (row pass: taps in index order; column pass: centre + symmetric pairs;
// no fused multiply-add) is the one oracle/probpose_oracle.py uses; OpenCV's own SIMD order may differ in the
// last float32 bit.
// Defined behaviour where the reference indexes out of bounds: a map whose maximum is <= 0 gets the location
// (-1, -1) (heatmap.py:47) and the reference then reads the NEIGHBOURING map through negative flat indices
// (codec.py:345-354); here such a keypoint is returned un-refined.
#include "pp_common.h"

namespace pp {

constexpr int DK_THREADS = 256;
constexpr int DK_MAX_TAPS = 31;

struct DarkTaps {
  float k[DK_MAX_TAPS];
};

__device__ __forceinline__ float2 block_max2(float a, float b, float *red) {   // max of a and of b over the block
  a = wave_max(a);
  b = wave_max(b);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[2 * wave] = a;
    red[2 * wave + 1] = b;
  }
  __syncthreads();
  float ra = red[0], rb = red[1];
#pragma unroll
  for (int w = 1; w < DK_THREADS / 64; ++w) {
    ra = fmaxf(ra, red[2 * w]);
    rb = fmaxf(rb, red[2 * w + 1]);
  }
  return make_float2(ra, rb);
}

__global__ __launch_bounds__(DK_THREADS) void dark_decode_kernel(const float *__restrict__ heatmaps, int H, int W,
                                                                 DarkTaps taps, int ksize, double in_w, double in_h,
                                                                 double *__restrict__ out_kpts,
                                                                 float *__restrict__ out_scores,
                                                                 float *__restrict__ out_locs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[2 * DK_THREADS / 64];
  __shared__ unsigned long long red_arg[DK_THREADS / 64];
  const int b = ksize / 2;                    // border = kernel radius (codec.py:302)
  const int PW = W + 2 * b, PH = H + 2 * b;
  float *P = reinterpret_cast<float *>(smem);                 // [PH][PW] zero-padded raw map, later [H][W] blurred
  float *R = P + (size_t)PH * PW;                             // [H][PW] column-pass output of ... see below
  const int tid = threadIdx.x;
  const float *src = heatmaps + (size_t)blockIdx.x * H * W;
  // ---- load (zero border) + raw maximum with first-index arg-max (np.argmax / np.amax, NaN-free heatmaps)
  for (int i = tid; i < PH * PW; i += DK_THREADS) P[i] = 0.f;
  __syncthreads();
  float vmax = -INFINITY;
  int imax = 0x7fffffff;
  for (int i = tid; i < H * W; i += DK_THREADS) {
    const int y = i / W, x = i - y * W;
    const float v = src[i];
    P[(y + b) * PW + x + b] = v;
    if (v > vmax) {            // ascending i per thread: the first maximum of this thread's pixels
      vmax = v;
      imax = i;
    }
  }
  // (value, index) -> one key: larger value wins, then the SMALLER index.  Heatmaps are finite; map the float to an
  // order-preserving unsigned key.
  unsigned ub = __float_as_uint(vmax);
  ub = (ub & 0x80000000u) ? ~ub : (ub | 0x80000000u);
  unsigned long long key = ((unsigned long long)ub << 32) | (unsigned)(0x7fffffff - imax);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o, 64);
    key = other > key ? other : key;
  }
  if ((tid & 63) == 0) red_arg[tid >> 6] = key;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < DK_THREADS / 64; ++w) key = red_arg[w] > key ? red_arg[w] : key;
  const int amax = 0x7fffffff - (int)(unsigned)(key & 0xffffffffu);
  const int ay = amax / W, ax = amax - ay * W;
  const float origin_max = P[(ay + b) * PW + ax + b];
  // ---- row pass over the H interior rows (border rows are all zero): R[y][xp] for every padded column is not
  // needed either -- only the W interior columns feed the cropped result.  R holds [PH][W] with zero border rows.
  for (int i = tid; i < PH * W; i += DK_THREADS) {
    const int yp = i / W, x = i - yp * W;
    float s = 0.f;
    if (yp >= b && yp < b + H) {
      const float *row = P + yp * PW + x;            // taps x .. x + 2b of the padded row
      s = taps.k[0] * row[0];
      for (int t = 1; t < ksize; ++t) s = s + taps.k[t] * row[t];
    }
    R[i] = s;
  }
  __syncthreads();
  // ---- column pass (symmetric form: centre tap + pairs), cropped output into P as [H][W]; track its maximum
  float bmax = -INFINITY;
  for (int i = tid; i < H * W; i += DK_THREADS) {
    const int y = i / W, x = i - y * W;
    const float *col = R + (y + b) * W + x;          // centre row y + b of the padded rows
    float s = taps.k[b] * col[0];
    for (int t = 1; t <= b; ++t) s = s + taps.k[b + t] * (col[t * W] + col[-t * W]);
    P[i] = s;
    bmax = fmaxf(bmax, s);
  }
  const float2 mx = block_max2(bmax, 0.f, red);       // barrier inside: every P[i] of the blur is written
  if (tid == 0) {
    const float scale = origin_max / (mx.x + 1e-12f);
    float x = (float)ax, y = (float)ay;
    const bool dead = !(origin_max > 0.0f);            // vals <= 0 -> locs = -1 (heatmap.py:47)
    if (dead) x = y = -1.f;
    const size_t o = blockIdx.x;
    out_locs[2 * o] = x;
    out_locs[2 * o + 1] = y;
    out_scores[o] = origin_max;
    if (!dead) {
      auto L = [&](int yy, int xx) {                    // log(clip(blur * scale)) with one-pixel edge padding
        yy = min(max(yy, 0), H - 1);
        xx = min(max(xx, 0), W - 1);
        const float v = fminf(fmaxf(P[yy * W + xx] * scale, 1e-3f), 50.f);
        return logf(v);
      };
      const float i_ = L(ay, ax), ix1 = L(ay, ax + 1), iy1 = L(ay + 1, ax), ix1y1 = L(ay + 1, ax + 1),
                  ix1_y1_ = L(ay - 1, ax - 1), ix1_ = L(ay, ax - 1), iy1_ = L(ay - 1, ax);
      const float dx = 0.5f * (ix1 - ix1_), dy = 0.5f * (iy1 - iy1_);
      const float dxx = ix1 - 2.f * i_ + ix1_, dyy = iy1 - 2.f * i_ + iy1_;
      const float dxy = 0.5f * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_);
      // pinv of the symmetric 2x2 matrix in float64 (np.linalg.pinv: singular values below 1e-15 * largest are dropped)
      const double eps = 1.1920928955078125e-07;       // np.finfo(np.float32).eps
      const double a = (double)dxx + eps, c = (double)dxy, d = (double)dyy + eps;
      const double tr = 0.5 * (a + d), df = 0.5 * (a - d), rad = sqrt(df * df + c * c);
      const double l1 = tr + rad, l2 = tr - rad;
      const double smax = fmax(fabs(l1), fabs(l2)), smin = fmin(fabs(l1), fabs(l2));
      double px = 0.0, py = 0.0;                        // pinv(M) (dx, dy)
      if (smax > 0.0) {
        if (smin > 1e-15 * smax) {
          const double det = a * d - c * c;
          px = (d * (double)dx - c * (double)dy) / det;
          py = (a * (double)dy - c * (double)dx) / det;
        } else {
          // rank one: project on the eigenvector of the dominant eigenvalue
          const double lam = fabs(l1) >= fabs(l2) ? l1 : l2;
          double vx = c, vy = lam - a;
          if (fabs(vx) + fabs(vy) < 1e-300) {
            vx = lam - d;
            vy = c;
          }
          if (fabs(vx) + fabs(vy) < 1e-300) {
            vx = fabs(a) >= fabs(d) ? 1.0 : 0.0;
            vy = 1.0 - vx;
          }
          const double nn = vx * vx + vy * vy;
          const double proj = (vx * (double)dx + vy * (double)dy) / nn / lam;
          px = proj * vx;
          py = proj * vy;
        }
      }
      x = (float)((double)x - px);                      // keypoints[n] -= ... on a float32 array (codec.py:371)
      y = (float)((double)y - py);
    }
    out_kpts[2 * o] = (double)x / (double)(W - 1) * in_w;          // codec.py:541
    out_kpts[2 * o + 1] = (double)y / (double)(H - 1) * in_h;
  }
}

}  // namespace pp

extern "C" size_t pp_dark_decode_lds_bytes(int H, int W, int ksize) {
  const int b = ksize / 2;
  return ((size_t)(H + 2 * b) * (W + 2 * b) + (size_t)(H + 2 * b) * W) * sizeof(float);
}

extern "C" int pp_dark_decode_f32(const float *heatmaps, int B, int K, int H, int W, const float *taps_host,
                                  int ksize, double in_w, double in_h, double *out_kpts, float *out_scores,
                                  float *out_locs, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && K > 0 && H > 1 && W > 1, "pp_dark_decode_f32: bad shape B=%d K=%d H=%d W=%d", B, K, H, W);
  PP_REQUIRE(ksize >= 1 && (ksize & 1) && ksize <= DK_MAX_TAPS, "pp_dark_decode_f32: blur kernel size %d (odd, <= %d)",
             ksize, DK_MAX_TAPS);
  if (B == 0) return 0;
  PP_REQUIRE(heatmaps && taps_host && out_kpts && out_scores && out_locs, "pp_dark_decode_f32: null pointer");
  PP_REQUIRE((long long)B * K < (1ll << 31), "pp_dark_decode_f32: too many maps");
  const size_t lds = pp_dark_decode_lds_bytes(H, W, ksize);
  PP_REQUIRE(lds <= 150 * 1024, "pp_dark_decode_f32: a %dx%d map with a %d-tap blur needs %zu B of LDS (> 150 KB)", H, W,
             ksize, lds);
  DarkTaps t;
  for (int i = 0; i < DK_MAX_TAPS; ++i) t.k[i] = i < ksize ? taps_host[i] : 0.f;
  static thread_local unsigned long long attr_mask = 0;
  int dev_ = 0;
  if (lds > 64 * 1024 && attr_needed(attr_mask, dev_))
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dark_decode_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  hipLaunchKernelGGL(dark_decode_kernel, dim3((unsigned)(B * K)), dim3(DK_THREADS), lds, (hipStream_t)stream, heatmaps,
                     H, W, t, ksize, in_w, in_h, out_kpts, out_scores, out_locs);
  PP_CHECK_LAUNCH("dark_decode_kernel");
  return 0;
}
