// Batched evaluation metrics on the device: PCK over [N, K] keypoint pairs.
//
// Replaces the per-instance host loop of the reference (heatmap.py:55-111 normalised distances + thresholding,
// called from loss.py:825-866 keypoint_pck_accuracy / loss.py:767-822 pose_pck_accuracy) by ONE pass over all
// N x K pairs: every lane owns one (instance, keypoint) pair, evaluates the normalised distance with numpy's
// operation order and roundings, and the per-keypoint hit / valid COUNTS are reduced in LDS and leave the device as
// 2 K integers.  Counts are integers, so the accuracies hits / valid computed from them on the host are the
// reference's numbers bit for bit (tests/golden/metrics.npz).
//
// HBM-bound by construction (20 B read per pair, nothing written but 8 K bytes); compiled -ffp-contract=off: the
// float64 sum of squares must round where numpy rounds.
#include <algorithm>

#include "pp_common.h"

namespace pp {

// pred / gt [N, K, 2] f32; mask [N, K] u8; norm [N, 2] f64 (AFTER the reference's "<= 0 -> 1e6" substitution, with
// `skip[n]` = 1 for instances that had an exact zero in their normalisation factor: heatmap.py:78-82);
// counts [2][K] int32 (hits, valid), zeroed by the caller's memset node; dist (optional) [K, N] f32, -1 = masked out.
template <typename T>   // coordinate type of pred / gt: the difference is taken in THAT type, as numpy does
__global__ __launch_bounds__(256) void pck_counts_kernel(const T *__restrict__ pred, const T *__restrict__ gt,
                                                         const unsigned char *__restrict__ mask,
                                                         const double *__restrict__ norm,
                                                         const unsigned char *__restrict__ skip, double thr, int N,
                                                         int K, int *__restrict__ counts, float *__restrict__ dist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int *lc = reinterpret_cast<int *>(smem);              // [2][K]
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x) lc[i] = 0;
  __syncthreads();
  const long long total = (long long)N * K;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    float d = -1.0f;
    if (mask[idx] && !skip[n]) {
      const T px = pred[2 * idx], py = pred[2 * idx + 1], gx = gt[2 * idx], gy = gt[2 * idx + 1];
      // (pred - gt) in the arrays' own type (float32 for decoded locations), the division by the float64 factor in
      // float64, np.linalg.norm(axis=-1) = sqrt(x0 * x0 + x1 * x1) in float64, rounded to float32 by the assignment
      const double q0 = (double)(T)(px - gx) / norm[2 * n], q1 = (double)(T)(py - gy) / norm[2 * n + 1];
      d = (float)sqrt(q0 * q0 + q1 * q1);
      atomicAdd(&lc[K + k], 1);                          // a computed distance is never -1: every unmasked pair is valid
      if ((double)d < thr) atomicAdd(&lc[k], 1);         // NaN: valid, never a hit (like `nan < thr`)
    }
    if (dist) dist[(size_t)k * N + n] = d;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * K; i += blockDim.x)
    if (lc[i]) atomicAdd(&counts[i], lc[i]);
}

// ---- get_heatmap_maximum (heatmap.py:13-52) for maps of ANY size: flat arg-max with numpy's semantics (the first
// maximum in row-major order wins; a NaN is the maximum, and the first NaN wins), vals = the maximum itself, the
// location (-1, -1) where it is <= 0.  One workgroup per map, 16-byte loads when the map allows, the map read once.
struct ArgBest {
  float v;
  int i;
};
__device__ __forceinline__ bool arg_better(float v, int i, const ArgBest &b) {   // is (v, i) ahead of b in np.argmax's order?
  const bool vn = v != v, bn = b.v != b.v;
  if (vn != bn) return vn;
  if (!vn && v != b.v) return v > b.v;
  return i < b.i;
}
__global__ __launch_bounds__(256) void heatmap_argmax_kernel(const float *__restrict__ hm, int HW, int W,
                                                             float *__restrict__ locs, float *__restrict__ vals) {
  const float *m = hm + (size_t)blockIdx.x * HW;
  ArgBest best{-__builtin_inff(), 0x7fffffff};
  if ((HW & 3) == 0 && (((uintptr_t)m) & 15) == 0) {
    for (int q = threadIdx.x; q < HW / 4; q += 256) {
      const float4 t = reinterpret_cast<const float4 *>(m)[q];
      const float e[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (arg_better(e[j], 4 * q + j, best)) best = ArgBest{e[j], 4 * q + j};
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += 256)
      if (arg_better(m[i], i, best)) best = ArgBest{m[i], i};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best.v, o, 64);
    const int oi = __shfl_xor(best.i, o, 64);
    if (arg_better(ov, oi, best)) best = ArgBest{ov, oi};
  }
  __shared__ ArgBest wb[4];
  if ((threadIdx.x & 63) == 0) wb[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (arg_better(wb[w].v, wb[w].i, best)) best = wb[w];
    const bool dead = best.v <= 0.0f;                      // NaN: not dead (like `nan <= 0`)
    locs[2 * blockIdx.x + 0] = dead ? -1.0f : (float)(best.i % W);
    locs[2 * blockIdx.x + 1] = dead ? -1.0f : (float)(best.i / W);
    vals[blockIdx.x] = best.v;
  }
}

}  // namespace pp

extern "C" int pp_pck_counts(const void *pred, const void *gt, int coord_f64, const unsigned char *mask,
                             const double *norm, const unsigned char *skip, double thr, int N, int K, int *counts,
                             float *dist, void *stream) {
  using namespace pp;
  PP_REQUIRE(N >= 0 && K >= 0, "pp_pck_counts: bad shape N=%d K=%d", N, K);
  PP_REQUIRE(K <= 8192, "pp_pck_counts: K=%d exceeds the 8192 keypoints the LDS counters hold", K);
  if (K == 0) return 0;
  PP_REQUIRE(counts, "pp_pck_counts: null counts");
  hipStream_t s = (hipStream_t)stream;
  PP_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int) * 2 * K, s));
  if (N == 0) return 0;
  PP_REQUIRE(pred && gt && mask && norm && skip, "pp_pck_counts: null operand");
  const long long total = (long long)N * K;
  const int grid = (int)std::min<long long>((total + 255) / 256, 2048);
  if (coord_f64)
    hipLaunchKernelGGL(pck_counts_kernel<double>, dim3(grid), dim3(256), sizeof(int) * 2 * K, s, (const double *)pred,
                       (const double *)gt, mask, norm, skip, thr, N, K, counts, dist);
  else
    hipLaunchKernelGGL(pck_counts_kernel<float>, dim3(grid), dim3(256), sizeof(int) * 2 * K, s, (const float *)pred,
                       (const float *)gt, mask, norm, skip, thr, N, K, counts, dist);
  PP_CHECK_LAUNCH("pck_counts_kernel");
  return 0;
}

extern "C" int pp_heatmap_argmax(const float *heatmaps, long long maps, int H, int W, float *locs, float *vals,
                                 void *stream) {
  using namespace pp;
  PP_REQUIRE(maps >= 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 31), "pp_heatmap_argmax: bad shape");
  if (maps == 0) return 0;
  PP_REQUIRE(heatmaps && locs && vals, "pp_heatmap_argmax: null operand");
  PP_REQUIRE(maps < (1ll << 31), "pp_heatmap_argmax: too many maps");
  hipLaunchKernelGGL(heatmap_argmax_kernel, dim3((unsigned)maps), dim3(256), 0, (hipStream_t)stream, heatmaps, H * W, W,
                     locs, vals);
  PP_CHECK_LAUNCH("heatmap_argmax_kernel");
  return 0;
}
