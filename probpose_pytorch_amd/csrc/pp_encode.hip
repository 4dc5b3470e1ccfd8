// Training-target generation on the GPU: the OKS probability maps of ProbMap.encode.
// Replaces generate_probmaps (reference codec.py:11-70, called by ProbMap.encode codec.py:176-182), batched
// over crops: for every visible keypoint the map exp(-dist^2 / (2 s)) with dist = sqrt(dx^2 + dy^2),
// dx = x - kx, dy = y - ky evaluated in float64 exactly as numpy does (the keypoint is a float32 value,
// the grid index an int64: the difference is float64; dist**2 is a multiply; the map is cast to float32
// when stored, codec.py:69) and the weight (map.max() > 0) taken on the float64 map.  A memory-bound write
// kernel: K*H*W*4 bytes per crop leave, 16*K bytes enter.
// float64 exp: the device library's and numpy's agree to <= 1 ulp of float64, so the float32 maps are
// identical except where the float64 value sits within 2^-29 of a float32 rounding boundary
// (the goldens minted from the reference are reproduced bit for bit).
#include "pp_common.h"

namespace pp {

__global__ __launch_bounds__(256) void encode_probmaps_kernel(const float *__restrict__ kpts,
                                                              const float *__restrict__ visible,
                                                              const double *__restrict__ two_s, int K, int H, int W,
                                                              float *__restrict__ heatmaps,
                                                              float *__restrict__ weights) {
  const int map = blockIdx.x, k = map % K;
  const int HW = H * W;
  float *o = heatmaps + (size_t)map * HW;
  const float vis = visible[map];
  __shared__ double red[4];
  if (vis < 0.5f) {   // codec.py:53-54: unlabelled keypoints keep a zero map and their visibility as weight
    for (int p = threadIdx.x; p < HW; p += 256) o[p] = 0.f;
    if (threadIdx.x == 0) weights[map] = vis;
    return;
  }
  const double kx = (double)kpts[2 * map], ky = (double)kpts[2 * map + 1], den = two_s[k];
  double best = 0.0;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, x = p - y * W;
    const double dx = (double)x - kx, dy = (double)y - ky;
    const double dist = sqrt(dx * dx + dy * dy);
    const double oks = exp(-((dist * dist) / den));
    best = fmax(best, oks);
    o[p] = (float)oks;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) weights[map] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) > 0.0 ? 1.f : 0.f;
}

}  // namespace pp

extern "C" int pp_encode_probmaps(const float *kpts_hm, const float *visible, const double *two_s, int B, int K, int H,
                                  int W, float *heatmaps, float *weights, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), "pp_encode_probmaps: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(kpts_hm && visible && two_s && heatmaps && weights, "pp_encode_probmaps: null pointer");
  PP_REQUIRE((long long)B * K < (1ll << 31), "pp_encode_probmaps: too many maps");
  hipLaunchKernelGGL(encode_probmaps_kernel, dim3((unsigned)(B * K)), dim3(256), 0, (hipStream_t)stream, kpts_hm,
                     visible, two_s, K, H, W, heatmaps, weights);
  PP_CHECK_LAUNCH("encode_probmaps_kernel");
  return 0;
}
