// Sparsemax normalisation of the heatmap rows (reference head.py:237-245, 526-532):
//
//     x = final_layer(...)                      (B, K, H*W)
//     x = Sparsemax(dim=-1)(x / temperature)    third-party sparsemax==0.1.9, not in the reference checkout
//     x = clamp(x * normalize, 0, 1)
//
// Sparsemax (Martins & Astudillo 2016) is the Euclidean projection of a row z onto the probability simplex:
// p = max(z - tau, 0) with the unique tau for which sum(p) = 1.  The library's forward shifts the row by its
// maximum, sorts it, takes k = max{ j : 1 + j z_(j) > sum_{i<=j} z_(i) } and tau = (sum_{i<=k} z_(i) - 1) / k.
// The same tau is the fixed point of Michelot's iteration
//     tau <- (sum_{z_i > tau} z_i - 1) / #{ z_i > tau },   tau_0 = (sum z - 1) / n,
// which is non-decreasing, never passes tau* and stops after finitely many steps with exactly the sorted
// algorithm's support -- no sort.  One workgroup per (crop, keypoint) row: the row (3 072 or 6 912 floats for
// the 64x48 / 96x72 maps) is read from HBM once into LDS, shifted by its maximum in float32 as the library does,
// the sums run in float64 (the library accumulates float32 cumulative sums; its result differs from the exact
// projection by that rounding, ours by none), and the row is written back once:  HBM-bound, 8 B per pixel.
#include "pp_common.h"

namespace pp {

constexpr int SM_THREADS = 256;
constexpr int SM_MAX_ITERS = 64;     // Michelot steps before the bisection fallback (typical rows need 3-8)

struct SumCount {
  double s;
  int c;
};

__device__ __forceinline__ SumCount block_sum_count(double s, int c, double *red_s, int *red_c) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                       // previous round's readers are done with red_*
  if (lane == 0) {
    red_s[wave] = s;
    red_c[wave] = c;
  }
  __syncthreads();
  SumCount r{0.0, 0};
#pragma unroll
  for (int w = 0; w < SM_THREADS / 64; ++w) {
    r.s += red_s[w];
    r.c += red_c[w];
  }
  return r;
}

// ROW_IN_LDS: the shifted row lives in LDS; otherwise (rows beyond 160 KB) it is re-read from the output buffer,
// which holds the shifted row between the passes.
template <bool ROW_IN_LDS>
__global__ __launch_bounds__(SM_THREADS) void sparsemax_rows_kernel(float *__restrict__ x, int n, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red_s[SM_THREADS / 64];
  __shared__ int red_c[SM_THREADS / 64];
  __shared__ float red_m[SM_THREADS / 64];
  float *row = ROW_IN_LDS ? reinterpret_cast<float *>(smem) : x + (size_t)blockIdx.x * n;
  float *g = x + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  // pass 1: load + row maximum
  float m = -INFINITY;
  if ((n & 3) == 0) {
    for (int i = tid * 4; i < n; i += SM_THREADS * 4) {
      const float4 v = *reinterpret_cast<const float4 *>(g + i);
      if (ROW_IN_LDS) *reinterpret_cast<float4 *>(row + i) = v;
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  } else {
    for (int i = tid; i < n; i += SM_THREADS) {
      const float v = g[i];
      if (ROW_IN_LDS) row[i] = v;
      m = fmaxf(m, v);
    }
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red_m[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  // pass 2: shift by the maximum (float32, as the library: input - max) and the plain sum
  double s = 0.0;
  for (int i = tid; i < n; i += SM_THREADS) {
    const float z = row[i] - m;
    row[i] = z;
    s += (double)z;
  }
  SumCount sc = block_sum_count(s, 0, red_s, red_c);
  double tau = (sc.s - 1.0) / (double)n;
  int count = n;
  bool done = false;
  for (int it = 0; it < SM_MAX_ITERS; ++it) {
    double ps = 0.0;
    int pc = 0;
    for (int i = tid; i < n; i += SM_THREADS) {
      const double z = (double)row[i];
      if (z > tau) {
        ps += z;
        ++pc;
      }
    }
    sc = block_sum_count(ps, pc, red_s, red_c);
    if (sc.c == 0) break;                // NaN rows: leave tau as it is (the output is NaN like the library's)
    tau = (sc.s - 1.0) / (double)sc.c;
    if (sc.c == count) {
      done = true;
      break;
    }
    count = sc.c;
  }
  if (!done && count > 0) {
    // not converged in SM_MAX_ITERS steps (adversarial rows): bisection on f(tau) = sum max(z - tau, 0) - 1 over
    // [-1, 0] (the shifted maximum is 0, so tau* lies there), then the closed form on the support found
    double lo = -1.0, hi = 0.0;
    for (int it = 0; it < 48; ++it) {
      const double mid = 0.5 * (lo + hi);
      double ps = 0.0;
      for (int i = tid; i < n; i += SM_THREADS) {
        const double d = (double)row[i] - mid;
        if (d > 0.0) ps += d;
      }
      sc = block_sum_count(ps, 0, red_s, red_c);
      if (sc.s > 1.0) lo = mid; else hi = mid;
    }
    double ps = 0.0;
    int pc = 0;
    for (int i = tid; i < n; i += SM_THREADS) {
      const double z = (double)row[i];
      if (z > lo) {
        ps += z;
        ++pc;
      }
    }
    sc = block_sum_count(ps, pc, red_s, red_c);
    if (sc.c > 0) tau = (sc.s - 1.0) / (double)sc.c;
  }
  const float tf = (float)tau;
  // pass 3: p = max(z - tau, 0) in float32 (the library: max(0, input - taus)), * normalize, clamp (head.py:530-532)
  if ((n & 3) == 0) {
    for (int i = tid * 4; i < n; i += SM_THREADS * 4) {
      const float4 z = *reinterpret_cast<const float4 *>(row + i);
      float4 o;
      o.x = fminf(fmaxf(fmaxf(z.x - tf, 0.f) * scale, 0.f), 1.f);
      o.y = fminf(fmaxf(fmaxf(z.y - tf, 0.f) * scale, 0.f), 1.f);
      o.z = fminf(fmaxf(fmaxf(z.z - tf, 0.f) * scale, 0.f), 1.f);
      o.w = fminf(fmaxf(fmaxf(z.w - tf, 0.f) * scale, 0.f), 1.f);
      *reinterpret_cast<float4 *>(g + i) = o;
    }
  } else {
    for (int i = tid; i < n; i += SM_THREADS) g[i] = fminf(fmaxf(fmaxf(row[i] - tf, 0.f) * scale, 0.f), 1.f);
  }
}

}  // namespace pp

extern "C" int pp_sparsemax_rows(float *x, long long rows, int n, float scale, void *stream) {
  using namespace pp;
  PP_REQUIRE(rows >= 0 && n > 0, "pp_sparsemax_rows: bad shape rows=%lld n=%d", rows, n);
  if (rows == 0) return 0;
  PP_REQUIRE(x, "pp_sparsemax_rows: null pointer");
  PP_REQUIRE(rows < (1ll << 31), "pp_sparsemax_rows: too many rows");
  PP_REQUIRE(((uintptr_t)x & 15) == 0, "pp_sparsemax_rows: x must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)n * 4;
  if (lds <= 150 * 1024) {
    static thread_local unsigned long long attr_mask = 0;
    int dev_ = 0;
    if (attr_needed(attr_mask, dev_))
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sparsemax_rows_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(sparsemax_rows_kernel<true>, dim3((unsigned)rows), dim3(SM_THREADS), lds, s, x, n, scale);
  } else {
    hipLaunchKernelGGL(sparsemax_rows_kernel<false>, dim3((unsigned)rows), dim3(SM_THREADS), 0, s, x, n, scale);
  }
  PP_CHECK_LAUNCH("sparsemax_rows_kernel");
  return 0;
}
