// Shared by the GEMM translation units (pp_gemm.hip, pp_gemm_quad.hip): launch parameters and the small device
// helpers every kernel form uses (counted waits, LDS-DMA issue, GELU).  Device code only; no state.
#pragma once
#include "pp_common.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct GemmParams {
  const char *A;
  const char *W;
  char *C;
  const float *bias;
  const float *residual;
  const float *rowbias;
  const int32_t *rowoff;
  const int32_t *out_rowmap;
  int M, N, Kd;
  int lda, ldw, ldc;
  int seg_len, rowbias_period;
  long long strideA, strideW, strideC, strideBias, strideRowoff, strideRowmap;
  int splitk;        // K splits per batch entry (grid.y = batch * splitk)
  long long strideA_k, strideW_k, strideC_k, strideRowoff_k;
  int epilogue;
  int hm_K, hm_HW;
  float hm_temperature;
  int tiles_m, tiles_n;
  const float *colsum;
  float out_scale;   // fp8 output: value * out_scale is what gets rounded to e4m3
  int blocked;       // XCD-blocked tile order (large grids) vs plain order
  int rn;            // column tiles per XCD block (<= tiles_n, so narrow-N launches carry no empty slots)
  int lds_epilogue;  // bf16 C tile staged through LDS and stored as whole rows
  const char *final_w;     // PP_EPI_FUSE_FINAL: [hm_K, N] storage-dtype weights of the 1x1 heatmap layer
  const float *final_b;    // [hm_K]
};

// LDS-DMA of 16 B per lane: LDS destination = wave-uniform byte offset (M0) + lane * 16.  Issued
// from inline asm on purpose: hipcc cannot tell that the DMA into buffer t+1 never aliases the
// ds_reads of buffer t and would drain vmcnt(0) in front of every fragment read, serialising the
// prefetch behind the MFMAs.  The asm DMA is invisible to its wait-count bookkeeping; completion is
// enforced by the explicit s_waitcnt vmcnt(0) + barrier that ends each K-step.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_off_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_off_uniform)
      : "memory");
}

__device__ __forceinline__ unsigned lds_offset_of(const void *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)p;
}

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// bf16 epilogues only: GELU(x) = x * Phi(x) with Phi(x) ~ sigmoid(x * (c0 + c1 t + c2 t^2)), t = min(x^2, 50),
// coefficients fitted (minimax) against the exact-erf GELU: |error| <= 3.0e-5 absolute over all x (the tanh
// form's is 4.7e-4), i.e. below half a bf16 ulp of every output with |GELU| > 0.016 and an absolute 3e-5
// for the rest; the output is rounded to bf16 right after.  7 VALU + exp + rcp per element instead of the 16 +
// exp + rcp of an Abramowitz-Stegun erf: the GELU arithmetic was 15 us of a 100 us fc1 GEMM (measured).
// The constants carry the factor -log2(e) so the sigmoid is 1 / (1 + exp2(x * p)).
// Two elements per call: the polynomial runs on packed-fp32 instructions (v_pk_mul/fma/add_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 t = __builtin_elementwise_min(x * x, (f32x2){50.0f, 50.0f});
  const f32x2 c2 = {0.001035082619637251f, 0.001035082619637251f};
  const f32x2 c1 = {-0.10690470039844513f, -0.10690470039844513f};
  const f32x2 c0 = {-2.300978660583496f, -2.300978660583496f};
  const f32x2 u = x * __builtin_elementwise_fma(__builtin_elementwise_fma(c2, t, c1), t, c0);
  const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)} + (f32x2){1.0f, 1.0f};
  return x * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// N values at once, stage by stage: N / 2 independent chains side by side.  One chain alone is a string of dependent
// packed / transcendental instructions that hipcc pads with an s_nop per link (seen in the ISA: 591 nops in the GELU of a
// 128 x 96 wave tile, twice the time of the arithmetic).
template <int N>
__device__ __forceinline__ void gelu_fast_n(float (&v)[N]) {
  static_assert(N % 2 == 0, "pairs");
  const f32x2 c2 = {0.001035082619637251f, 0.001035082619637251f};
  const f32x2 c1 = {-0.10690470039844513f, -0.10690470039844513f};
  const f32x2 c0 = {-2.300978660583496f, -2.300978660583496f};
  f32x2 x[N / 2], t[N / 2];
#pragma unroll
  for (int k = 0; k < N / 2; ++k) {
    x[k] = (f32x2){v[2 * k], v[2 * k + 1]};
    t[k] = __builtin_elementwise_min(x[k] * x[k], (f32x2){50.0f, 50.0f});
  }
#pragma unroll
  for (int k = 0; k < N / 2; ++k) t[k] = __builtin_elementwise_fma(__builtin_elementwise_fma(c2, t[k], c1), t[k], c0);
#pragma unroll
  for (int k = 0; k < N / 2; ++k) t[k] = x[k] * t[k];
#pragma unroll
  for (int k = 0; k < N / 2; ++k) t[k] = (f32x2){__builtin_amdgcn_exp2f(t[k].x), __builtin_amdgcn_exp2f(t[k].y)};
#pragma unroll
  for (int k = 0; k < N / 2; ++k) t[k] = t[k] + (f32x2){1.0f, 1.0f};
#pragma unroll
  for (int k = 0; k < N / 2; ++k) t[k] = (f32x2){__builtin_amdgcn_rcpf(t[k].x), __builtin_amdgcn_rcpf(t[k].y)};
#pragma unroll
  for (int k = 0; k < N / 2; ++k) {
    const f32x2 r = x[k] * t[k];
    v[2 * k] = r.x;
    v[2 * k + 1] = r.y;
  }
}
template <typename T>
__device__ __forceinline__ void gelu4(float (&v)[4]) {
  if constexpr (sizeof(T) <= 2) {
    const f32x2 a = gelu_fast2((f32x2){v[0], v[1]}), b = gelu_fast2((f32x2){v[2], v[3]});
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
  }
}

}  // namespace pp
