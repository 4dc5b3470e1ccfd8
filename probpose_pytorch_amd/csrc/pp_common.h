// Shared helpers for the gfx950 kernels of libprobpose_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/probpose_hip.h"

namespace pp {

// thread-local error string behind pp_last_error()
char *err_buf();
int fail(const char *fmt, ...);

#define PP_CHECK_HIP(expr)                                                            \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) return pp::fail("%s: %s", #expr, hipGetErrorString(_e));    \
  } while (0)

#define PP_CHECK_LAUNCH(name)                                                         \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess) return pp::fail("launch %s: %s", name, hipGetErrorString(_e)); \
  } while (0)

#define PP_REQUIRE(cond, ...)                                                         \
  do {                                                                                \
    if (!(cond)) return pp::fail(__VA_ARGS__);                                        \
  } while (0)

// ---- storage types -------------------------------------------------------
typedef unsigned short bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __hip_bfloat16 h = __float2bfloat16(f);
  return *reinterpret_cast<bf16_t *>(&h);
}

// two floats -> one packed bf16 pair by ONE v_cvt_pk_bf16_f32 (the same rounding as f32_to_bf16, which the compiler
// lowers to that instruction with one live input, a shift and an or around it)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

template <typename T> struct Store;
template <> struct Store<float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Store<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute (dynamic-LDS limit) is per device: the "already raised" caches are bitmasks over the device
// ordinal of the calling thread's current device, not a single flag (a second GPU in the same thread would otherwise
// launch with the 64 KB default limit and fail).
inline bool attr_needed(unsigned long long &mask, int &dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  dev_out = dev;
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

}  // namespace pp
