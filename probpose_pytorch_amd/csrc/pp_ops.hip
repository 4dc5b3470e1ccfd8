// Memory-bound companions of the MFMA GEMM on the ProbPose forward path:
// LayerNorm, patch im2col, max-pool+ReLU, the aux-branch 1x1 tail, layout
// transposes, and the exact-fp32 VALU attention used in parity mode (the bf16
// MFMA attention lives in pp_attention.hip).
#include <algorithm>

#include "pp_common.h"

namespace pp {

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row, the row held in registers (C <= 2048) so HBM/L2
// is read once; fp32 statistics (two-pass, like torch's RowwiseMoments result).
// ---------------------------------------------------------------------------
// T = unsigned char: the output is OCP e4m3 of value * qscale (static per-tensor scale), saturating at 448.
__device__ __forceinline__ unsigned ln_pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
  b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
  c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
  d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps,
                                                        int rows, int C, T *__restrict__ out, float qscale = 1.0f) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *xr = x + (size_t)row * C;
  T *orow = out + (size_t)row * C;
  constexpr int MAXV = 8;
  if ((C & 3) == 0 && C <= MAXV * 256) {
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) {
        v[i] = *reinterpret_cast<const float4 *>(xr + c);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) {
        const float4 g = *reinterpret_cast<const float4 *>(gamma + c);
        const float4 b = *reinterpret_cast<const float4 *>(beta + c);
        const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y,
                    o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4 *>(orow + c) = make_float4(o0, o1, o2, o3);
        } else if constexpr (sizeof(T) == 1) {
          *reinterpret_cast<unsigned *>(orow + c) = ln_pack_fp8x4(o0 * qscale, o1 * qscale, o2 * qscale, o3 * qscale);
        } else {
          ushort4 pk;
          pk.x = f32_to_bf16(o0); pk.y = f32_to_bf16(o1); pk.z = f32_to_bf16(o2); pk.w = f32_to_bf16(o3);
          *reinterpret_cast<ushort4 *>(orow + c) = pk;
        }
      }
    }
    return;
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = xr[c] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  if constexpr (sizeof(T) != 1)   // fp8 output is launched for C % 4 == 0, C <= 2048 only (checked on the host)
    for (int c = lane; c < C; c += 64) Store<T>::st(orow + c, (xr[c] - mean) * rstd * gamma[c] + beta[c]);
}

// Multi-row form (C % 4 == 0, C <= NV * 256): a wave normalises RPW consecutive rows and issues ALL their loads
// before the first reduction (RPW x NV x 16 B per lane in flight), and the launch is sized in whole rounds of
// workgroups (ViT-B bs 64: 12 288 rows = 256 CUs x 4 workgroups x 4 waves x 3 rows).  The one-row form above ran
// 3 072 workgroups on 2 048 slots -- 1.5 rounds, the second half empty -- at 3.8 TB/s; the arithmetic (two-pass
// statistics in the same order) is unchanged, so results are bit-identical.
template <typename T, int RPW, int NV>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, int rows,
                                                             int C, T *__restrict__ out, float qscale) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  float4 v[RPW][NV];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const float *xr = x + (size_t)min(row0 + r, rows - 1) * C;     // tail rows re-read the last row (not stored)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) v[r][i] = *reinterpret_cast<const float4 *>(xr + c);
    }
  }
  float4 g[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      g[i] = *reinterpret_cast<const float4 *>(gamma + c);
      bt[i] = *reinterpret_cast<const float4 *>(beta + c);
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) {
        const float a = v[r][i].x - mean, b = v[r][i].y - mean, cc = v[r][i].z - mean, d = v[r][i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (row0 + r >= rows) continue;
    T *orow = out + (size_t)(row0 + r) * C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < C) {
        const float o0 = (v[r][i].x - mean) * rstd * g[i].x + bt[i].x, o1 = (v[r][i].y - mean) * rstd * g[i].y + bt[i].y,
                    o2 = (v[r][i].z - mean) * rstd * g[i].z + bt[i].z, o3 = (v[r][i].w - mean) * rstd * g[i].w + bt[i].w;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<float4 *>(orow + c) = make_float4(o0, o1, o2, o3);
        } else if constexpr (sizeof(T) == 1) {
          *reinterpret_cast<unsigned *>(orow + c) = ln_pack_fp8x4(o0 * qscale, o1 * qscale, o2 * qscale, o3 * qscale);
        } else {
          ushort4 pk;
          pk.x = f32_to_bf16(o0); pk.y = f32_to_bf16(o1); pk.z = f32_to_bf16(o2); pk.w = f32_to_bf16(o3);
          *reinterpret_cast<ushort4 *>(orow + c) = pk;
        }
      }
    }
  }
}

template <typename T>
static bool layernorm_rows_launch(const float *x, const float *gamma, const float *beta, float eps, int rows, int C,
                                  T *out, float qscale, hipStream_t s) {
  if ((C & 3) != 0 || C > 1280 || C <= 512 || rows < 2048) return false;   // narrow rows: the one-row form is faster (measured at C = 384)
#ifdef PP_LN_ONE_ROW
  return false;
#endif
  // rows per wave: as close as the variants allow to one round of 4 workgroups (16 waves) per CU on 256 CUs
  const int want = (rows + 4095) / 4096;
  const int rpw = want >= 3 ? 3 : (want >= 2 ? 2 : 1);
  const int grid = cdiv(rows, 4 * rpw);
#define PP_LN_ROWS(RPW_, NV_)                                                                                     \
  hipLaunchKernelGGL((layernorm_rows_kernel<T, RPW_, NV_>), dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, rows, C, \
                     out, qscale)
  if (C <= 768) {
    if (rpw == 3) PP_LN_ROWS(3, 3); else if (rpw == 2) PP_LN_ROWS(2, 3); else PP_LN_ROWS(1, 3);
  } else {
    if (rpw == 3) PP_LN_ROWS(3, 5); else if (rpw == 2) PP_LN_ROWS(2, 5); else PP_LN_ROWS(1, 5);
  }
#undef PP_LN_ROWS
  return true;
}

// ---------------------------------------------------------------------------
// Patch im2col (+cast): NCHW fp32 image -> rows of 3*p*p, k = c*p*p + py*p + px.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float *__restrict__ x, T *__restrict__ out,
                                                       int B, int H, int W, int p) {
  const int W4 = W >> 2;
  const long long total = (long long)B * 3 * H * W4;
  const int gw = W / p, gh = H / p;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xq = (int)(i % W4);
    long long t = i / W4;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % 3);
    const int b = (int)(t / 3);
    const int x0 = xq * 4;
    const int gy = y / p, py = y - gy * p, gx = x0 / p, px = x0 - gx * p;
    if (gy >= gh || gx >= gw) continue;  // pixels beyond the last full patch are dropped (conv stride p)
    const float4 v = *reinterpret_cast<const float4 *>(x + (((size_t)b * 3 + c) * H + y) * W + x0);
    T *o = out + ((size_t)(b * gh + gy) * gw + gx) * (3 * p * p) + (c * p + py) * p + px;
    Store<T>::st(o + 0, v.x); Store<T>::st(o + 1, v.y); Store<T>::st(o + 2, v.z); Store<T>::st(o + 3, v.w);
  }
}

// ---------------------------------------------------------------------------
// MaxPool(kh,kw) stride (kh,kw) + ReLU, channels-last rows, 16 B per lane.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_relu_kernel(const T *__restrict__ x, T *__restrict__ out,
                                                           int B, int h, int w, int C, int kh, int kw) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int oh = h / kh, ow = w / kw, CV = C / VEC;
  const long long total = (long long)B * oh * ow * CV;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long long t = i / CV;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int b = (int)(t / oh);
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = 0.f;  // ReLU folded in: max(0, max(window))
    for (int dy = 0; dy < kh; ++dy)
      for (int dx = 0; dx < kw; ++dx) {
        const T *src = x + (((size_t)b * h + oy * kh + dy) * w + ox * kw + dx) * C + cv * VEC;
        const uint4 raw = *reinterpret_cast<const uint4 *>(src);
        const T *e8 = reinterpret_cast<const T *>(&raw);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float f = Store<T>::ld(e8 + e);
          m[e] = (f > m[e] || f != f) ? f : m[e];  // NaN propagates like torch's max_pool2d
        }
      }
    uint4 pk;
    T *o8 = reinterpret_cast<T *>(&pk);
#pragma unroll
    for (int e = 0; e < VEC; ++e) Store<T>::st(o8 + e, m[e]);
    *reinterpret_cast<uint4 *>(out + (((size_t)b * oh + oy) * ow + ox) * C + cv * VEC) = pk;
  }
}

// ---------------------------------------------------------------------------
// Split-K consumer: the same pooling on the SUM of nsplit f32 partial maps (+ bias), 16 B (4 channels) per lane.
// The aux convolutions after the first pooling have few rows (M = 16 B, 4 B) and K = 9 C: as one GEMM they fill a
// fraction of the chip with 108 sequential K-tiles per workgroup (measured 104 / 99 us at 418 / 109 TFLOP/s); split
// over K into workgroups their partials are reduced here, where the map is read anyway.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_relu_sum_kernel(const float *__restrict__ x, int nsplit,
                                                               long long split_stride, const float *__restrict__ bias,
                                                               T *__restrict__ out, int B, int h, int w, int C, int kh,
                                                               int kw) {
  const int oh = h / kh, ow = w / kw, CV = C / 4;
  const long long total = (long long)B * oh * ow * CV;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long long t = i / CV;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int b = (int)(t / oh);
    const float4 bv = *reinterpret_cast<const float4 *>(bias + cv * 4);
    float m[4] = {0.f, 0.f, 0.f, 0.f};                 // ReLU folded in: max(0, max(window))
    for (int dy = 0; dy < kh; ++dy)
      for (int dx = 0; dx < kw; ++dx) {
        const float *src = x + (((size_t)b * h + oy * kh + dy) * w + ox * kw + dx) * C + cv * 4;
        float4 a = *reinterpret_cast<const float4 *>(src);
        for (int sp = 1; sp < nsplit; ++sp) {          // split order 0, 1, 2, ...: a fixed summation order
          const float4 q = *reinterpret_cast<const float4 *>(src + (size_t)sp * split_stride);
          a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
        const float f[4] = {a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = (f[e] > m[e] || f[e] != f[e]) ? f[e] : m[e];
      }
    T *o = out + (((size_t)b * oh + oy) * ow + ox) * C + cv * 4;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4 *>(o) = make_float4(m[0], m[1], m[2], m[3]);
    } else {
      ushort4 pk;
      pk.x = f32_to_bf16(m[0]); pk.y = f32_to_bf16(m[1]); pk.z = f32_to_bf16(m[2]); pk.w = f32_to_bf16(m[3]);
      *reinterpret_cast<ushort4 *>(o) = pk;
    }
  }
}

// ---------------------------------------------------------------------------
// Aux tail: per branch 1x1 conv C->K on the pooled 1x1 feature + Sigmoid/ReLU.
// x [B, nbr*C] (branch-major columns), w [nbr][K][C], bias [nbr][K], out [nbr][B][K].
// One wave per output value.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void aux_tail_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                       const float *__restrict__ bias,
                                                       float *__restrict__ out, int B, int C, int K,
                                                       int nbr, unsigned relu_mask) {
  const int lane = threadIdx.x & 63;
  const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long long)nbr * B * K) return;
  const int k = (int)(o % K);
  const int b = (int)((o / K) % B);
  const int br = (int)(o / ((long long)K * B));
  const T *xr = x + ((size_t)b * nbr + br) * C;
  const T *wr = w + ((size_t)br * K + k) * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += Store<T>::ld(xr + c) * Store<T>::ld(wr + c);
  s = wave_sum(s) + bias[br * K + k];
  if (lane == 0) out[o] = ((relu_mask >> br) & 1) ? fmaxf(s, 0.f) : 1.0f / (1.0f + expf(-s));
}

// ---------------------------------------------------------------------------
// Final 1x1 heatmap layer (head.py:525-532): heat[b,k,hw] = clamp((x[b*HW+hw,:] . w[k,:] + bias[k]) / T, 0, 1).
// K is tiny (17) and the layer reads 100 MB of activations: HBM-bound, so a streaming kernel (64
// pixel rows per workgroup staged once through LDS, weights broadcast from LDS, NCHW float32 stores
// that are 256-B contiguous per k) instead of a 128-wide MFMA tile that would be 87 % padding.
// ---------------------------------------------------------------------------
constexpr int FH_ROWS = 64;
// CLAMP = false: the Sparsemax path (head.py:526 with normalize != None) takes the logits z / T unclamped
template <bool CLAMP>
__device__ __forceinline__ float fh_act(float v) {
  return CLAMP ? fminf(fmaxf(v, 0.f), 1.f) : v;
}

template <typename T, bool CLAMP>
__global__ __launch_bounds__(256) void final_heatmap_kernel(const T *__restrict__ x, const T *__restrict__ w,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ out, int M, int HW, int Cin,
                                                            int K, float temperature) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = 16 / (int)sizeof(T);       // elements per 16-B chunk
  const int chunks = Cin / VEC;                  // per row
  const int xstride = (chunks + 1) * 16;         // +1 chunk pad: 64 rows hit distinct bank slots
  char *xs = smem;                               // [FH_ROWS][xstride]
  char *ws = smem + FH_ROWS * xstride;           // [K][chunks*16]
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * FH_ROWS;
  for (int i = tid; i < FH_ROWS * chunks; i += 256) {
    const int r = i / chunks, c = i - r * chunks;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m0 + r < M) v = *reinterpret_cast<const uint4 *>(x + (size_t)(m0 + r) * Cin + c * VEC);
    *reinterpret_cast<uint4 *>(xs + r * xstride + c * 16) = v;
  }
  for (int i = tid; i < K * chunks; i += 256)
    *reinterpret_cast<uint4 *>(ws + (size_t)i * 16) = *reinterpret_cast<const uint4 *>(w + (size_t)i * VEC);
  __syncthreads();
  const int r = tid & 63, kg = tid >> 6;
  const int m = m0 + r;
  constexpr int KB = 5;                           // k values per pass per thread (k = kg + 4*u)
  for (int kbase = kg; kbase < K; kbase += 4 * KB) {
    float acc[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) acc[u] = 0.f;
    for (int c = 0; c < chunks; ++c) {
      const uint4 xv = *reinterpret_cast<const uint4 *>(xs + r * xstride + c * 16);
      const T *xe = reinterpret_cast<const T *>(&xv);
      float xf[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) xf[e] = Store<T>::ld(xe + e);
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int k = kbase + 4 * u;
        if (k < K) {
          const uint4 wv = *reinterpret_cast<const uint4 *>(ws + ((size_t)k * chunks + c) * 16);
          const T *we = reinterpret_cast<const T *>(&wv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[u] = fmaf(xf[e], Store<T>::ld(we + e), acc[u]);
        }
      }
    }
    if (m < M) {
      const int b = m / HW, hw = m - b * HW;
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int k = kbase + 4 * u;
        if (k < K) out[((size_t)b * K + k) * HW + hw] = fh_act<CLAMP>((acc[u] + bias[k]) / temperature);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 form on the matrix cores (HW % 16 == 0, Cin % 32 == 0): the VALU form above spends ~50 us of its 77 us
// on bf16 -> f32 conversions and FMAs for ViT-B bs 64; here a wave takes 16 pixel rows at a time, loads their
// Cin values straight from HBM into MFMA A fragments (lane (row, kq): 16 B at k = 32 s + 8 kq; the next
// tile's loads are issued before the current tile's MFMAs), multiplies by W^T held in LDS (rows padded by 16 B
// against bank conflicts) and stores float4 = 4 consecutive pixels of one keypoint map: the accumulator of
// v_mfma_f32_16x16x32_bf16(A = pixels, B = maps) holds rows 4 (lane >> 4) + r of column lane & 15.
// ---------------------------------------------------------------------------
typedef __bf16 fh_bf16x8 __attribute__((ext_vector_type(8)));
typedef float fh_f32x4 __attribute__((ext_vector_type(4)));
constexpr int FHM_MAXKS = 16;   // Cin <= 512

template <int NT, bool CLAMP>
__global__ __launch_bounds__(256) void final_heatmap_mfma_kernel(const bf16_t *__restrict__ x,
                                                                 const bf16_t *__restrict__ w,
                                                                 const float *__restrict__ bias,
                                                                 float *__restrict__ out, int tiles, int HW, int Cin,
                                                                 int K, float temperature) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // W^T: [NT * 16][Cin * 2 + 16] bytes
  const int wstride = Cin * 2 + 16;
  const int chunks = Cin / 8, ks = Cin / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < NT * 16 * chunks; i += 256) {
    const int n = i / chunks, c = i - n * chunks;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < K) v = *reinterpret_cast<const uint4 *>(w + (size_t)n * Cin + c * 8);
    *reinterpret_cast<uint4 *>(smem + n * wstride + c * 16) = v;
  }
  __syncthreads();
  const int row = lane & 15, kq = lane >> 4;
  const int wave = blockIdx.x * 4 + (tid >> 6), nwaves = gridDim.x * 4;
  float bn[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bn[j] = (j * 16 + row < K) ? bias[j * 16 + row] : 0.f;
  uint4 cur[FHM_MAXKS], nxt[FHM_MAXKS];
  int tile = wave;
  if (tile < tiles) {
    const bf16_t *xr = x + ((size_t)tile * 16 + row) * Cin + kq * 8;
#pragma unroll
    for (int s_ = 0; s_ < FHM_MAXKS; ++s_)
      if (s_ < ks) cur[s_] = *reinterpret_cast<const uint4 *>(xr + s_ * 32);
  }
  for (; tile < tiles; tile += nwaves) {
    const int nt = tile + nwaves;
    if (nt < tiles) {
      const bf16_t *xr = x + ((size_t)nt * 16 + row) * Cin + kq * 8;
#pragma unroll
      for (int s_ = 0; s_ < FHM_MAXKS; ++s_)
        if (s_ < ks) nxt[s_] = *reinterpret_cast<const uint4 *>(xr + s_ * 32);
    }
    fh_f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = fh_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < FHM_MAXKS; ++s_) {
      if (s_ < ks) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const uint4 wf = *reinterpret_cast<const uint4 *>(smem + (j * 16 + row) * wstride + (4 * s_ + kq) * 16);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const fh_bf16x8 *>(&cur[s_]),
                                                           *reinterpret_cast<const fh_bf16x8 *>(&wf), acc[j], 0, 0, 0);
        }
      }
    }
    const long long m0 = (long long)tile * 16;           // 16 | HW: the tile lies inside one crop
    const int b = (int)(m0 / HW), hw = (int)(m0 - (long long)b * HW) + 4 * kq;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = j * 16 + row;
      if (n < K) {
        float4 o;
        o.x = fh_act<CLAMP>((acc[j][0] + bn[j]) / temperature);
        o.y = fh_act<CLAMP>((acc[j][1] + bn[j]) / temperature);
        o.z = fh_act<CLAMP>((acc[j][2] + bn[j]) / temperature);
        o.w = fh_act<CLAMP>((acc[j][3] + bn[j]) / temperature);
        *reinterpret_cast<float4 *>(out + ((size_t)b * K + n) * HW + hw) = o;
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < FHM_MAXKS; ++s_) cur[s_] = nxt[s_];
  }
}

// ---------------------------------------------------------------------------
// Batched transpose with dtype change: in [B, R, S] -> out [B, S, R].
// ---------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI *__restrict__ in, TO *__restrict__ out,
                                                        int R, int S) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const TI *ib = in + (size_t)b * R * S;
  TO *ob = out + (size_t)b * R * S;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && s0 + tx < S) tile[i][tx] = Store<TI>::ld(ib + (size_t)(r0 + i) * S + s0 + tx);
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (s0 + i < S && r0 + tx < R) Store<TO>::st(ob + (size_t)(s0 + i) * R + r0 + tx, tile[tx][i]);
}

// ---------------------------------------------------------------------------
// Exact-fp32 attention on the VALU (parity mode and head dims the MFMA kernel
// does not cover).  One workgroup per (crop, head); one query row per thread;
// K/V staged through LDS as fp32 in chunks; online softmax in groups of 8 keys.
// ---------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void attention_valu_kernel(const T *__restrict__ qkv,
                                                             T *__restrict__ out, int N, int heads,
                                                             float scale, int KC) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *Ks = reinterpret_cast<float *>(smem);  // [KC][HD]
  float *Vs = Ks + (size_t)KC * HD;             // [KC][HD]
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int C = heads * HD, ld = 3 * C;
  const T *base = qkv + (size_t)b * N * ld;
  for (int q0 = 0; q0 < N; q0 += blockDim.x) {
    const int qi = q0 + threadIdx.x;
    const bool active = qi < N;
    float q[HD], o[HD];
    float m = -__builtin_inff(), l = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      q[d] = active ? Store<T>::ld(base + (size_t)qi * ld + h * HD + d) * scale : 0.f;
      o[d] = 0.f;
    }
    for (int k0 = 0; k0 < N; k0 += KC) {
      const int kc = min(KC, N - k0);
      __syncthreads();
      for (int i = threadIdx.x; i < kc * HD; i += blockDim.x) {
        const int kk = i / HD, d = i - kk * HD;
        const T *row = base + (size_t)(k0 + kk) * ld + h * HD + d;
        Ks[i] = Store<T>::ld(row + C);
        Vs[i] = Store<T>::ld(row + 2 * C);
      }
      __syncthreads();
      for (int g = 0; g < kc; g += 8) {
        float s[8];
        float gm = -__builtin_inff();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float a = 0.f;
          if (g + u < kc) {
            const float4 *kr = reinterpret_cast<const float4 *>(Ks + (size_t)(g + u) * HD);
#pragma unroll
            for (int d4 = 0; d4 < HD / 4; ++d4) {
              const float4 kv = kr[d4];
              a = fmaf(q[4 * d4 + 0], kv.x, a);
              a = fmaf(q[4 * d4 + 1], kv.y, a);
              a = fmaf(q[4 * d4 + 2], kv.z, a);
              a = fmaf(q[4 * d4 + 3], kv.w, a);
            }
          } else {
            a = -__builtin_inff();
          }
          s[u] = a;
          gm = fmaxf(gm, a);
        }
        const float mn = fmaxf(m, gm);
        const float alpha = expf(m - mn);  // first group: exp(-inf) = 0
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (g + u < kc) {
            const float pu = expf(s[u] - mn);
            l += pu;
            const float4 *vr = reinterpret_cast<const float4 *>(Vs + (size_t)(g + u) * HD);
#pragma unroll
            for (int d4 = 0; d4 < HD / 4; ++d4) {
              const float4 vv = vr[d4];
              o[4 * d4 + 0] = fmaf(pu, vv.x, o[4 * d4 + 0]);
              o[4 * d4 + 1] = fmaf(pu, vv.y, o[4 * d4 + 1]);
              o[4 * d4 + 2] = fmaf(pu, vv.z, o[4 * d4 + 2]);
              o[4 * d4 + 3] = fmaf(pu, vv.w, o[4 * d4 + 3]);
            }
          }
        }
        m = mn;
      }
    }
    if (active) {
      const float inv = 1.0f / l;
      T *orow = out + ((size_t)b * N + qi) * C + h * HD;
#pragma unroll
      for (int d = 0; d < HD; ++d) Store<T>::st(orow + d, o[d] * inv);
    }
  }
}

template <typename T, int HD>
static int launch_attention_valu(const void *qkv, void *out, int B, int N, int heads, hipStream_t s) {
  int KC = (64 * 1024) / (2 * HD * 4);
  KC = (KC / 8) * 8;
  if (KC > N) KC = ((N + 7) / 8) * 8;
  const size_t lds = (size_t)2 * KC * HD * 4;
  int threads = ((N + 63) / 64) * 64;
  if (threads > 256) threads = 256;
  const float scale = 1.0f / sqrtf((float)HD);
  hipLaunchKernelGGL((attention_valu_kernel<T, HD>), dim3(B * heads), dim3(threads), lds, s,
                     (const T *)qkv, (T *)out, N, heads, scale, KC);
  PP_CHECK_LAUNCH("attention_valu_kernel");
  return 0;
}

template <typename T>
int attention_valu(const void *qkv, void *out, int B, int N, int heads, int hd, hipStream_t s) {
  switch (hd) {
    case 32: return launch_attention_valu<T, 32>(qkv, out, B, N, heads, s);
    case 64: return launch_attention_valu<T, 64>(qkv, out, B, N, heads, s);
    case 80: return launch_attention_valu<T, 80>(qkv, out, B, N, heads, s);
    default: return fail("pp_attention: head_dim %d not supported (32, 64, 80)", hd);
  }
}
template int attention_valu<float>(const void *, void *, int, int, int, int, hipStream_t);
template int attention_valu<bf16_t>(const void *, void *, int, int, int, int, hipStream_t);

static int grid_for(long long work_items) {
  long long g = (work_items + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace pp

using namespace pp;

extern "C" int pp_layernorm(const float *x, const float *gamma, const float *beta, float eps, int rows,
                            int C, void *out, int dtype, void *stream) {
  PP_REQUIRE(rows >= 0 && C > 0, "pp_layernorm: bad shape");
  if (rows == 0) return 0;
  PP_REQUIRE(x && gamma && beta && out, "pp_layernorm: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int grid = cdiv(rows, 4);
  if (dtype == PP_BF16) {
    if (!layernorm_rows_launch<bf16_t>(x, gamma, beta, eps, rows, C, (bf16_t *)out, 1.0f, s))
      hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, rows, C,
                         (bf16_t *)out);
  } else if (dtype == PP_F32) {
    if (!layernorm_rows_launch<float>(x, gamma, beta, eps, rows, C, (float *)out, 1.0f, s))
      hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, x, gamma, beta, eps, rows, C,
                         (float *)out);
  } else {
    return fail("pp_layernorm: bad dtype %d", dtype);
  }
  PP_CHECK_LAUNCH("layernorm_kernel");
  return 0;
}

extern "C" int pp_layernorm_fp8(const float *x, const float *gamma, const float *beta, float eps, int rows, int C,
                                unsigned char *out, float inv_scale, void *stream) {
  PP_REQUIRE(rows >= 0 && C > 0 && (C & 3) == 0 && C <= 2048, "pp_layernorm_fp8: C must be a multiple of 4, <= 2048");
  PP_REQUIRE(inv_scale > 0.f, "pp_layernorm_fp8: inv_scale must be positive");
  if (rows == 0) return 0;
  PP_REQUIRE(x && gamma && beta && out, "pp_layernorm_fp8: null pointer");
  if (!layernorm_rows_launch<unsigned char>(x, gamma, beta, eps, rows, C, out, inv_scale, (hipStream_t)stream))
    hipLaunchKernelGGL(layernorm_kernel<unsigned char>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, eps, rows, C, out, inv_scale);
  PP_CHECK_LAUNCH("layernorm_kernel<fp8>");
  return 0;
}

extern "C" int pp_patchify(const float *x, void *out, int B, int H, int W, int patch, int dtype,
                           void *stream) {
  PP_REQUIRE(B >= 0 && H > 0 && W > 0 && patch > 0, "pp_patchify: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(x && out, "pp_patchify: null pointer");
  PP_REQUIRE(W % 4 == 0 && patch % 4 == 0, "pp_patchify: W and patch must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for((long long)B * 3 * H * (W / 4));
  if (dtype == PP_BF16)
    hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, (bf16_t *)out, B, H, W, patch);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, s, x, (float *)out, B, H, W, patch);
  else
    return fail("pp_patchify: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("patchify_kernel");
  return 0;
}

extern "C" int pp_maxpool_relu(const void *x, void *out, int B, int h, int w, int C, int kh, int kw,
                               int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && h > 0 && w > 0 && C > 0 && kh > 0 && kw > 0, "pp_maxpool_relu: bad shape");
  PP_REQUIRE(h / kh > 0 && w / kw > 0, "pp_maxpool_relu: window %dx%d larger than input %dx%d", kh, kw, h, w);
  if (B == 0) return 0;
  PP_REQUIRE(x && out, "pp_maxpool_relu: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int vec = dtype == PP_BF16 ? 8 : 4;
  PP_REQUIRE(C % vec == 0, "pp_maxpool_relu: C=%d must be a multiple of %d", C, vec);
  const int grid = grid_for((long long)B * (h / kh) * (w / kw) * (C / vec));
  if (dtype == PP_BF16)
    hipLaunchKernelGGL(maxpool_relu_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)x,
                       (bf16_t *)out, B, h, w, C, kh, kw);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL(maxpool_relu_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x,
                       (float *)out, B, h, w, C, kh, kw);
  else
    return fail("pp_maxpool_relu: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("maxpool_relu_kernel");
  return 0;
}

extern "C" int pp_aux_tail(const void *x, const void *w, const float *bias, float *out, int B, int C,
                           int K, int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && C > 0 && K > 0, "pp_aux_tail: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(x && w && bias && out, "pp_aux_tail: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int nbr = 4;
  const unsigned relu_mask = 1u << 3;  // probability, visibility, oks: sigmoid; error: ReLU
  const int grid = cdiv((long long)nbr * B * K, 4);
  if (dtype == PP_BF16)
    hipLaunchKernelGGL(aux_tail_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)x,
                       (const bf16_t *)w, bias, out, B, C, K, nbr, relu_mask);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL(aux_tail_kernel<float>, dim3(grid), dim3(256), 0, s, (const float *)x,
                       (const float *)w, bias, out, B, C, K, nbr, relu_mask);
  else
    return fail("pp_aux_tail: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("aux_tail_kernel");
  return 0;
}

extern "C" int pp_maxpool_relu_sum(const float *x, int nsplit, long long split_stride, const float *bias, void *out,
                                   int B, int h, int w, int C, int kh, int kw, int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && h > 0 && w > 0 && C > 0 && kh > 0 && kw > 0 && nsplit >= 1, "pp_maxpool_relu_sum: bad shape");
  PP_REQUIRE(h / kh > 0 && w / kw > 0, "pp_maxpool_relu_sum: window %dx%d larger than input %dx%d", kh, kw, h, w);
  if (B == 0) return 0;
  PP_REQUIRE(x && bias && out, "pp_maxpool_relu_sum: null pointer");
  PP_REQUIRE(C % 4 == 0 && split_stride % 4 == 0, "pp_maxpool_relu_sum: C and the split stride must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  const int grid = grid_for((long long)B * (h / kh) * (w / kw) * (C / 4));
  if (dtype == PP_BF16)
    hipLaunchKernelGGL(maxpool_relu_sum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, x, nsplit, split_stride, bias,
                       (bf16_t *)out, B, h, w, C, kh, kw);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL(maxpool_relu_sum_kernel<float>, dim3(grid), dim3(256), 0, s, x, nsplit, split_stride, bias,
                       (float *)out, B, h, w, C, kh, kw);
  else
    return fail("pp_maxpool_relu_sum: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("maxpool_relu_sum_kernel");
  return 0;
}

template <bool CLAMP>
static int final_heatmap_launch(const void *x, const void *w, const float *bias, float *out, int B, int HW,
                                int Cin, int K, float temperature, int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && HW > 0 && Cin > 0 && K > 0 && temperature != 0.f, "pp_final_heatmap: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(x && w && bias && out, "pp_final_heatmap: null pointer");
  PP_REQUIRE(dtype == PP_BF16 || dtype == PP_F32, "pp_final_heatmap: bad dtype %d", dtype);
  const int es = dtype == PP_BF16 ? 2 : 4, vec = 16 / es;
  PP_REQUIRE(Cin % vec == 0, "pp_final_heatmap: Cin=%d must be a multiple of %d", Cin, vec);
  const long long M = (long long)B * HW;
  PP_REQUIRE(M < (1ll << 31), "pp_final_heatmap: too many rows");
  if (dtype == PP_BF16 && HW % 16 == 0 && Cin % 32 == 0 && Cin <= 32 * FHM_MAXKS && K <= 144 &&
      ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0) {
    const int nt = cdiv(K, 16), tiles = (int)(M / 16);
    const size_t lds_w = (size_t)nt * 16 * (Cin * 2 + 16);
    const int grid = (int)std::min<long long>(cdiv(tiles, 4), 256 * 2);
    hipStream_t sm = (hipStream_t)stream;
#define PP_FHM(NT_)                                                                                       \
  do {                                                                                                    \
    if (lds_w > 64 * 1024)                                                                                \
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&final_heatmap_mfma_kernel<NT_, CLAMP>),    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));          \
    hipLaunchKernelGGL((final_heatmap_mfma_kernel<NT_, CLAMP>), dim3(grid), dim3(256), lds_w, sm, (const bf16_t *)x, \
                       (const bf16_t *)w, bias, out, tiles, HW, Cin, K, temperature);                     \
  } while (0)
    if (nt <= 2) PP_FHM(2);
    else if (nt <= 4) PP_FHM(4);
    else PP_FHM(9);
#undef PP_FHM
    PP_CHECK_LAUNCH("final_heatmap_mfma_kernel");
    return 0;
  }
  const size_t lds = (size_t)FH_ROWS * (Cin / vec + 1) * 16 + (size_t)K * Cin * es;
  PP_REQUIRE(lds <= 160 * 1024, "pp_final_heatmap: K=%d x Cin=%d does not fit LDS", K, Cin);
  hipStream_t s = (hipStream_t)stream;
  const int grid = cdiv(M, FH_ROWS);
  if (dtype == PP_BF16) {
    if (lds > 64 * 1024)
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(final_heatmap_kernel<bf16_t, CLAMP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((final_heatmap_kernel<bf16_t, CLAMP>), dim3(grid), dim3(256), lds, s, (const bf16_t *)x,
                       (const bf16_t *)w, bias, out, (int)M, HW, Cin, K, temperature);
  } else {
    if (lds > 64 * 1024)
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(final_heatmap_kernel<float, CLAMP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((final_heatmap_kernel<float, CLAMP>), dim3(grid), dim3(256), lds, s, (const float *)x,
                       (const float *)w, bias, out, (int)M, HW, Cin, K, temperature);
  }
  PP_CHECK_LAUNCH("final_heatmap_kernel");
  return 0;
}

extern "C" int pp_final_heatmap(const void *x, const void *w, const float *bias, float *out, int B, int HW,
                                int Cin, int K, float temperature, int dtype, void *stream) {
  return final_heatmap_launch<true>(x, w, bias, out, B, HW, Cin, K, temperature, dtype, stream);
}

extern "C" int pp_final_logits(const void *x, const void *w, const float *bias, float *out, int B, int HW,
                               int Cin, int K, float temperature, int dtype, void *stream) {
  return final_heatmap_launch<false>(x, w, bias, out, B, HW, Cin, K, temperature, dtype, stream);
}

extern "C" int pp_tokens_to_nchw(const void *x, float *out, int B, int N, int C, int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && N > 0 && C > 0, "pp_tokens_to_nchw: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(x && out, "pp_tokens_to_nchw: null pointer");
  PP_REQUIRE(B <= 65535, "pp_tokens_to_nchw: batch too large");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(C, 32), cdiv(N, 32), B);
  if (dtype == PP_BF16)
    hipLaunchKernelGGL((transpose_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t *)x, out, N, C);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float *)x, out, N, C);
  else
    return fail("pp_tokens_to_nchw: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("transpose_kernel");
  return 0;
}

extern "C" int pp_nchw_to_tokens(const float *x, void *out, int B, int C, int HW, int dtype, void *stream) {
  PP_REQUIRE(B >= 0 && HW > 0 && C > 0, "pp_nchw_to_tokens: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(x && out, "pp_nchw_to_tokens: null pointer");
  PP_REQUIRE(B <= 65535, "pp_nchw_to_tokens: batch too large");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
  if (dtype == PP_BF16)
    hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, s, x, (bf16_t *)out, C, HW);
  else if (dtype == PP_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, x, (float *)out, C, HW);
  else
    return fail("pp_nchw_to_tokens: bad dtype %d", dtype);
  PP_CHECK_LAUNCH("transpose_kernel");
  return 0;
}
