// Multi-head self-attention (timm Attention.forward semantics:
// softmax(q k^T * hd^-1/2) v per (crop, head)).  N = 192 tokens at 256x192, so
// one (crop, head) problem lives entirely on a CU: no online-softmax tiling.
//
// bf16 MFMA kernel (head_dim 64, N <= 192), one workgroup = 4 waves per
// (crop, head), 48 query rows per wave:
//   S^T = K Q^T      keys on the MFMA row axis, queries on the lane (column)
//                    axis, so a query's scores sit in one lane quartet and the
//                    row max / sum is 47 in-lane ops + 2 shuffles;
//   P^T stays in the accumulator registers and is fed straight back as the B
//                    operand of the second product (no LDS round trip); the keys are
//                    processed as two blocks of 96 with an online-softmax rescale so
//                    that three workgroups fit a CU;
//   O^T = V^T P^T    V is transposed once while being staged into LDS.
// The k-slot order inside each 32-key MFMA step is permuted identically on both
// operands (slot (g,j) <-> key 4g + j for j < 4, 16 + 4g + (j-4) otherwise),
// which is what makes the accumulator directly reusable.
// Everything else (fp32 parity mode, other head dims, N > 192) runs the exact
// fp32 VALU kernel in pp_ops.hip.
#include "pp_common.h"

namespace pp {

template <typename T>
int attention_valu(const void *qkv, void *out, int B, int N, int heads, int hd, hipStream_t s);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_HD = 64;
constexpr int AT_NMAX = 192;              // keys / queries held on chip
constexpr int AT_NT = AT_NMAX / 16;       // 12 key tiles
constexpr int AT_QT = 3;                  // query tiles (16 rows) per wave
constexpr int AT_KROW = AT_HD * 2;        // 128 B per K row in LDS
constexpr int AT_VROW = (AT_NMAX + 8) * 2;  // V^T row: 192 keys + 8 pad (bank spread), bytes
constexpr int AT_LDS = AT_NMAX * AT_KROW + AT_HD * AT_VROW;  // 24576 + 25600

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

__global__ __launch_bounds__(256, 3) void attention_mfma_kernel(const bf16_t *__restrict__ qkv,
                                                                bf16_t *__restrict__ out, int N,
                                                                int heads, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Ks = smem;                          // [192][64] bf16, 16-B chunks XOR-swizzled by (row & 7)
  char *Vt = smem + AT_NMAX * AT_KROW;      // [64][200] bf16 (V transposed)
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int C = heads * AT_HD, ld = 3 * C;
  const bf16_t *base = qkv + (size_t)b * N * ld + h * AT_HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;

  // ---- stage K (row-major, swizzled) and V (transposed) into LDS; rows >= N are zero
  for (int i = tid; i < AT_NMAX * 8; i += 256) {  // K: 192 rows x 8 chunks of 16 B
    const int r = i >> 3, c = i & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < N) v = *reinterpret_cast<const uint4 *>(base + (size_t)r * ld + C + c * 8);
    *reinterpret_cast<uint4 *>(Ks + r * AT_KROW + ((c ^ (r & 7)) << 4)) = v;
  }
  // V: (key pair, 8-dim chunk) items; consecutive lanes take consecutive key pairs so the transposed
  // 32-bit stores of one wave-instruction spread over all 32 LDS banks (2-way, free).
  for (int i = tid; i < (AT_NMAX / 2) * 8; i += 256) {
    const int c = i / (AT_NMAX / 2), kp = i - c * (AT_NMAX / 2);
    const int k0 = 2 * kp;
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
    if (k0 < N) v0 = *reinterpret_cast<const uint4 *>(base + (size_t)k0 * ld + 2 * C + c * 8);
    if (k0 + 1 < N) v1 = *reinterpret_cast<const uint4 *>(base + (size_t)(k0 + 1) * ld + 2 * C + c * 8);
    const unsigned short *a = reinterpret_cast<const unsigned short *>(&v0);
    const unsigned short *bb = reinterpret_cast<const unsigned short *>(&v1);
#pragma unroll
    for (int e = 0; e < 8; ++e)  // Vt[d][k0], Vt[d][k0+1] as one 32-bit store
      *reinterpret_cast<unsigned *>(Vt + (c * 8 + e) * AT_VROW + k0 * 2) = (unsigned)a[e] | ((unsigned)bb[e] << 16);
  }

  // ---- Q fragments straight from global: B operand, lane (col q = lrow, g) holds Q[q][32s + 8g .. +7]
  const int q0 = wave * (AT_QT * 16);
  uint4 qf[AT_QT][2];
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    const int q = q0 + t * 16 + lrow;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qf[t][s] = make_uint4(0, 0, 0, 0);
      if (q < N) qf[t][s] = *reinterpret_cast<const uint4 *>(base + (size_t)q * ld + s * 32 + g * 8);
    }
  }
  __syncthreads();

  // ---- two key blocks of 96 keys, flash style: only 6 x QT score tiles (72 registers) are live at a
  // time, which lets three workgroups share a CU (768 (crop, head) problems = one full round of
  // 256 CUs x 3 instead of 1.5 rounds of 2).  Running max m, per-lane partial sum l, and the output
  // accumulator are rescaled by alpha = exp2((m_old - m_new) * c) between the blocks.
  constexpr int KB_TILES = AT_NT / 2;  // 6 key tiles per block
  float m_run[AT_QT], l_run[AT_QT];
  f32x4 oacc[4][AT_QT];
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    m_run[t] = -__builtin_inff();
    l_run[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    // S^T block = K_block Q^T : sacc[kt][t] rows = keys 96*kb + 16*kt + 4g + r, col = query
    f32x4 sacc[KB_TILES][AT_QT];
#pragma unroll
    for (int kt = 0; kt < KB_TILES; ++kt) {
#pragma unroll
      for (int t = 0; t < AT_QT; ++t) sacc[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int r = (kb * KB_TILES + kt) * 16 + lrow;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint4 kf = *reinterpret_cast<const uint4 *>(Ks + r * AT_KROW + (((4 * s + g) ^ (r & 7)) << 4));
#pragma unroll
        for (int t = 0; t < AT_QT; ++t)
          sacc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&kf),
                                                                *reinterpret_cast<const bf16x8 *>(&qf[t][s]),
                                                                sacc[kt][t], 0, 0, 0);
      }
    }
    // online softmax over this block's keys, per query column
#pragma unroll
    for (int t = 0; t < AT_QT; ++t) {
      float mb = -__builtin_inff();
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = (kb * KB_TILES + kt) * 16 + g * 4 + r;
          if (key >= N) sacc[kt][t][r] = -__builtin_inff();
          mb = fmaxf(mb, sacc[kt][t][r]);
        }
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run[t], mb);          // finite: block 0 always holds key 0
      const float alpha = exp2f((m_run[t] - m_new) * scale_log2e);  // block 0: exp2(-inf) = 0
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = exp2f((sacc[kt][t][r] - m_new) * scale_log2e);
          sacc[kt][t][r] = pv;
          l += pv;
        }
      l_run[t] = l_run[t] * alpha + l;
      m_run[t] = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        oacc[dt][t][0] *= alpha; oacc[dt][t][1] *= alpha; oacc[dt][t][2] *= alpha; oacc[dt][t][3] *= alpha;
      }
    }
    // O^T += V_block^T P_block^T : rows = dims 16*dt + 4g + r, col = query
#pragma unroll
    for (int u = 0; u < KB_TILES / 2; ++u) {  // 32 keys per step
      uint4 pf[AT_QT];
#pragma unroll
      for (int t = 0; t < AT_QT; ++t) {
        pf[t].x = pack_bf16x2(sacc[2 * u][t][0], sacc[2 * u][t][1]);
        pf[t].y = pack_bf16x2(sacc[2 * u][t][2], sacc[2 * u][t][3]);
        pf[t].z = pack_bf16x2(sacc[2 * u + 1][t][0], sacc[2 * u + 1][t][1]);
        pf[t].w = pack_bf16x2(sacc[2 * u + 1][t][2], sacc[2 * u + 1][t][3]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const char *vrow = Vt + (dt * 16 + lrow) * AT_VROW + (kb * (KB_TILES * 16) + 32 * u + 4 * g) * 2;
        uint4 vf;
        const uint2 lo = *reinterpret_cast<const uint2 *>(vrow);
        const uint2 hi = *reinterpret_cast<const uint2 *>(vrow + 32);
        vf.x = lo.x; vf.y = lo.y; vf.z = hi.x; vf.w = hi.y;
#pragma unroll
        for (int t = 0; t < AT_QT; ++t)
          oacc[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&vf),
                                                                *reinterpret_cast<const bf16x8 *>(&pf[t]),
                                                                oacc[dt][t], 0, 0, 0);
      }
    }
  }
  float inv_l[AT_QT];
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    inv_l[t] = 1.0f / l;
  }

  // ---- store: lane holds 4 consecutive dims of one query -> 8-byte stores
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    const int q = q0 + t * 16 + lrow;
    if (q >= N) continue;
    bf16_t *orow = out + ((size_t)b * N + q) * C + h * AT_HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 pk;
      pk.x = pack_bf16x2(oacc[dt][t][0] * inv_l[t], oacc[dt][t][1] * inv_l[t]);
      pk.y = pack_bf16x2(oacc[dt][t][2] * inv_l[t], oacc[dt][t][3] * inv_l[t]);
      *reinterpret_cast<uint2 *>(orow + dt * 16 + g * 4) = pk;
    }
  }
}

}  // namespace pp

extern "C" int pp_attention(const void *qkv, void *out, int B, int N, int heads, int hd, int dtype,
                            void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && N > 0 && heads > 0 && hd > 0, "pp_attention: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(qkv && out, "pp_attention: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PP_BF16) {
    if (hd == AT_HD && N <= AT_NMAX && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 7) == 0) {
      const float scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
      hipLaunchKernelGGL(attention_mfma_kernel, dim3(B * heads), dim3(256), AT_LDS, s, (const bf16_t *)qkv,
                         (bf16_t *)out, N, heads, scale_log2e);
      PP_CHECK_LAUNCH("attention_mfma_kernel");
      return 0;
    }
    return attention_valu<bf16_t>(qkv, out, B, N, heads, hd, s);
  }
  if (dtype == PP_F32) return attention_valu<float>(qkv, out, B, N, heads, hd, s);
  return fail("pp_attention: bad dtype %d", dtype);
}
