// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,Kd] * W[N,Kd]^T), fp32 accumulate.
//
// One kernel family serves every dense contraction of the path (patch-embed,
// qkv, proj, fc1, fc2, 3x3 aux convolutions, the four stride-2 deconvolution
// parities, the final 1x1 heatmap layer): both operands are K-contiguous rows,
// A rows optionally addressed through a gather table (implicit GEMM, no im2col
// buffer), C rows optionally scattered through a row map.
//
// Structure (CDNA4): BM x BN output tile per 256-thread workgroup (4 waves as
// 2x2, each (BM/2)x(BN/2) = TMxTN MFMA 16x16 tiles), two instantiations:
// 192x96 (one crop's 192 tokens per M-tile: at B = 64 the four ViT GEMMs are
// exactly 1/3/4/1 rounds of 512 resident workgroups, no tail) and 128x128;
// K-tiles of 128 B per row
// (64 bf16 / 32 fp32) staged HBM->LDS by global_load_lds_dwordx4 (no VGPR
// round trip) into a double buffer, XOR-swizzled on the SOURCE address so the
// LDS image stays lane-linear for the DMA while ds_read_b128 fragment reads
// are bank-conflict free.  bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity mode):
// v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 FMA chain.
#include "pp_common.h"

namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROW_BYTES = 128;                 // bytes of K per staged row
constexpr int GEMM_THREADS = 256;
constexpr int gemm_lds_bytes(int BM, int BN) { return 2 * (BM + BN) * ROW_BYTES; }  // double buffer

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

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
  int epilogue;
  int hm_K, hm_HW;
  float hm_temperature;
  int tiles_m, tiles_n;
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
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp): bf16 epilogues only.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erfz = 1.0f - poly * t * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erfz, x));
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = ROW_BYTES / ES;  // elements of K per tile
  constexpr int PA = BM / 32, PB = BN / 32;      // 1-KiB DMA pieces (8 rows) per wave per K-tile
  constexpr int TM = BM / 32, TN = BN / 32;      // 16x16 MFMA tiles per wave: (BM/2)/16 x (BN/2)/16
  constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;

  // ---- tile assignment: XCD-aware remap (blocks b, b+8 share an XCD/L2) so
  // the tiles sharing one A row-panel run on one XCD back to back.
  const int ntiles = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntiles >> 3, r = ntiles & 7, x = bid & 7, j = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const char *Ab = p.A + (size_t)z * p.strideA * ES;
  const char *Wb = p.W + (size_t)z * p.strideW * ES;
  const int32_t *rowoff = p.rowoff ? p.rowoff + (size_t)z * p.strideRowoff : nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- staging geometry: each wave issues PA A-pieces + PB W-pieces (8 rows x 128 B) per K-tile.
  // lane -> (row in piece, physical 16-B chunk); logical chunk = physical ^ (row & 7).
  const int prow = lane >> 3, pchunk = lane & 7;
  int a_row[PA], w_row[PB];
  const char *a_src[PA];
  const char *w_src[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (wave * PA + j) * 8 + prow;  // row inside the BM-row tile
    a_row[j] = min(m0 + r, p.M - 1);           // tail rows re-read the last row (never stored)
    a_src[j] = Ab + (size_t)a_row[j] * p.lda * ES + (pchunk ^ prow) * 16;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int r = (wave * PB + j) * 8 + prow;
    w_row[j] = n0 + r;
    w_src[j] = (w_row[j] < p.N) ? Wb + (size_t)w_row[j] * p.ldw * ES + (pchunk ^ prow) * 16
                                : (const char *)g_zero_page + (pchunk ^ prow) * 16;
  }
  const int lchunk_off = (pchunk ^ prow) * 16;  // (row & 7) == prow for every piece
  const int nkt = p.Kd / BK;

  const unsigned lds0 = lds_offset_of(smem);
  // gather mode: the row offsets of a K-segment (one convolution tap) stay in registers for the
  // whole segment and the next segment's are fetched one K-tile ahead, so the dependent
  // table-load -> DMA-address chain never sits on the critical path.
  int cur_off[PA], nxt_off[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j) cur_off[j] = nxt_off[j] = 0;
  if (rowoff) {
#pragma unroll
    for (int j = 0; j < PA; ++j) cur_off[j] = rowoff[a_row[j]];
  }
  auto stage = [&](int kt, int buf) {
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * PA * 1024);
    const unsigned ldsB = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + A_BYTES + wave * PB * 1024);
    const size_t koff = (size_t)kt * ROW_BYTES;
    if (rowoff) {
      const int k0 = kt * BK;
      const int seg = k0 / p.seg_len;
      const int kin = k0 - seg * p.seg_len;
      if (kin == 0 && kt > 0) {
#pragma unroll
        for (int j = 0; j < PA; ++j) cur_off[j] = nxt_off[j];
      }
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const char *src = cur_off[j] >= 0 ? Ab + ((size_t)cur_off[j] + kin) * ES + lchunk_off
                                          : (const char *)g_zero_page + lchunk_off;
        glds16(src, ldsA + j * 1024);
      }
      if (kin + BK == p.seg_len && k0 + BK < p.Kd) {
#pragma unroll
        for (int j = 0; j < PA; ++j) nxt_off[j] = rowoff[(size_t)(seg + 1) * p.M + a_row[j]];
      }
    } else {
#pragma unroll
      for (int j = 0; j < PA; ++j) glds16(a_src[j] + koff, ldsA + j * 1024);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j)
      glds16(w_row[j] < p.N ? w_src[j] + koff : w_src[j], ldsB + j * 1024);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses: row = tile row (lane & 15), logical chunk = 4*s + (lane >> 4)
  const int frow = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const char *ldsA = smem + buf * STAGE_BYTES;
    const char *ldsB = ldsA + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ra = wm * (BM / 2) + i * 16 + frow;
        af[i] = *reinterpret_cast<const uint4 *>(ldsA + ra * ROW_BYTES + (((4 * s + fq) ^ (ra & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int rb = wn * (BN / 2) + j * 16 + frow;
        bf[j] = *reinterpret_cast<const uint4 *>(ldsB + rb * ROW_BYTES + (((4 * s + fq) ^ (rb & 7)) << 4));
      }
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                *reinterpret_cast<bf16x8 *>(&bf[j]), *reinterpret_cast<bf16x8 *>(&af[i]), acc[i][j], 0, 0, 0);
      } else {
        // fp32: the chunk holds 4 consecutive k; MFMA step e takes element e of every lane's
        // chunk (k slots 16s + 4*fq + e, the same permutation on both operands).
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                  __uint_as_float(reinterpret_cast<const unsigned *>(&bf[j])[e]),
                  __uint_as_float(reinterpret_cast<const unsigned *>(&af[i])[e]), acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- main loop: DMA of tile t+1 in flight while tile t feeds the MFMAs
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt - 1; ++kt) {
    stage(kt + 1, cur ^ 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  compute(cur);

  // ---- epilogue.  The W fragment is the MFMA "A" operand and the activation fragment the "B"
  // operand, so a 16x16 accumulator tile holds C^T: lane (frow, fq) owns output row m = .. + frow and
  // the 4 CONSECUTIVE columns n = .. + 4*fq + e -> 8-byte (bf16) / 16-byte (fp32) vector stores.
  const int epi = p.epilogue;
  const float *bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const int32_t *rowmap = p.out_rowmap ? p.out_rowmap + (size_t)z * p.strideRowmap : nullptr;
  char *Cb = p.C + (size_t)z * p.strideC * ((epi & (PP_EPI_OUT_F32 | PP_EPI_HEATMAP)) ? 4 : ES);
  const float *Rb = p.residual ? p.residual + (size_t)z * p.strideC : nullptr;
  const bool vec_ok = (p.N & 3) == 0 && (p.ldc & 3) == 0 && !(epi & PP_EPI_HEATMAP);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 16 + frow;
    if (m >= p.M) continue;
    const int r = rowmap ? rowmap[m] : m;
    const float *rb = (epi & PP_EPI_ROWBIAS) ? p.rowbias + (size_t)(m % p.rowbias_period) * p.ldc : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (vec_ok) {
        if (epi & PP_EPI_BIAS) {
          const float4 b4 = *reinterpret_cast<const float4 *>(bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (rb) {
          const float4 b4 = *reinterpret_cast<const float4 *>(rb + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (epi & PP_EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (sizeof(T) == 2) ? gelu_fast(v[e]) : gelu_erf(v[e]);
        }
        if (epi & PP_EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const size_t idx = (size_t)r * p.ldc + n;
        if (epi & PP_EPI_RESIDUAL) {
          const float4 r4 = *reinterpret_cast<const float4 *>(Rb + idx);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
        if ((epi & PP_EPI_OUT_F32) || sizeof(T) == 4) {
          *reinterpret_cast<float4 *>(reinterpret_cast<float *>(Cb) + idx) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 pk;
          pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
          pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
          *reinterpret_cast<uint2 *>(reinterpret_cast<bf16_t *>(Cb) + idx) = pk;
        }
        continue;
      }
      // scalar path: ragged N (e.g. the K=17 heatmap layer) or unaligned ldc
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ne = n + e;
        if (ne >= p.N) continue;
        float x = v[e];
        if (epi & PP_EPI_BIAS) x += bias[ne];
        if (rb) x += rb[ne];
        if (epi & PP_EPI_GELU) x = (sizeof(T) == 2) ? gelu_fast(x) : gelu_erf(x);
        if (epi & PP_EPI_RELU) x = fmaxf(x, 0.f);
        if (epi & PP_EPI_HEATMAP) {
          x = fminf(fmaxf(x / p.hm_temperature, 0.f), 1.f);
          const int b = r / p.hm_HW, hw = r - b * p.hm_HW;
          reinterpret_cast<float *>(Cb)[((size_t)b * p.hm_K + ne) * p.hm_HW + hw] = x;
          continue;
        }
        const size_t idx = (size_t)r * p.ldc + ne;
        if (epi & PP_EPI_RESIDUAL) x += Rb[idx];
        if (epi & PP_EPI_OUT_F32)
          reinterpret_cast<float *>(Cb)[idx] = x;
        else
          Store<T>::st(reinterpret_cast<T *>(Cb) + idx, x);
      }
    }
  }
}

}  // namespace pp

extern "C" int pp_gemm(const pp_gemm_args *a, void *stream) {
  using namespace pp;
  PP_REQUIRE(a, "pp_gemm: null args");
  PP_REQUIRE(a->dtype == PP_F32 || a->dtype == PP_BF16, "pp_gemm: bad dtype %d", a->dtype);
  const int es = a->dtype == PP_BF16 ? 2 : 4;
  const int bk = ROW_BYTES / es;
  PP_REQUIRE(a->M >= 0 && a->N > 0 && a->Kd > 0, "pp_gemm: bad shape M=%d N=%d K=%d", a->M, a->N, a->Kd);
  if (a->M == 0 || a->batch == 0) return 0;
  PP_REQUIRE(a->A && a->W && a->C, "pp_gemm: null operand");
  PP_REQUIRE(a->Kd % bk == 0, "pp_gemm: K=%d must be a multiple of %d for this dtype", a->Kd, bk);
  PP_REQUIRE(a->lda % (16 / es) == 0 && a->ldw % (16 / es) == 0,
             "pp_gemm: lda/ldw must keep rows 16-byte aligned");
  PP_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "pp_gemm: operands must be 16-byte aligned");
  if (a->rowoff)
    PP_REQUIRE(a->seg_len > 0 && a->seg_len % bk == 0 && a->Kd % a->seg_len == 0,
               "pp_gemm: gather segment length %d must divide K=%d and be a multiple of %d", a->seg_len,
               a->Kd, bk);
  if (a->epilogue & PP_EPI_BIAS) PP_REQUIRE(a->bias, "pp_gemm: PP_EPI_BIAS without bias");
  if (a->epilogue & PP_EPI_RESIDUAL) PP_REQUIRE(a->residual, "pp_gemm: PP_EPI_RESIDUAL without residual");
  if (a->epilogue & PP_EPI_ROWBIAS)
    PP_REQUIRE(a->rowbias && a->rowbias_period > 0, "pp_gemm: PP_EPI_ROWBIAS without rowbias/period");
  if (a->epilogue & PP_EPI_HEATMAP)
    PP_REQUIRE(a->hm_K >= a->N && a->hm_HW > 0 && a->hm_temperature != 0.f, "pp_gemm: bad heatmap epilogue");
  GemmParams p;
  p.A = (const char *)a->A;
  p.W = (const char *)a->W;
  p.C = (char *)a->C;
  p.bias = a->bias;
  p.residual = a->residual;
  p.rowbias = a->rowbias;
  p.rowoff = a->rowoff;
  p.out_rowmap = a->out_rowmap;
  p.M = a->M; p.N = a->N; p.Kd = a->Kd;
  p.lda = a->lda; p.ldw = a->ldw; p.ldc = a->ldc;
  p.seg_len = a->seg_len > 0 ? a->seg_len : a->Kd;
  p.rowbias_period = a->rowbias_period;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideC = a->strideC;
  p.strideBias = a->strideBias; p.strideRowoff = a->strideRowoff; p.strideRowmap = a->strideRowmap;
  p.epilogue = a->epilogue;
  p.hm_K = a->hm_K; p.hm_HW = a->hm_HW; p.hm_temperature = a->hm_temperature;
  const int batch = a->batch > 0 ? a->batch : 1;
  PP_REQUIRE(batch <= 65535, "pp_gemm: batch too large");
  // tile shape: fewest rounds of the 512 co-resident workgroups (256 CUs x 2), weighted by tile area
  auto cost = [&](int bm, int bn) {
    const long long tiles = (long long)cdiv(a->M, bm) * cdiv(a->N, bn) * batch;
    return ((tiles + 511) / 512) * (long long)bm * bn;
  };
  PP_REQUIRE(a->tile >= 0 && a->tile <= 2, "pp_gemm: bad tile selector %d", a->tile);
  const bool wide = a->tile == 2 || (a->tile == 0 && cost(192, 96) < cost(128, 128));
  const int bm = wide ? 192 : 128, bn = wide ? 96 : 128;
  p.tiles_m = cdiv(a->M, bm);
  p.tiles_n = cdiv(a->N, bn);
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  hipStream_t s = (hipStream_t)stream;
  const int lds = gemm_lds_bytes(bm, bn);
#define PP_LAUNCH_GEMM(T, BM_, BN_)                                                                   \
  do {                                                                                                \
    static thread_local bool attr = false;                                                            \
    if (!attr) {                                                                                      \
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kernel<T, BM_, BN_>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));             \
      attr = true;                                                                                    \
    }                                                                                                 \
    hipLaunchKernelGGL((gemm_kernel<T, BM_, BN_>), grid, dim3(GEMM_THREADS), lds, s, p);              \
  } while (0)
  if (a->dtype == PP_BF16) {
    if (wide) PP_LAUNCH_GEMM(bf16_t, 192, 96); else PP_LAUNCH_GEMM(bf16_t, 128, 128);
  } else {
    if (wide) PP_LAUNCH_GEMM(float, 192, 96); else PP_LAUNCH_GEMM(float, 128, 128);
  }
#undef PP_LAUNCH_GEMM
  PP_CHECK_LAUNCH("gemm_kernel");
  return 0;
}
