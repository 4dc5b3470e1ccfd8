// Multi-head self-attention (timm Attention.forward semantics:
// softmax(q k^T * hd^-1/2) v per (crop, head)).  N = 192 tokens at 256x192, so
// one (crop, head) problem (K and V: 48 KB) lives entirely on a CU.
//
// One-shot bf16 MFMA kernel (head_dim 64 / 32, N <= 192), one workgroup = 4 waves per
// (crop, head), 48 query rows per wave:
//   S^T = K Q^T      keys on the MFMA row axis, queries on the lane (column)
//                    axis, so a query's scores sit in one lane quartet and the
//                    row max / sum is 47 in-lane ops + 2 shuffles;
//   P^T stays in the accumulator registers and is fed straight back as the B
//                    operand of the second product (no LDS round trip); the keys are
//                    processed as two blocks of 96 with an online-softmax rescale so
//                    that three workgroups fit a CU;
//   O^T = V^T P^T    V stays row-major in LDS and is read transposed by
//                    ds_read_b64_tr_b16 (one 4-key x 16-dim block per 16-lane group).
// The k-slot order inside each 32-key MFMA step is permuted identically on both
// operands (slot (g,j) <-> key 4g + j for j < 4, 16 + 4g + (j-4) otherwise),
// which is what makes the accumulator directly reusable.
// attention_stream_kernel is the same computation for any N and head_dim 32 / 64 / 80 (ViT-H: N = 432
// at 384x288, hd = 80): one workgroup per (crop, head, block of 128 queries), the keys stream
// through LDS in blocks of 96 with the online-softmax rescale between blocks.
// Everything else (fp32 parity mode, other head dims) runs the exact fp32 VALU kernel in pp_ops.hip.
#include "pp_common.h"

namespace pp {

template <typename T>
int attention_valu(const void *qkv, void *out, int B, int N, int heads, int hd, hipStream_t s);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_NMAX = 192;              // keys / queries held on chip
constexpr int AT_NT = AT_NMAX / 16;       // 12 key tiles
constexpr int AT_QT = 3;                  // query tiles (16 rows) per wave
// head_dim 64 (ViT-B/L) or 32 (the reference's own ViT: embed 384 / 12 heads, backbone.py:30)
template <int HD> struct AttGeom {
  static constexpr int KROW = HD * 2;                 // bytes per K / V row in LDS (128 or 64)
  static constexpr int LDS = 2 * AT_NMAX * KROW;      // K + V, row-major
  static constexpr int CHUNKS = HD / 8;               // 16-B chunks per row
  static constexpr int KS = (HD + 31) / 32;           // 32-deep MFMA steps of the first product (hd 80: the last is half empty)
  static constexpr int KLIVE = (HD % 32) ? (HD % 32) / 8 : 4;  // lane quartets (g) that carry data in the last step
  static constexpr int DT = HD / 16;                  // 16-dim output tiles
  static constexpr int RPP = 1024 / KROW;             // rows per 1-KiB DMA piece (one-shot kernel: HD 32 / 64 only)
  // 16-B chunk swizzle that makes a 16-row ds_read_b128 column hit 16 distinct bank slots.  160-B rows
  // (hd 80) need none: row r starts at bank slot 10 r mod 16, and the b128 lane groups (rows {0-3,12-15}
  // at chunk c with rows {4-11} at chunk c+1) as well as the 8 rows x 32 B of a transposing read
  // (40 r mod 64 dwords) already land on distinct banks.
  static __device__ __forceinline__ int swz(int row, int chunk) {
    return HD == 64 ? (chunk ^ (row & 7)) : (HD == 32 ? (chunk ^ ((row >> 2) & 3)) : chunk);
  }
};
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float att_f32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DPP_ATT_STAMPS): phase cycle counts of wave 0 go behind the output tensor.
#ifdef PP_ATT_STAMPS
__device__ __forceinline__ unsigned long long att_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define ATT_STAMP(v) const unsigned long long v = att_stamp()
#else
#define ATT_STAMP(v)
#endif

__device__ __attribute__((aligned(256))) unsigned char g_att_zero[1024];

__device__ __forceinline__ void att_glds16(const void *gsrc, unsigned lds_off_uniform) {
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 1
  return;   // ablation (tools/att_ablate.sh): no K / V traffic
#endif
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_off_uniform)
      : "memory");
}

// OCP e4m3, saturating at +-448 (fp8 mode: the attention output feeds the fp8 proj GEMM)
__device__ __forceinline__ unsigned att_pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
  b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
  c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
  d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

template <int PENDING>
__device__ __forceinline__ void att_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PENDING) : "memory");
}

// The staging is issued as Q, then groups of 12 one-KiB pieces (3 per wave) in the order the products consume them
// (hd 64: K block 0, V block 0, K block 1, V block 1; hd 32: K+V block 0, K+V block 1), and every product waits only
// for its own group (counted vmcnt + barrier).  Only Q and the first group are requested up front; the other groups
// are requested from inside the first product, because a wave sits in the issue of a load for as long as the memory
// pipeline is backed up.  (The round-1 form -- everything requested up front, one vmcnt(0) + barrier -- measured
// 23.7 us against 23.2 us for ViT-B bs 64 and 19.4 against 16.7 us for hd 32 in one process.)
// FULL: N == 192 exactly (every 256x192 model): no key masks and no row guards.
template <int AT_HD, bool FULL>
__global__ __launch_bounds__(256, 3) void attention_mfma_kernel(const bf16_t *__restrict__ qkv,
                                                                bf16_t *__restrict__ out, int N,
                                                                int heads, float scale_log2e, float fp8_inv_scale) {
  using G = AttGeom<AT_HD>;
  constexpr int AT_KROW = G::KROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Ks = smem;                          // [192][HD] bf16, 16-B chunks XOR-swizzled
  char *Vs = smem + AT_NMAX * AT_KROW;      // [192][HD] bf16, same layout; read transposed (ds_read_b64_tr_b16)
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int C = heads * AT_HD, ld = 3 * C;
  const bf16_t *base = qkv + (size_t)b * N * ld + h * AT_HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;

  ATT_STAMP(t0);
  const int q0 = wave * (AT_QT * 16);
  uint4 qf[AT_QT][G::KS];
  constexpr int NG = (AT_HD == 64) ? 4 : 2;            // DMA groups, 3 pieces per wave each
  constexpr int NLATE = 3 * (NG - 1);                  // pieces per wave issued inside the first product
  // K and V go into LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip), both row-major; the 16-B chunk XOR
  // swizzle is applied on the per-lane SOURCE address so the LDS image stays lane-linear.  V is NOT transposed here:
  // the second product reads it through ds_read_b64_tr_b16.  Piece j of group gi for this wave:
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  auto issue_piece = [&](int gi, int j) __attribute__((always_inline)) {
    constexpr int PPB = 96 / G::RPP;                    // pieces per key block per matrix: 12 (hd 64) or 6 (hd 32)
    const int prow = lane / G::CHUNKS, pchunk = lane % G::CHUNKS;
    const int p = wave * 3 + j;                         // piece of the group, 0..11
    const int mat = (AT_HD == 64) ? (gi & 1) : p / PPB;                        // 0 = K, 1 = V
    const int piece = (AT_HD == 64) ? (gi >> 1) * PPB + p : gi * PPB + p % PPB;     // 1-KiB piece of that matrix
    const int r = piece * G::RPP + prow;
    const int lchunk = G::swz(r, pchunk);
    const void *src = (FULL || r < N) ? (const void *)(base + (size_t)r * ld + (1 + mat) * C + lchunk * 8)
                                      : (const void *)(g_att_zero + lane * 16);
    att_glds16(src, __builtin_amdgcn_readfirstlane(lds0 + mat * (AT_NMAX * AT_KROW) + piece * 1024));
  };
  // ---- Q fragments first (B operand, lane (col q = lrow, g) holds Q[q][32s + 8g .. +7]).  Ordinary loads, and the
  // OLDEST memory operations of the wave: the waits the compiler places for them count only these six, which with
  // younger DMA in flight is conservative, never short.  (Loads hidden from the compiler in inline asm are not safe
  // here: it may copy their destination registers before the data has landed.)
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    const int q = q0 + t * 16 + lrow;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      qf[t][s] = make_uint4(0, 0, 0, 0);
      if (FULL || q < N) qf[t][s] = *reinterpret_cast<const uint4 *>(base + (size_t)q * ld + s * 32 + g * 8);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) issue_piece(0, j);        // group 0 now; the others are issued inside the first product
  ATT_STAMP(t1);
  att_wait_barrier<0>();                               // Q and the first group have landed, for every wave
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // the same vmcnt(0), visible to the compiler: no later Q waits
  ATT_STAMP(t2);

  // ---- two key blocks of 96 keys, flash style: only 6 x QT score tiles (72 registers) are live at a
  // time, which lets three workgroups share a CU (768 (crop, head) problems = one full round of
  // 256 CUs x 3 instead of 1.5 rounds of 2).  Running max m, per-lane partial sum l, and the output
  // accumulator are rescaled by alpha = exp2((m_old - m_new) * c) between the blocks.
  constexpr int KB_TILES = AT_NT / 2;  // 6 key tiles per block
  float m_run[AT_QT], l_run[AT_QT];
  f32x4 oacc[G::DT][AT_QT];
#pragma unroll
  for (int t = 0; t < AT_QT; ++t) {
    m_run[t] = -__builtin_inff();
    l_run[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) oacc[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 3
  if (N < 0)   // ablation: loads and stores only
#endif
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    if (kb == 1) att_wait_barrier<(AT_HD == 64) ? 3 : 0>();        // K (hd 32: K and V) of block 1
    // S^T block = K_block Q^T : sacc[kt][t] rows = keys 96*kb + 16*kt + 4g + r, col = query
    f32x4 sacc[KB_TILES][AT_QT];
#pragma unroll
    for (int kt = 0; kt < KB_TILES; ++kt) {
      // keys >= N (zero rows in LDS) start at -inf, so their scores stay -inf and their weights 0
      const int key_lo = (kb * KB_TILES + kt) * 16 + g * 4;
      const f32x4 init = FULL ? f32x4{0.f, 0.f, 0.f, 0.f}
                              : f32x4{key_lo + 0 < N ? 0.f : -__builtin_inff(), key_lo + 1 < N ? 0.f : -__builtin_inff(),
                                      key_lo + 2 < N ? 0.f : -__builtin_inff(), key_lo + 3 < N ? 0.f : -__builtin_inff()};
#pragma unroll
      for (int t = 0; t < AT_QT; ++t) sacc[kt][t] = init;
      const int r = (kb * KB_TILES + kt) * 16 + lrow;
#pragma unroll
      for (int s = 0; s < G::KS; ++s) {
        const uint4 kf = *reinterpret_cast<const uint4 *>(Ks + r * AT_KROW + (G::swz(r, 4 * s + g) << 4));
#pragma unroll
        for (int t = 0; t < AT_QT; ++t)
          sacc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&kf),
                                                                *reinterpret_cast<const bf16x8 *>(&qf[t][s]),
                                                                sacc[kt][t], 0, 0, 0);
      }
      // the rest of K / V is requested here, spread over the first product, so that no wave sits in the (memory-
      // throttled) issue of all 18 loads before it computes
      if (kb == 0) {
#pragma unroll
        for (int i = kt * NLATE / KB_TILES; i < (kt + 1) * NLATE / KB_TILES; ++i) issue_piece(1 + i / 3, i % 3);
      }
    }
    // online softmax over this block's keys, per query column.  VALU diet (this phase, not the MFMAs,
    // bounds the kernel): padded keys are masked through the accumulator init (below), scores are
    // pre-scaled by c = log2(e)/sqrt(hd) so each probability is one FMA + one raw v_exp_f32.
#pragma unroll
    for (int t = 0; t < AT_QT; ++t) {
      float mb = sacc[0][t][0];
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mb = fmaxf(mb, sacc[kt][t][r]);
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run[t], mb * scale_log2e);   // running max of the SCALED scores; finite
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);  // block 0: exp2(-inf) = 0
      // two scores per instruction: v_pk_fma_f32 for the exponent, v_pk_add_f32 for the partial sums
      att_f32x2 l2 = {0.f, 0.f};
      const att_f32x2 c2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 4
          const att_f32x2 e = {fmaf(sacc[kt][t][r], scale_log2e, -m_new), fmaf(sacc[kt][t][r + 1], scale_log2e, -m_new)};
#else
          const att_f32x2 e = __builtin_elementwise_fma((att_f32x2){sacc[kt][t][r], sacc[kt][t][r + 1]}, c2, nm2);
#endif
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 2
          const att_f32x2 pv = e;   // ablation: no exponentials
#else
          const att_f32x2 pv = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
#endif
          sacc[kt][t][r] = pv.x;
          sacc[kt][t][r + 1] = pv.y;
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 4
          l2.x += pv.x;
          l2.x += pv.y;
#else
          l2 += pv;
#endif
        }
      const float l = l2.x + l2.y;
      l_run[t] = l_run[t] * alpha + l;
      m_run[t] = m_new;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        oacc[dt][t][0] *= alpha; oacc[dt][t][1] *= alpha; oacc[dt][t][2] *= alpha; oacc[dt][t][3] *= alpha;
      }
    }
    // O^T += V_block^T P_block^T : rows = dims 16*dt + 4g + r, col = query
    if constexpr (AT_HD == 64) {
      if (kb == 0) att_wait_barrier<6>(); else att_wait_barrier<0>();   // V of this block
    }
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
      for (int dt = 0; dt < G::DT; ++dt) {
        // A operand = V^T fragment: lane (dim = lrow, g) needs keys {4g..4g+3} and {16+4g..16+4g+3}
        // of this 32-key step for its dim.  Each 16-lane group g issues one transposing read per key
        // quartet: lane 4q+p of the group addresses row (key) q, dims 4p..4p+3 of the 16-dim block;
        // lane i receives dim i of the 4 keys.  (EXEC is all ones here; addresses are 8-B aligned.)
        const int tq = lrow >> 2, tp = lrow & 3;
        const int key0 = kb * (KB_TILES * 16) + 32 * u + 4 * g + tq;
        const int ch = 2 * dt + (tp >> 1);
        const int a0 = key0 * AT_KROW + (G::swz(key0, ch) << 4) + (tp & 1) * 8;
        const int key1 = key0 + 16;
        const int a1 = key1 * AT_KROW + (G::swz(key1, ch) << 4) + (tp & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(Vs + a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(Vs + a1));
        uint4 vf;
        vf.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        vf.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        vf.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        vf.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
#pragma unroll
        for (int t = 0; t < AT_QT; ++t)
          oacc[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&vf),
                                                                *reinterpret_cast<const bf16x8 *>(&pf[t]),
                                                                oacc[dt][t], 0, 0, 0);
      }
    }
  }
  ATT_STAMP(t3);
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
    if (!FULL && q >= N) continue;
    if (fp8_inv_scale > 0.f) {   // out is an e4m3 byte tensor [B*N, C]
      unsigned char *orow8 = reinterpret_cast<unsigned char *>(out) + ((size_t)b * N + q) * C + h * AT_HD;
      const float sc = inv_l[t] * fp8_inv_scale;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
        *reinterpret_cast<unsigned *>(orow8 + dt * 16 + g * 4) =
            att_pack_fp8x4(oacc[dt][t][0] * sc, oacc[dt][t][1] * sc, oacc[dt][t][2] * sc, oacc[dt][t][3] * sc);
      continue;
    }
    bf16_t *orow = out + ((size_t)b * N + q) * C + h * AT_HD;
    if constexpr (G::DT == 4 && FULL) {
      // head_dim 64: a head's output row is ONE aligned 128-byte line.  What a store costs is the lines it touches
      // (tools/ubench/store_bw.hip), and 8 bytes per lane is 16 rows x 32 B = 16 lines per 512 B.  As in
      // gemm_quad_stream_kernel: two v_permlane16_swap_b32 per pair of dim tiles give a lane 8 consecutive dims, rows
      // lrow / lrow ^ 1 then trade halves by DPP so that one 16-byte store holds the whole line of 8 rows: 2 stores of
      // 8 lines per query tile instead of 4 stores of 16.
      unsigned w[8];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        w[2 * dt] = pack_bf16x2(oacc[dt][t][0] * inv_l[t], oacc[dt][t][1] * inv_l[t]);
        w[2 * dt + 1] = pack_bf16x2(oacc[dt][t][2] * inv_l[t], oacc[dt][t][3] * inv_l[t]);
      }
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      u32x4 PQ[2];
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {       // dim tiles 2 jp, 2 jp + 1: even g -> dims 4 g .. + 7 of tile 2 jp, odd g -> of tile 2 jp + 1
        const auto r0 = __builtin_amdgcn_permlane16_swap(w[4 * jp], w[4 * jp + 2], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(w[4 * jp + 1], w[4 * jp + 3], false, false);
        PQ[jp] = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      const bool odd = (lrow & 1) != 0;
      u32x4 X, Y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned give = odd ? PQ[0][e] : PQ[1][e];                                   // what lane lrow ^ 1 needs
        const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
        X[e] = odd ? got : PQ[0][e];       // the even row: dims 0 .. 31 from the even lane, 32 .. 63 from the odd lane
        Y[e] = odd ? PQ[1][e] : got;       // the odd row
      }
      const int lane_dim = ((g & 1) ? 16 + (g - 1) * 4 : g * 4) + (odd ? 32 : 0);
      bf16_t *even_row = out + ((size_t)b * N + (q & ~1)) * C + h * AT_HD + lane_dim;
      *reinterpret_cast<u32x4 *>(even_row) = X;
      *reinterpret_cast<u32x4 *>(even_row + C) = Y;
      continue;
    }
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      uint2 pk;
      pk.x = pack_bf16x2(oacc[dt][t][0] * inv_l[t], oacc[dt][t][1] * inv_l[t]);
      pk.y = pack_bf16x2(oacc[dt][t][2] * inv_l[t], oacc[dt][t][3] * inv_l[t]);
      *reinterpret_cast<uint2 *>(orow + dt * 16 + g * 4) = pk;
    }
  }
#ifdef PP_ATT_STAMPS
  {
    ATT_STAMP(t4);
    if (tid == 0) {
      unsigned long long *o = reinterpret_cast<unsigned long long *>(out + (size_t)gridDim.x / heads * N * C) + (size_t)blockIdx.x * 8;
      o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t4 - t0; o[5] = t0;
    }
  }
#endif
}

// ---- streaming form: any N, head_dim 32 / 64 / 80.  Workgroup = NW waves x ST_QT query tiles (4 x 2 = 128 or
// 3 x 3 = 144 queries, whichever wastes fewer query rows: N = 432 is 3 x 144) of one (crop, head); key blocks of ST_KB
// rows of K and V go HBM/L2 -> LDS by LDS-DMA into a two-deep ring -- block kb + 1 is requested before block kb is
// computed, one barrier per block -- then exactly the block body of the kernel above.  Rows are addressed
// block-relative in LDS; masks use the global key.
template <int HD, int NW, int ST_QT, int ST_KB>
__global__ __launch_bounds__(NW * 64, 3) void attention_stream_kernel(const bf16_t *__restrict__ qkv,
                                                                  bf16_t *__restrict__ out, int N, int heads,
                                                                  int qblocks, int nprob, float scale_log2e,
                                                                      float fp8_inv_scale, int headmajor) {
  using G = AttGeom<HD>;
  constexpr int KROW = G::KROW;
  constexpr int BLK_BYTES = ST_KB * KROW;
  static_assert(BLK_BYTES % 1024 == 0, "a key block is a whole number of 1-KiB DMA pieces");
  constexpr int NPIECE = BLK_BYTES / 1024;
  constexpr int KB_TILES = ST_KB / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware order: workgroups go round-robin to the 8 XCDs, each with its own L2, so the query blocks of one
  // (crop, head) are made consecutive WITHIN an XCD's share: its K / V blocks cross the fabric once, not qblocks times
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;
  const int qb = wx % qblocks, bh = (wx / qblocks) * 8 + xcd;
  if (bh >= nprob) return;                   // the grid is rounded up to whole groups of 8 problems
  const int b = bh / heads, h = bh - b * heads;
  const int C = heads * HD;
  // row-major qkv [B*N][3][heads][HD] (timm's layout): a head's rows sit 3 C apart.  Head-major qkv
  // [3][heads][B*N][HD] (PP_EPI_HEADMAJOR of the qkv GEMM): they are contiguous -- a 160-byte head row (HD = 80) then
  // fills whole cache lines instead of touching 2 - 3 of them at a 7 680-byte stride.
  const size_t plane = (size_t)heads * (nprob / heads) * N * HD;          // one of q / k / v, head-major
  const int ld = headmajor ? HD : 3 * C;
  const size_t kofs = headmajor ? plane : (size_t)C, vofs = 2 * kofs;
  const bf16_t *base = headmajor ? qkv + ((size_t)h * (nprob / heads) + b) * N * HD : qkv + (size_t)b * N * ld + h * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  // Q fragments (B operand): lane (col q = lrow, g) holds Q[q][32s + 8g .. +7]; dims >= HD are zero
  const int q0 = (qb * NW + wave) * (ST_QT * 16);
  uint4 qf[ST_QT][G::KS];
#pragma unroll
  for (int t = 0; t < ST_QT; ++t) {
    const int q = q0 + t * 16 + lrow;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      qf[t][s] = make_uint4(0, 0, 0, 0);
      if (q < N && (s < G::KS - 1 || g < G::KLIVE))
        qf[t][s] = *reinterpret_cast<const uint4 *>(base + (size_t)q * ld + s * 32 + g * 8);
    }
  }
  float m_run[ST_QT], l_run[ST_QT];
  f32x4 oacc[G::DT][ST_QT];
#pragma unroll
  for (int t = 0; t < ST_QT; ++t) {
    m_run[t] = -__builtin_inff();
    l_run[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) oacc[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nkb = (N + ST_KB - 1) / ST_KB;   // every block holds at least one real key
  // K / V block kb -> ring slot kb & 1: piece p = 64 consecutive 16-B chunks of the row-major block image
  auto issue_block = [&](int kb) __attribute__((always_inline)) {
    const unsigned slot = lds0 + (kb & 1) * (2 * BLK_BYTES);
#pragma unroll
    for (int j = 0; j < (NPIECE + NW - 1) / NW; ++j) {
      const int pc = j * NW + wave;
      if (pc < NPIECE) {
        const int c = pc * 64 + lane;
        const int row = c / G::CHUNKS, pchunk = c - row * G::CHUNKS;
        const int key = kb * ST_KB + row;
        const bf16_t *src = base + (size_t)key * ld + G::swz(row, pchunk) * 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane(slot + pc * 1024);
        const void *zsrc = g_att_zero + lane * 16;
        att_glds16(key < N ? (const void *)(src + kofs) : zsrc, dst);
        att_glds16(key < N ? (const void *)(src + vofs) : zsrc, dst + BLK_BYTES);
      }
    }
  };
  issue_block(0);
  for (int kb = 0; kb < nkb; ++kb) {
    // block kb has landed (the only DMA in flight) for every wave, and every wave is done with block kb - 1, whose
    // slot the next request overwrites
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (kb + 1 < nkb) issue_block(kb + 1);
    const char *Ks = smem + (kb & 1) * (2 * BLK_BYTES);
    const char *Vs = Ks + BLK_BYTES;

    // ---- S^T block = K_block Q^T
    f32x4 sacc[KB_TILES][ST_QT];
#pragma unroll
    for (int kt = 0; kt < KB_TILES; ++kt) {
      const int key_lo = kb * ST_KB + kt * 16 + g * 4;
      const f32x4 init = f32x4{key_lo + 0 < N ? 0.f : -__builtin_inff(), key_lo + 1 < N ? 0.f : -__builtin_inff(),
                               key_lo + 2 < N ? 0.f : -__builtin_inff(), key_lo + 3 < N ? 0.f : -__builtin_inff()};
#pragma unroll
      for (int t = 0; t < ST_QT; ++t) sacc[kt][t] = init;
      const int r = kt * 16 + lrow;
#pragma unroll
      for (int s = 0; s < G::KS; ++s) {
        uint4 kf = make_uint4(0, 0, 0, 0);
        if (s < G::KS - 1 || g < G::KLIVE)
          kf = *reinterpret_cast<const uint4 *>(Ks + r * KROW + (G::swz(r, 4 * s + g) << 4));
#pragma unroll
        for (int t = 0; t < ST_QT; ++t)
          sacc[kt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&kf),
                                                                *reinterpret_cast<const bf16x8 *>(&qf[t][s]),
                                                                sacc[kt][t], 0, 0, 0);
      }
    }
    // ---- online softmax over the block (see the one-shot kernel)
#pragma unroll
    for (int t = 0; t < ST_QT; ++t) {
      float mb = sacc[0][t][0];
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mb = fmaxf(mb, sacc[kt][t][r]);
      mb = fmaxf(mb, __shfl_xor(mb, 16, 64));
      mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
      const float m_new = fmaxf(m_run[t], mb * scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
      // two scores per instruction: v_pk_fma_f32 for the exponent, v_pk_add_f32 for the partial sums
      att_f32x2 l2 = {0.f, 0.f};
      const att_f32x2 c2 = {scale_log2e, scale_log2e}, nm2 = {-m_new, -m_new};
#pragma unroll
      for (int kt = 0; kt < KB_TILES; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 4
          const att_f32x2 e = {fmaf(sacc[kt][t][r], scale_log2e, -m_new), fmaf(sacc[kt][t][r + 1], scale_log2e, -m_new)};
#else
          const att_f32x2 e = __builtin_elementwise_fma((att_f32x2){sacc[kt][t][r], sacc[kt][t][r + 1]}, c2, nm2);
#endif
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 2
          const att_f32x2 pv = e;   // ablation: no exponentials
#else
          const att_f32x2 pv = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
#endif
          sacc[kt][t][r] = pv.x;
          sacc[kt][t][r + 1] = pv.y;
#if defined(PP_ATT_ABL) && PP_ATT_ABL == 4
          l2.x += pv.x;
          l2.x += pv.y;
#else
          l2 += pv;
#endif
        }
      const float l = l2.x + l2.y;
      l_run[t] = l_run[t] * alpha + l;
      m_run[t] = m_new;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        oacc[dt][t][0] *= alpha; oacc[dt][t][1] *= alpha; oacc[dt][t][2] *= alpha; oacc[dt][t][3] *= alpha;
      }
    }
    // ---- O^T += V_block^T P_block^T
#pragma unroll
    for (int u = 0; u < KB_TILES / 2; ++u) {
      uint4 pf[ST_QT];
#pragma unroll
      for (int t = 0; t < ST_QT; ++t) {
        pf[t].x = pack_bf16x2(sacc[2 * u][t][0], sacc[2 * u][t][1]);
        pf[t].y = pack_bf16x2(sacc[2 * u][t][2], sacc[2 * u][t][3]);
        pf[t].z = pack_bf16x2(sacc[2 * u + 1][t][0], sacc[2 * u + 1][t][1]);
        pf[t].w = pack_bf16x2(sacc[2 * u + 1][t][2], sacc[2 * u + 1][t][3]);
      }
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        const int tq = lrow >> 2, tp = lrow & 3;
        const int key0 = 32 * u + 4 * g + tq;            // block-relative row
        const int ch = 2 * dt + (tp >> 1);
        const int a0 = key0 * KROW + (G::swz(key0, ch) << 4) + (tp & 1) * 8;
        const int key1 = key0 + 16;
        const int a1 = key1 * KROW + (G::swz(key1, ch) << 4) + (tp & 1) * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(Vs + a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4 *)(Vs + a1));
        uint4 vf;
        vf.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
        vf.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
        vf.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
        vf.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
#pragma unroll
        for (int t = 0; t < ST_QT; ++t)
          oacc[dt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&vf),
                                                                *reinterpret_cast<const bf16x8 *>(&pf[t]),
                                                                oacc[dt][t], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int t = 0; t < ST_QT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv_l = 1.0f / l;
    const int q = q0 + t * 16 + lrow;
    if (fp8_inv_scale > 0.f) {
      if (q >= N) continue;
      unsigned char *orow8 = reinterpret_cast<unsigned char *>(out) + ((size_t)b * N + q) * C + h * HD;
      const float sc = inv_l * fp8_inv_scale;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
        *reinterpret_cast<unsigned *>(orow8 + dt * 16 + g * 4) =
            att_pack_fp8x4(oacc[dt][t][0] * sc, oacc[dt][t][1] * sc, oacc[dt][t][2] * sc, oacc[dt][t][3] * sc);
      continue;
    }
    if constexpr (G::DT < 4) {      // head_dim 32: one pair of dim tiles only; the 8-byte form stays (the wide form returned
      if (q >= N) continue;         // NaN rows at head_dim 32 in this kernel -- rows q with (q & 0x14) == 0x10 -- for a reason not
      bf16_t *orow = out + ((size_t)b * N + q) * C + h * HD;      // found in the ISA; head_dim 64 / 80 are bit-identical to this form)
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        uint2 pk;
        pk.x = pack_bf16x2(oacc[dt][t][0] * inv_l, oacc[dt][t][1] * inv_l);
        pk.y = pack_bf16x2(oacc[dt][t][2] * inv_l, oacc[dt][t][3] * inv_l);
        *reinterpret_cast<uint2 *>(orow + dt * 16 + g * 4) = pk;
      }
      continue;
    }
    // bf16 rows leave 16 bytes per lane.  What a store costs is the 128-byte lines it touches (tools/ubench/store_bw.hip):
    // 8 bytes per lane is 16 rows x 32 B = 16 lines per 512 B.  Two v_permlane16_swap_b32 per pair of dim tiles give a lane
    // 8 consecutive dims (64 B per row and store); where there are two pairs, rows lrow / lrow ^ 1 trade one each by DPP
    // so that a store holds 128 contiguous bytes of 8 rows.  Every lane takes part in the exchanges (rows past N hold
    // garbage and are not stored): q is a lane's own row, the exchanged chunks belong to rows q & ~1 and q | 1.
    {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int NP = G::DT / 2;                  // pairs of dim tiles
      unsigned w[2 * G::DT];
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        w[2 * dt] = pack_bf16x2(oacc[dt][t][0] * inv_l, oacc[dt][t][1] * inv_l);
        w[2 * dt + 1] = pack_bf16x2(oacc[dt][t][2] * inv_l, oacc[dt][t][3] * inv_l);
      }
      u32x4 ch[NP > 0 ? NP : 1];
#pragma unroll
      for (int jp = 0; jp < NP; ++jp) {       // even g: dims 4 g .. + 7 of tile 2 jp; odd g: dims 4 (g - 1) .. + 7 of tile 2 jp + 1
        const auto r0 = __builtin_amdgcn_permlane16_swap(w[4 * jp], w[4 * jp + 2], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(w[4 * jp + 1], w[4 * jp + 3], false, false);
        ch[jp] = u32x4{r0[0], r1[0], r0[1], r1[1]};
      }
      const bool odd = (lrow & 1) != 0;
      const int lane_dim = (g & 1) ? 16 + (g - 1) * 4 : g * 4;
      bf16_t *head0 = out + (size_t)b * N * C + h * HD;
#pragma unroll
      for (int pp = 0; pp < NP / 2; ++pp) {
        u32x4 X, Y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned give = odd ? ch[2 * pp][e] : ch[2 * pp + 1][e];
          const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
          X[e] = odd ? got : ch[2 * pp][e];
          Y[e] = odd ? ch[2 * pp + 1][e] : got;
        }
        const int dim = (2 * pp + (odd ? 1 : 0)) * 32 + lane_dim;
        if ((q & ~1) < N) *reinterpret_cast<u32x4 *>(head0 + (size_t)(q & ~1) * C + dim) = X;
        if ((q | 1) < N) *reinterpret_cast<u32x4 *>(head0 + (size_t)(q | 1) * C + dim) = Y;
      }
      if (q < N) {
        if constexpr (NP & 1) *reinterpret_cast<u32x4 *>(head0 + (size_t)q * C + (NP - 1) * 32 + lane_dim) = ch[NP - 1];
        if constexpr (G::DT & 1) *reinterpret_cast<uint2 *>(head0 + (size_t)q * C + (G::DT - 1) * 16 + g * 4) = uint2{w[2 * G::DT - 2], w[2 * G::DT - 1]};
      }
    }
  }
}

template <int HD, int NW, int QT, int KB>
static int launch_stream_cfg(const void *qkv, void *out, int B, int N, int heads, float fp8_inv_scale, hipStream_t s,
                             int headmajor) {
  const int qblocks = (N + NW * QT * 16 - 1) / (NW * QT * 16);
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
  const int nprob = B * heads;
  constexpr int LDS = 4 * KB * AttGeom<HD>::KROW;     // two slots of K + V
  hipLaunchKernelGGL((attention_stream_kernel<HD, NW, QT, KB>), dim3((unsigned)((size_t)((nprob + 7) / 8) * 8 * qblocks)),
                     dim3(NW * 64), LDS, s, (const bf16_t *)qkv, (bf16_t *)out, N, heads, qblocks, nprob, scale_log2e,
                     fp8_inv_scale, headmajor);
  return 0;
}

// 144-query workgroups (3 waves x 3 tiles) when they waste fewer query rows than 128-query ones (4 waves x 2); the
// key block shrinks to 32 rows there (hd 64 / 80: 3 score tiles per key tile would not fit 168 registers otherwise).
// Lab builds only (-DPP_ATT_LAB, tools/): the environment variable PP_ATT_STREAM = 4 / 3 forces one form.  The shipped
// library reads no environment.
template <int HD>
static int launch_stream(const void *qkv, void *out, int B, int N, int heads, float fp8_inv_scale, hipStream_t s,
                         int headmajor = 0) {
#ifdef PP_ATT_LAB
  static const int forced = []() { const char *e = getenv("PP_ATT_STREAM"); return e ? atoi(e) : 0; }();
#else
  constexpr int forced = 0;
#endif
  const int waste128 = (N + 127) / 128 * 128 - N, waste144 = (N + 143) / 144 * 144 - N;
  const bool use144 = forced ? forced != 4 : waste144 < waste128;
  if (use144) return launch_stream_cfg<HD, 3, 3, 32>(qkv, out, B, N, heads, fp8_inv_scale, s, headmajor);
  return launch_stream_cfg<HD, 4, 2, 64>(qkv, out, B, N, heads, fp8_inv_scale, s, headmajor);
}

}  // namespace pp

static int attention_dispatch(const void *qkv, void *out, int B, int N, int heads, int hd, int dtype,
                              float fp8_inv_scale, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && N > 0 && heads > 0 && hd > 0, "pp_attention: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(qkv && out, "pp_attention: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PP_BF16) {
    if ((hd == 64 || hd == 32) && N <= AT_NMAX && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 7) == 0) {
      const float scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
      const dim3 grid(B * heads), block(256);
      const bf16_t *qp = (const bf16_t *)qkv;
      bf16_t *op = (bf16_t *)out;
      // FULL at head_dim 64 stores whole 128-byte output lines, 16 bytes per lane: needs a 16-byte aligned `out`
      const bool full = N == AT_NMAX && (hd != 64 || fp8_inv_scale > 0.f || ((uintptr_t)out & 15) == 0);
      if (hd == 64 && full)
        hipLaunchKernelGGL((attention_mfma_kernel<64, true>), grid, block, AttGeom<64>::LDS, s, qp, op, N, heads, scale_log2e, fp8_inv_scale);
      else if (hd == 64)
        hipLaunchKernelGGL((attention_mfma_kernel<64, false>), grid, block, AttGeom<64>::LDS, s, qp, op, N, heads, scale_log2e, fp8_inv_scale);
      else if (full)
        hipLaunchKernelGGL((attention_mfma_kernel<32, true>), grid, block, AttGeom<32>::LDS, s, qp, op, N, heads, scale_log2e, fp8_inv_scale);
      else
        hipLaunchKernelGGL((attention_mfma_kernel<32, false>), grid, block, AttGeom<32>::LDS, s, qp, op, N, heads, scale_log2e, fp8_inv_scale);
      PP_CHECK_LAUNCH("attention_mfma_kernel");
      return 0;
    }
    if ((hd == 32 || hd == 64 || hd == 80) && ((uintptr_t)qkv & 15) == 0 &&
        ((uintptr_t)out & (fp8_inv_scale > 0.f ? 7 : 15)) == 0 &&          // bf16 rows leave 16 bytes per lane
        ((size_t)B * heads + 8) * ((N + 127) / 128) < (1ull << 31)) {
      if (hd == 80) launch_stream<80>(qkv, out, B, N, heads, fp8_inv_scale, s);
      else if (hd == 64) launch_stream<64>(qkv, out, B, N, heads, fp8_inv_scale, s);
      else launch_stream<32>(qkv, out, B, N, heads, fp8_inv_scale, s);
      PP_CHECK_LAUNCH("attention_stream_kernel");
      return 0;
    }
    PP_REQUIRE(fp8_inv_scale <= 0.f, "pp_attention_fp8out: needs head_dim 32 / 64 / 80 and aligned operands");
    return attention_valu<bf16_t>(qkv, out, B, N, heads, hd, s);
  }
  PP_REQUIRE(fp8_inv_scale <= 0.f, "pp_attention_fp8out: bf16 qkv only");
  if (dtype == PP_F32) return attention_valu<float>(qkv, out, B, N, heads, hd, s);
  return fail("pp_attention: bad dtype %d", dtype);
}

extern "C" int pp_attention(const void *qkv, void *out, int B, int N, int heads, int hd, int dtype,
                            void *stream) {
  return attention_dispatch(qkv, out, B, N, heads, hd, dtype, 0.f, stream);
}

extern "C" int pp_attention_fp8out(const void *qkv, unsigned char *out, int B, int N, int heads, int hd,
                                   float inv_scale, void *stream) {
  if (!(inv_scale > 0.f)) return pp::fail("pp_attention_fp8out: inv_scale must be positive");
  return attention_dispatch(qkv, out, B, N, heads, hd, PP_BF16, inv_scale, stream);
}

/* Head-major qkv: see include/probpose_hip.h.  The streaming MFMA kernel only (bf16, head_dim 32 / 64 / 80, any N). */
extern "C" int pp_attention_headmajor(const void *qkv, void *out, int B, int N, int heads, int hd, void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && N > 0 && heads > 0 && hd > 0, "pp_attention_headmajor: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(qkv && out, "pp_attention_headmajor: null pointer");
  PP_REQUIRE((hd == 32 || hd == 64 || hd == 80) && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "pp_attention_headmajor: head_dim 32 / 64 / 80, qkv 16-byte aligned (got head_dim %d)", hd);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 80) launch_stream<80>(qkv, out, B, N, heads, 0.f, s, 1);
  else if (hd == 64) launch_stream<64>(qkv, out, B, N, heads, 0.f, s, 1);
  else launch_stream<32>(qkv, out, B, N, heads, 0.f, s, 1);
  PP_CHECK_LAUNCH("attention_stream_kernel (head-major)");
  return 0;
}
