// "Quad" GEMM forms for gfx950 (pp_gemm tiles 18 - 20): C[M,N] = act(A[M,K] W[N,K]^T + bias), bf16 in / bf16 out, fp32
// accumulate.  The qkv and fc1 layers of the ViT blocks: the shapes on which the vendor library led the 8-wave forms.
//
// FOUR waves per workgroup (2 x 2), ONE wave per SIMD, each wave owning a (16 TM) x (16 TN) output tile in 192 - 216
// accumulator registers (128 x 96, 96 x 144, 96 x 128): the wave-tile shapes of the vendor's macro tiles
// (profiles/r03_gemm_vs_vendor.txt).  Against the 8-wave forms of pp_gemm.hip (96 x 48 wave tiles) a K-step reads half
// the LDS bytes per flop.
//
// With one wave per SIMD nothing hides a wave's own latencies, so (1) the loop is pipelined in registers: while the
// MFMAs of K-tile t run from fragment set X, the fragments of K-tile t + 1 are read into set Y (two ds_read_b128 per row
// of MFMAs) and the LDS-DMA pieces of K-tile t + 4 go into the ring buffer K-tile t has just left (K-tiles 32 deep,
// 64-byte rows, the 4-chunk swizzle of gemm_duo_kernel, FOUR ring buffers of (BM + BN) x 64 B: a K-tile is requested
// three iterations before its fragments are read); (2) the MFMAs are inline asm with the accumulator tied in the AGPRs
// -- hipcc's allocator otherwise shuffles a 200-register accumulator through v_accvgpr_mov inside the loop -- which
// makes every MFMA hazard the author's business (see the s_nop at the loop ends); (3) the K-loop has NO branch but
// its back edge: a taken branch costs a lone wave ~26 cycles of issue, measured, so there is no predicate on a DMA
// piece and no choice between wait immediates at run time; (4) the kernel is a persistent stream (below).
// What the lab measurements of the per-launch form say about a K-tile of the 192 x 288 tile (54 MFMAs = 864 cycles per
// wave): 1 361 cycles as built; 8 DMA pieces cost 31 cycles each, the fragment reads and the barrier 75, loop control
// and waits the rest (profiles/r03_gemm_quad_experiments.txt).
//
// The per-launch form (tiles 15 - 17) is kept for lab builds (tools/build_lab.sh), where its ablation switches live.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "pp_gemm_shared.h"

namespace pp {

constexpr int quad_lds_bytes(int TM, int TN) {
  const int BM = 32 * TM, BN = 32 * TN;
  const int ring = 4 * (BM + BN) * 64, ctile = BM * (BN * 2 + 16);
  return (ring > ctile ? ring : ctile) + BN * 4;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The MFMA from inline asm with the accumulator TIED in the accumulation registers ("+a"): with 256 of the 256 AGPRs
// holding the 128 x 128 wave tile hipcc's allocator otherwise un-ties destination and addend of a third of the MFMAs and
// shuffles tiles through v_accvgpr_mov / s_nop inside the loop (seen in the ISA of the builtin form: 136 moves and 74
// nops per two K-tiles).  volatile also fixes the issue order the code is written in.
__device__ __forceinline__ void mfma_bf16(f32x4 &c, const u32x4 &w, const u32x4 &a) {
#ifdef PP_QUAD_BUILTIN_MFMA
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&w), *reinterpret_cast<const bf16x8 *>(&a), c, 0, 0, 0);
#else
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
#endif
}

// LDS-DMA pieces: 64 lanes x 16 B each from (uniform 64-bit base + per-lane 32-bit offset) to LDS offset M0 + lane * 16.
// The scalar-base form keeps the K-loop free of 64-bit vector address arithmetic (the K offset is a scalar add on the
// base).  A burst (1 - 4 pieces) is issued only when bit BIT of the wave-uniform `mask` is set.  The predicate lives INSIDE
// the asm statement on purpose: with 192 - 256 accumulator registers tied to asm MFMAs, any C++-level branch inside the
// K-loop makes hipcc's allocator split the accumulators' live ranges (hundreds of v_accvgpr moves and scratch spills per
// K-tile; seen in the ISA).  M0 is written without save / restore: nothing else in this translation unit uses it.
#ifndef PP_QUAD_BURST
#define PP_QUAD_BURST 0     /* lab (per-launch form): 0: every wave one piece per slot (2 722 cycles per two K-tiles of the
                               192 x 288 tile); 1: staggered bursts skipped by a branch (2 841); 2: skipped by EXEC = 0 (4 575) */
#endif
#if PP_QUAD_BURST == 2
#define PP_QB_OPEN(BIT) "s_bitcmp1_b32 %0, " #BIT "\n\ts_cselect_b64 exec, -1, 0\n\t"
#define PP_QB_CLOSE "s_mov_b64 exec, -1"
#else
#define PP_QB_OPEN(BIT) "s_bitcmp1_b32 %0, " #BIT "\n\ts_cbranch_scc0 .Lsk_%=\n\t"
#define PP_QB_CLOSE ".Lsk_%=:"
#endif
#define PP_QB_PIECE(L, O, B) "s_mov_b32 m0, " L "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " O ", " B "\n\t"
struct QuadPiece {
  unsigned voff;                 // per-lane byte offset from the base
  unsigned long long sbase;      // uniform
  unsigned lds;                  // uniform LDS byte offset of the piece
};
// one piece, unconditionally (the stream form: a K-tile always has a successor to request)
__device__ __forceinline__ void glds_piece(const QuadPiece &a) {
  asm volatile(PP_QB_PIECE("%0", "%1", "%2") : : "s"(a.lds), "v"(a.voff), "s"(a.sbase) : "memory");
}
template <int BIT>
__device__ __forceinline__ void glds_burst(int mask, const QuadPiece &a) {
  asm volatile(PP_QB_OPEN(%c1) PP_QB_PIECE("%2", "%3", "%4") PP_QB_CLOSE
               :
               : "s"(__builtin_amdgcn_readfirstlane(mask)), "n"(BIT), "s"(a.lds), "v"(a.voff), "s"(a.sbase)
               : "memory", "scc");
}
template <int BIT>
__device__ __forceinline__ void glds_burst(int mask, const QuadPiece &a, const QuadPiece &b, const QuadPiece &c) {
  asm volatile(PP_QB_OPEN(%c1) PP_QB_PIECE("%2", "%3", "%4") PP_QB_PIECE("%5", "%6", "%7") PP_QB_PIECE("%8", "%9", "%10")
                   PP_QB_CLOSE
               :
               : "s"(__builtin_amdgcn_readfirstlane(mask)), "n"(BIT), "s"(a.lds), "v"(a.voff), "s"(a.sbase), "s"(b.lds),
                 "v"(b.voff), "s"(b.sbase), "s"(c.lds), "v"(c.voff), "s"(c.sbase)
               : "memory", "scc");
}
template <int BIT>
__device__ __forceinline__ void glds_burst(int mask, const QuadPiece &a, const QuadPiece &b, const QuadPiece &c,
                                           const QuadPiece &d) {
  asm volatile(PP_QB_OPEN(%c1) PP_QB_PIECE("%2", "%3", "%4") PP_QB_PIECE("%5", "%6", "%7") PP_QB_PIECE("%8", "%9", "%10")
                   PP_QB_PIECE("%11", "%12", "%13") PP_QB_CLOSE
               :
               : "s"(__builtin_amdgcn_readfirstlane(mask)), "n"(BIT), "s"(a.lds), "v"(a.voff), "s"(a.sbase), "s"(b.lds),
                 "v"(b.voff), "s"(b.sbase), "s"(c.lds), "v"(c.voff), "s"(c.sbase), "s"(d.lds), "v"(d.voff), "s"(d.sbase)
               : "memory", "scc");
}
// Top of a K-tile: s_waitcnt vmcnt(rem >= 2 ? N2 : rem == 1 ? N1 : 0) -- the choice between immediates is a branch inside
// the asm statement, as above -- then lgkmcnt(0) and the barrier as BUILTINS: hipcc has to see that the fragment reads of
// the previous iteration are complete.  With the lgkmcnt wait hidden in asm its wait-count pass assumed them still in
// flight at the loop header and guarded this iteration's MFMAs with counted lgkmcnt waits that in fact waited for the
// reads just issued for the NEXT K-tile: ~500 of 1370 cycles per K-tile (measured: the loop ran as long with no DMA piece).
template <int N2, int N1>
__device__ __forceinline__ void wait_tiles_barrier(int rem) {
  asm volatile("s_cmp_lt_i32 %0, 2\n\ts_cbranch_scc1 .Lw1_%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lwd_%=\n"
               ".Lw1_%=:\n\ts_cmp_lt_i32 %0, 1\n\ts_cbranch_scc1 .Lw0_%=\n\ts_waitcnt vmcnt(%2)\n\ts_branch .Lwd_%=\n"
               ".Lw0_%=:\n\ts_waitcnt vmcnt(0)\n"
               ".Lwd_%=:"
               :
               : "s"(__builtin_amdgcn_readfirstlane(rem)), "n"(N2), "n"(N1)
               : "memory", "scc");
  __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0) only
#ifndef PP_QUAD_ABL_NOBAR      /* lab: no barrier (racy) */
  __builtin_amdgcn_s_barrier();
#endif
}
__device__ __forceinline__ unsigned long long uniform64(const void *p) {
  const unsigned long long v = (unsigned long long)p;
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

#ifdef PP_GEMM_LAB   // the per-launch form: lab builds only (tools/build_lab.sh), where its ablation switches live
template <int TM, int TN, int ACT>   // ACT: 0 none, 1 GELU, 2 ReLU
__global__ __launch_bounds__(256, 1) void gemm_quad_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PP_GEMM_TIMELINE
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long rt_loop0 = 0, rt_loop1 = 0, ct_loop0 = 0, ct_loop1 = 0;
#endif
  constexpr int BM = 32 * TM, BN = 32 * TN, STAGES = 4, BK = 32, RB = 64;   // RB: bytes of K per staged row
  constexpr int PTA = BM / 16, PT = (BM + BN) / 16;    // 1-KiB DMA pieces (16 rows x 64 B) of a K-tile: A rows, then W rows
  constexpr int PMAX = (PT + 3) / 4, PREM = PT % 4;    // a wave issues pieces w, w + 4, ...; the last one only if w < PREM
  constexpr int KA = PTA / 4;                           // pieces k < KA of every wave are A rows, the rest W rows
  constexpr int KH = (PMAX + 1) / 2;                    // pieces per burst (two bursts per wave and K-tile)
  constexpr int A_BYTES = BM * RB, STAGE_BYTES = (BM + BN) * RB, RING = STAGES * STAGE_BYTES;
  constexpr int CS = BN * 2 + 16, CPR = BN / 8;         // staged C row stride (bytes), 16-byte chunks per row
  constexpr int BIAS_OFF = RING > BM * CS ? RING : BM * CS;
  static_assert(PTA % 4 == 0, "A pieces split evenly over the four waves");
  static_assert(4 * PMAX + 2 <= 63, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int epi = p.epilogue;

  int tm, tn;
  if (p.blocked) {   // XCD-blocked order (see gemm_kernel): speed only
    constexpr int RM = 8;
    const int RN = p.rn, RT = RM * RN;
    const int nbm = (p.tiles_m + RM - 1) / RM, nbn = (p.tiles_n + RN - 1) / RN;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = (j / RT) * 8 + x, idx = j % RT;
    if (g >= nbm * nbn) return;
    const int bmi = g / nbn, bni = g - bmi * nbn;
    tm = bmi * RM + idx / RN;
    tn = bni * RN + idx % RN;
    if (tm >= p.tiles_m || tn >= p.tiles_n) return;
  } else {
    tm = blockIdx.x / p.tiles_n;
    tn = blockIdx.x - tm * p.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = p.Kd / BK;                 // even and >= 4 (host)

  // ---- staging: lane -> (row in piece = lane >> 2, physical chunk = lane & 3); logical chunk = physical ^ perm.
  // Address = uniform tile base (A + m0 rows / W + n0 rows, advanced by 64 B per K-tile) + a per-lane 32-bit offset
  // that never changes.  Rows past M / N re-read the last row: their outputs are never stored.
  const int prow = lane >> 2, pchunk = lane & 3;
  const int perm_p = (0x1230 >> (4 * ((prow >> 2) & 3))) & 3;      // {0, 3, 2, 1}[(row >> 2) & 3]
  const int lchunk_off = (pchunk ^ perm_p) * 16;
  unsigned off[PMAX];
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int q = wave + 4 * k;               // wave-uniform
    if (k < KA) off[k] = (unsigned)(min(q * 16 + prow, p.M - 1 - m0) * p.lda * 2 + lchunk_off);
    else off[k] = (unsigned)(min((min(q, PT - 1) - PTA) * 16 + prow, p.N - 1 - n0) * p.ldw * 2 + lchunk_off);
  }
  const unsigned long long baseA = uniform64(p.A + (size_t)m0 * p.lda * 2), baseW = uniform64(p.W + (size_t)n0 * p.ldw * 2);
  const unsigned lds0 = lds_offset_of(smem);
  // a wave without a piece PMAX - 1 (PT not a multiple of 4) re-issues its piece PMAX - 2 to the same place instead: every
  // wave has PMAX pieces per K-tile in its vmcnt queue and the counted waits are the same for all
  const int has_last = (PREM == 0 || wave < PREM) ? 1 : 0;
  if (!has_last) off[PMAX - 1] = off[PMAX - 2];
  const int last_k = has_last ? PMAX - 1 : PMAX - 2;
  unsigned st_lds = 0;
  unsigned long long st_A = 0, st_W = 0;
  auto stage_begin = [&](int kt, int buf) __attribute__((always_inline)) {
    st_lds = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * 1024);
    st_A = baseA + (unsigned)(kt * RB);
    st_W = baseW + (unsigned)(kt * RB);
  };
  auto piece = [&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    static_assert(PMAX - 2 >= KA, "the re-issued piece is a W piece like the one it stands in for");
    return QuadPiece{off[k], k < KA ? st_A : st_W,
                     st_lds + (k == PMAX - 1 ? last_k : k) * 4096};
  };
#define PP_QP(K) piece(std::integral_constant<int, (K)>{})
  // Slot B of a K-tile (B = 0 .. 7, spread evenly over the MFMA sequence).  Staggered (PP_QUAD_BURST 1 / 2): slot B belongs
  // to wave B & 3 and carries half B >> 2 of that wave's pieces; mask bit B says whether this wave issues it.
  // Unstaggered (0): slot B carries piece B of every wave (PMAX <= 8).
  auto stage_slot = [&](auto bc, int mask) __attribute__((always_inline)) {
    constexpr int B = decltype(bc)::value;
#if PP_QUAD_BURST == 0
    if constexpr (B < PMAX) glds_burst<B>(mask, PP_QP(B));
#else
    constexpr int h = B >> 2, k0 = h * KH, n = (h == 0 ? KH : PMAX - KH);
    static_assert(KH == 4 && (PMAX == 7 || PMAX == 8), "bursts of 4 + 3 or 4 + 4 pieces");
    if constexpr (n == 4) glds_burst<B>(mask, PP_QP(k0), PP_QP(k0 + 1), PP_QP(k0 + 2), PP_QP(k0 + 3));
    else glds_burst<B>(mask, PP_QP(k0), PP_QP(k0 + 1), PP_QP(k0 + 2));
#endif
  };
#if PP_QUAD_BURST == 0
  const int my_slots = 0xFF;
#else
  const int my_slots = 0x11 << wave;
#endif
  // top of a K-tile: at most min(rem, 2) K-tiles of this wave's pieces may stay in flight, then everyone's (barrier)
  auto top = [&](int rem) __attribute__((always_inline)) { wait_tiles_barrier<2 * PMAX, PMAX>(rem); };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets inside a stage: row (w? * tile + 16 i + frow), chunk fq ^ perm[(frow >> 2) & 3]
  const int perm_f = (0x1230 >> (4 * ((frow >> 2) & 3))) & 3;
  const unsigned offA = (wm * 16 * TM + frow) * RB + ((fq ^ perm_f) << 4);
  const unsigned offB = A_BYTES + (wn * 16 * TN + frow) * RB + ((fq ^ perm_f) << 4);
  auto read_frags = [&](int buf, u32x4 (&fa)[TM], u32x4 (&fb)[TN]) __attribute__((always_inline)) {
    const char *sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const u32x4 *>(sb + offB + j * 16 * RB);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const u32x4 *>(sb + offA + i * 16 * RB);
  };
  // One K-tile, straight-line code: TM x TN MFMAs from (ca, cb); the fragments of the next K-tile (ring buffer rbuf)
  // go into (na, nb), two ds_read_b128 per row of MFMAs; when `on`, this wave's DMA pieces of K-tile t + 4 leave in two
  // bursts, the eight bursts of the four waves spread evenly over the MFMA sequence: a wave's global_load_lds waits for
  // the CU's one address unit, and four waves that reach their pieces together wait for each other (~55 cycles per
  // piece measured, with one wave per SIMD all of it off the matrix pipe).
  auto iter = [&](u32x4 (&ca)[TM], u32x4 (&cb)[TN], u32x4 (&na)[TM], u32x4 (&nb)[TN], int rbuf, int on)
                  __attribute__((always_inline)) {
    const char *sb = smem + rbuf * STAGE_BYTES;
    [&]<int... MI>(std::integer_sequence<int, MI...>) {
      ([&] {
        constexpr int m = MI, i = m / TN, j = m % TN;
#ifndef PP_QUAD_ABL_NOREAD   /* lab: no fragment prefetch (the next K-tile reuses stale fragments; results wrong) */
        if constexpr (j == 0) {
          na[i] = *reinterpret_cast<const u32x4 *>(sb + offA + i * 16 * RB);
          [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
              if constexpr ((J * TM) / TN == i) nb[J] = *reinterpret_cast<const u32x4 *>(sb + offB + J * 16 * RB);
            }(), ...);
          }(std::make_integer_sequence<int, TN>{});
        }
#endif
        [&]<int... B>(std::integer_sequence<int, B...>) {
          ([&] {
#ifndef PP_QUAD_ABL_NOPIECE   /* lab: no DMA statements at all in the K-loop (not even skipped ones) */
            if constexpr ((B * TM * TN) / 8 == m) stage_slot(std::integral_constant<int, B>{}, on);
#else
            (void)B;
#endif
          }(), ...);
        }(std::make_integer_sequence<int, 8>{});
        mfma_bf16(acc[i][j], cb[j], ca[i]);
      }(), ...);
    }(std::make_integer_sequence<int, TM * TN>{});
  };

  // ---- fill: the bias first (oldest in the queue: every later counted wait retires it), then K-tiles 0 .. 3
  if (epi & PP_EPI_BIAS) {
    constexpr int NB = (BN + 255) / 256;
    if (wave < NB) {
      const int n = min(n0 + wave * 256 + lane * 4, p.N - 4);
      glds16(p.bias + n, __builtin_amdgcn_readfirstlane(lds0 + BIAS_OFF + wave * 1024));
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < STAGES; ++s_) {
    stage_begin(s_, s_);
    [&]<int... B>(std::integer_sequence<int, B...>) {
      (stage_slot(std::integral_constant<int, B>{}, my_slots), ...);
    }(std::make_integer_sequence<int, 8>{});
  }
  u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  wait_vmcnt<3 * PMAX>();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  read_frags(0, fa0, fb0);
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
  ct_loop0 = __builtin_amdgcn_s_memtime();
#endif

  // ---- ONE loop over all K-tiles, two per trip (the two fragment sets swap roles), no C++ branch inside: whether a
  // K-tile still requests one is an operand of the DMA statements, and the last K-tile's fragment prefetch reads a ring
  // buffer nobody uses any more.
  for (int t = 0, b = 0; t < nkt; t += 2) {          // b = ring buffer of K-tile t
    top(nkt - 2 - t);
    stage_begin(t + 4, b);
#ifdef PP_QUAD_ABL_NODMA      /* lab: K-loop without its DMA pieces (results wrong): what the pieces cost */
    iter(fa0, fb0, fa1, fb1, (b + 1) & 3, 0);
#else
    iter(fa0, fb0, fa1, fb1, (b + 1) & 3, t + 4 < nkt ? my_slots : 0);
#endif
    b = (b + 1) & 3;
    top(nkt - 3 - t);
    stage_begin(t + 5, b);
#ifdef PP_QUAD_ABL_NODMA
    iter(fa1, fb1, fa0, fb0, (b + 1) & 3, 0);
#else
    iter(fa1, fb1, fa0, fb0, (b + 1) & 3, t + 5 < nkt ? my_slots : 0);
#endif
    b = (b + 1) & 3;
    // hipcc does not know the asm statements are MFMAs: behind the loop it reads accumulators (v_accvgpr_read / _mov)
    // right after the MFMA that writes them, a hazard the hardware does not interlock (seen: the first register of the
    // last MFMA tile of a wave read before the last K-tile had been added).  12 wait states cover an 8-pass MFMA; they
    // sit at the end of the loop body because register copies may be placed anywhere behind the loop.
    asm volatile("s_nop 11" ::: "memory");
  }
#ifdef PP_GEMM_TIMELINE
  rt_loop1 = __builtin_amdgcn_s_memrealtime();
  ct_loop1 = __builtin_amdgcn_s_memtime();
#endif
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();     // every wave is past its last fragment read: the ring is dead

  // ---- epilogue: bias + activation, the bf16 tile through LDS (row stride CS), then whole rows in 16-byte chunks.
  // Lane (frow, fq) owns row 16 i + frow and the 4 consecutive columns 16 j + 4 fq .. + 3 of every MFMA tile.
  const float *lbias = reinterpret_cast<const float *>(smem + BIAS_OFF);
  const bool has_bias = (epi & PP_EPI_BIAS) != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_bias) b4 = *reinterpret_cast<const float4 *>(lbias + wn * 16 * TN + j * 16 + fq * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
      if constexpr (ACT == 1) gelu4<bf16_t>(v);
      if constexpr (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      uint2 pk;
      pk.x = pack_bf16x2(v[0], v[1]);
      pk.y = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2 *>(smem + (wm * 16 * TM + i * 16 + frow) * CS + (wn * 16 * TN + j * 16 + fq * 4) * 2) = pk;
    }
  }
  __syncthreads();
  const int ncols16 = max(0, min(CPR, (p.N - n0) / 8));
  const bool headmajor = (epi & PP_EPI_HEADMAJOR) != 0;
  for (int c = tid; c < BM * CPR; c += 256) {
    const int lr = c / CPR, cc = c - lr * CPR;
    const int r = m0 + lr;
    if (r >= p.M || cc >= ncols16) continue;
    const uint4 v = *reinterpret_cast<const uint4 *>(smem + lr * CS + cc * 16);
    size_t off;
    if (headmajor) {   // [3][heads][M][head_dim]: a 16-byte chunk never straddles a head (head_dim % 8 == 0)
      const int n = n0 + cc * 8, hd = p.hm_HW, hh = n / hd, d = n - hh * hd;   // hh = which * heads + head
      off = (((size_t)hh * p.M + r) * hd + d) * 2;
    } else {
      off = ((size_t)r * p.ldc + n0 + cc * 8) * 2;
    }
    *reinterpret_cast<uint4 *>(p.C + off) = v;
  }
#ifdef PP_GEMM_TIMELINE
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = rt_entry; o[1] = rt_loop0; o[2] = rt_loop1; o[3] = rt_end;
    o[4] = __builtin_amdgcn_s_getreg(63492);
    o[5] = __builtin_amdgcn_s_getreg(63508);
    o[6] = ct_loop1 - ct_loop0 + 1; o[7] = (unsigned long long)tm << 32 | (unsigned)tn;
  }
#endif
}

#endif  // PP_GEMM_LAB

// ---------------------------------------------------------------------------------------------------------
// STREAM form of the quad kernel (tiles 18 / 19): one workgroup per CU walks its tiles as ONE stream of K-tiles.
//
// The per-launch form above spends a third of a workgroup's life outside its K-loop (timeline, ViT-B qkv, 192 x 288
// tiles: 2.9 us fill + 17.4 us K-loop + 5.3 us epilogue + 0.3 us until the next workgroup starts), and with one wave
// per SIMD nothing runs beside it.  Here the ring never drains: the last four K-tile slots of a tile already request
// the next tile's first four K-tiles, and the finished tile leaves without LDS, without a workgroup barrier and without
// waiting for memory: at the end of its K-loop every wave turns its accumulators into packed bf16 16-byte chunks held in
// registers (bias from the wave's own LDS slot, activation, a lane-pair exchange), and those leave from the NEXT tile's
// first 12 K-tile iterations, two or three stores per iteration between the MFMAs.
//
// vmcnt is ONE in-order counter for LDS-DMA pieces and stores, so the counted wait of iteration t is
//   2 * PMAX + (the stores and the one bias piece that iterations t - 2 and t - 1 issued),
// a compile-time constant: the first 14 iterations of a tile are unrolled (their stores name registers), the rest is a
// loop with no branch but its back edge.  Every wave issues exactly the same VMEM operations per iteration; before a
// workgroup's first tile the chunk stores carry an out-of-range offset, which the buffer descriptor drops.
// Requires M % BM == 0, N % BN == 0 (per-lane staging offsets are tile-independent), K >= 512, C below 4 GiB.
// ---------------------------------------------------------------------------------------------------------
typedef int i32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_bf16_zero(f32x4 &c, const u32x4 &w, const u32x4 &a) {   // first K-tile of a tile
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(a));
}
// 16 bytes per lane to C at byte offset voff (raw buffer descriptor: offsets past the end are dropped).  From asm: a store
// hipcc can see makes it guard the K-loop with vmcnt(0).
__device__ __forceinline__ void store16(const u32x4 &v, unsigned voff, const i32x4s &srd) {
  // the nops stand in for the hazard handling hipcc gives its own stores (a store of more than 8 bytes reads its data
  // registers late: the next instruction must not overwrite them)
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(srd) : "memory");
}
// s_waitcnt vmcnt(sel ? N1 : N0), then lgkmcnt(0) and the barrier as builtins (see wait_tiles_barrier)
template <int N1, int N0>
__device__ __forceinline__ void wait_sel_barrier(int sel) {
  static_assert(N1 <= 63 && N0 <= 63, "vmcnt is a 6-bit counter");
  if constexpr (N1 == N0) {
    wait_vmcnt<N1>();
  } else {
    asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc0 .Lw0_%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lwd_%=\n"
                 ".Lw0_%=:\n\ts_waitcnt vmcnt(%2)\n"
                 ".Lwd_%=:"
                 :
                 : "s"(__builtin_amdgcn_readfirstlane(sel)), "n"(N1), "n"(N0)
                 : "memory", "scc");
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
}
// 8 bytes per lane (the odd last column tile of a 16-row block)
__device__ __forceinline__ void store8(const u32x2 &v, unsigned voff, const i32x4s &srd) {
  asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(srd) : "memory");
}

constexpr int quad_stream_lds_bytes(int TM, int TN) {
  return 4 * (32 * TM + 32 * TN) * 64 + 2 * 4096;
}

template <int TM, int TN, int ACT>   // ACT: 0 none, 1 GELU, 2 ReLU
__global__ __launch_bounds__(256, 1) void gemm_quad_stream_kernel(GemmParams p, int vblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PP_GEMM_TIMELINE
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long rt_loop0 = 0, rt_loop1 = 0, ct_loop0 = 0, ct_loop1 = 0, ct_conv = 0;
#endif
  constexpr int BM = 32 * TM, BN = 32 * TN, STAGES = 4, BK = 32, RB = 64;
  constexpr int PTA = BM / 16, PT = (BM + BN) / 16, PMAX = (PT + 3) / 4, PREM = PT % 4, KA = PTA / 4;
  constexpr int A_BYTES = BM * RB, STAGE_BYTES = (BM + BN) * RB, RING = STAGES * STAGE_BYTES;
  constexpr int BIAS_OFF = RING;                        // two tile parities x four waves x 1 KiB
  constexpr int NPAIR = TN / 2, NIT = NPAIR + (TN & 1);  // store instructions per 16-row block: column-tile pairs (+ an odd one)
  constexpr int ST = TM * NIT;                          // ... per tile and wave
  constexpr int NSI = 12;                               // the K-tiles of the NEXT tile that carry them (2 - 3 each)
  constexpr int SLOT0 = 8 - PMAX;                       // pieces sit in slots SLOT0 .. 7
  constexpr int P2 = 2 * PMAX;
  // PERM (the 192 x 288 form): a wave's 144 columns start at byte 0 / 32 / 64 / 96 of a 128-byte line depending on the
  // tile's parity and the wave column, so its 128-byte store segments would straddle lines in three cases of four.  The
  // 18 column tiles (16 columns each) of the tile are therefore dealt out by tile parity: wave column wn takes the eight
  // tiles from the first line-aligned one -- tiles 8 wn .. 8 wn + 7 of an even tile, 2 + 8 wn .. 9 + 8 wn of an odd one
  // (n0 * 2 bytes = 576 tn is 64 past a line there) -- as its column tiles 0 .. 7, and one of the two left-over tiles
  // (16 + wn, or wn) as its tile 8.  Only the W rows a DMA piece fetches, the bias slice and the store offsets know.
  // The 256 x 192 form (TN = 6: 192-byte wave segments, tiles always line-aligned) has the same problem in wave column 1
  // only: it takes the tile's last four column tiles as its tiles 0 - 3 and tiles 6, 7 as its 4, 5.
  constexpr bool PERM = TN == 9 || TN == 6;
  static_assert(PTA % 4 == 0 && PMAX <= 8 && PMAX - 2 >= KA, "piece layout");
  static_assert(P2 + 2 * ((ST + NSI - 1) / NSI) + 1 <= 63, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int epi = p.epilogue;
  const int nkt = p.Kd / BK;                            // even, >= 16 (host)
  const int G = (int)gridDim.x;

  // virtual block id -> tile: the per-launch kernel's XCD-aware order (G is a multiple of 8: a workgroup stays on its XCD)
  auto decode = [&](int vb, int &tm, int &tn) __attribute__((always_inline)) -> bool {
    if (p.blocked) {
      constexpr int RM = 8;
      const int RN = p.rn, RT = RM * RN;
      const int nbm = (p.tiles_m + RM - 1) / RM, nbn = (p.tiles_n + RN - 1) / RN;
      const int x = vb & 7, j = vb >> 3;
      const int g = (j / RT) * 8 + x, idx = j % RT;
      if (g >= nbm * nbn) return false;
      const int bmi = g / nbn, bni = g - bmi * nbn;
      tm = bmi * RM + idx / RN;
      tn = bni * RN + idx % RN;
      return tm < p.tiles_m && tn < p.tiles_n;
    }
    tm = vb / p.tiles_n;
    tn = vb - tm * p.tiles_n;
    return vb < p.tiles_m * p.tiles_n;
  };
  auto next_valid = [&](int vb) __attribute__((always_inline)) -> int {   // first valid id >= vb on this workgroup's stride, or -1
    int tm, tn;
    for (; vb < vblocks; vb += G)
      if (decode(vb, tm, tn)) return vb;
    return -1;
  };
  int vb = next_valid((int)blockIdx.x);
  if (vb < 0) return;

  // ---- staging geometry (see gemm_quad_kernel); the per-lane offsets hold for every tile (M % BM == N % BN == 0)
  const int prow = lane >> 2, pchunk = lane & 3;
  const int perm_p = (0x1230 >> (4 * ((prow >> 2) & 3))) & 3;
  const int lchunk_off = (pchunk ^ perm_p) * 16;
  const int has_last = (PREM == 0 || wave < PREM) ? 1 : 0;
  const int last_k = has_last ? PMAX - 1 : PMAX - 2;      // a wave without a piece PMAX - 1 re-issues piece PMAX - 2 in its place
  auto col_tile = [](int parity, int wn_, int j) constexpr {       // PERM: global column tile of wave column wn_'s tile j
    if (TN == 6) return wn_ == 0 ? j : (j < 4 ? 8 + j : 2 + j);
    return j < 8 ? (parity ? 2 : 0) + 8 * wn_ + j : (parity ? wn_ : 16 + wn_);
  };
  unsigned off[PMAX], off_odd[PERM ? PMAX : 1];             // off_odd: the W pieces of an odd tile (PERM)
#pragma unroll
  for (int k = 0; k < PMAX; ++k) {
    const int kk = (k == PMAX - 1) ? last_k : k;          // wave-uniform
    const int q = wave + 4 * kk;
    if constexpr (PERM) {
      const int lp = q - PTA;                             // LDS column tile of a W piece: wave column lp / 9, its tile lp % 9
      const int ge = col_tile(0, lp / TN, lp % TN), go = col_tile(1, lp / TN, lp % TN);
      off[k] = (unsigned)((k < KA ? (q * 16 + prow) * p.lda : (ge * 16 + prow) * p.ldw) * 2 + lchunk_off);
      off_odd[k] = (unsigned)((k < KA ? (q * 16 + prow) * p.lda : (go * 16 + prow) * p.ldw) * 2 + lchunk_off);
    } else {
      off[k] = (unsigned)((k < KA ? (q * 16 + prow) * p.lda : ((q - PTA) * 16 + prow) * p.ldw) * 2 + lchunk_off);
    }
  }
  int st_par = 0;                                           // parity of the tile being staged (PERM)
  const unsigned lds0 = lds_offset_of(smem);
  unsigned st_lds = 0;
  unsigned long long st_A = 0, st_W = 0;
  auto piece = [&](auto kc) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    unsigned o = off[k];
    if constexpr (PERM && k >= KA) o = st_par ? off_odd[k] : off[k];
    return QuadPiece{o, k < KA ? st_A : st_W, st_lds + (k == PMAX - 1 ? last_k : k) * 4096};
  };
  auto stage_slot = [&](auto bc) __attribute__((always_inline)) {   // slot B carries piece B - SLOT0 of every wave
    constexpr int B = decltype(bc)::value;
    if constexpr (B >= SLOT0) glds_piece(piece(std::integral_constant<int, B - SLOT0>{}));
  };
  auto tile_base = [&](int tm, int tn, unsigned long long &bA, unsigned long long &bW) __attribute__((always_inline)) {
    bA = uniform64(p.A + (size_t)tm * BM * p.lda * 2);
    bW = uniform64(p.W + (size_t)tn * BN * p.ldw * 2);
  };
  // the bias slice of a tile for THIS wave: columns n0 + wn * 16 TN .. + 255, clamped to the array (one 1-KiB piece)
  const char *bias_src = (epi & PP_EPI_BIAS) ? (const char *)p.bias : p.W;      // no bias: any readable bytes, never used
  const int bias_cols = (epi & PP_EPI_BIAS) ? p.N : 4;
  auto bias_piece = [&](int tn, int par) __attribute__((always_inline)) {
    int c0 = tn * BN + wn * 16 * TN;
    int c = min(c0 + lane * 4, bias_cols - 4);
    if constexpr (PERM) {       // float 16 j + e of the slot: column e of the wave's column tile j
      const int lt = min(lane >> 2, TN - 1);
      int g = col_tile(tn & 1, wn, 0) + lt;                       // tiles 0 .. 7 (TN 9) / 0 .. 3 (TN 6) are consecutive
      if (TN == 9 && lt == 8) g = col_tile(tn & 1, wn, 8);
      if (TN == 6 && lt >= 4) g = col_tile(0, wn, 4) + lt - 4;
      c0 = tn * BN + g * 16 + (lane & 3) * 4;
      c = min(c0, bias_cols - 4);
    }
    return QuadPiece{(unsigned)(c * 4), uniform64(bias_src),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + BIAS_OFF + par * 4096 + wave * 1024))};
  };

  // ---- C: raw buffer descriptor (stride 0, byte-granular range check)
  const unsigned long long c_base = (unsigned long long)p.C;
  const i32x4s c_srd = {(int)(unsigned)c_base, (int)(unsigned)((c_base >> 32) & 0xFFFFu),
                        (int)(unsigned)((size_t)p.M * p.ldc * 2), 0x00020000};

  f32x4 acc[TM][TN];
  const int perm_f = (0x1230 >> (4 * ((frow >> 2) & 3))) & 3;
  const unsigned offA = (wm * 16 * TM + frow) * RB + ((fq ^ perm_f) << 4);
  const unsigned offB = A_BYTES + (wn * 16 * TN + frow) * RB + ((fq ^ perm_f) << 4);

  // One K-tile of the stream (straight-line code).  ZC: first K-tile of a tile (the MFMAs start the accumulators from 0).
  // BP: the next tile's bias piece leaves first.
  // TI >= 0: chunk stores [TI ST / NSI, (TI + 1) ST / NSI) of the PREVIOUS tile leave between the MFMAs, all before slot 7.
  u32x4 pk16[TM][NPAIR > 0 ? NPAIR : 1];
  u32x2 pk8[TM];
  unsigned cp[NIT];           // byte offset of (row frow of block 0, the lane's columns in pair / odd tile jp) of the packed tile
  unsigned block_step = 0;    // bytes between 16-row blocks
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    pk8[i] = u32x2{0u, 0u};
#pragma unroll
    for (int jp = 0; jp < NPAIR; ++jp) pk16[i][jp] = u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int jp = 0; jp < NIT; ++jp) cp[jp] = 0xFFFFFFF0u;      // before the first tile: out of range, the descriptor drops them
  auto chunk_store = [&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value, i = q / NIT, jp = q % NIT;
    // saturating: an out-of-range offset stays out of range
    const unsigned o = __builtin_elementwise_add_sat(cp[jp], (unsigned)i * block_step);
    if constexpr (jp < NPAIR) store16(pk16[i][jp], o, c_srd);
    else store8(pk8[i], o, c_srd);
  };
  auto iter = [&](u32x4 (&ca)[TM], u32x4 (&cb)[TN], u32x4 (&na)[TM], u32x4 (&nb)[TN], int rbuf, auto zc_c, auto bp_c,
                  const QuadPiece &bpiece, auto ti_c) __attribute__((always_inline)) {
    constexpr bool ZC = decltype(zc_c)::value, BP = decltype(bp_c)::value;
    constexpr int TI = decltype(ti_c)::value;
    const char *sb = smem + rbuf * STAGE_BYTES;
    if constexpr (BP) glds_piece(bpiece);
    [&]<int... MI>(std::integer_sequence<int, MI...>) {
      ([&] {
        constexpr int m = MI, i = m / TN, j = m % TN;
        if constexpr (j == 0) {
          na[i] = *reinterpret_cast<const u32x4 *>(sb + offA + i * 16 * RB);
          [&]<int... J>(std::integer_sequence<int, J...>) {
            ([&] {
              if constexpr ((J * TM) / TN == i) nb[J] = *reinterpret_cast<const u32x4 *>(sb + offB + J * 16 * RB);
            }(), ...);
          }(std::make_integer_sequence<int, TN>{});
        }
        [&]<int... B>(std::integer_sequence<int, B...>) {
          ([&] {
            if constexpr ((B * TM * TN) / 8 == m) stage_slot(std::integral_constant<int, B>{});
          }(), ...);
        }(std::make_integer_sequence<int, 8>{});
        if constexpr (TI >= 0 && TI < NSI) {
          constexpr int Q0 = (TI * ST) / NSI, Q1 = ((TI + 1) * ST) / NSI, NS = Q1 - Q0;
          constexpr int LIMIT = (7 * TM * TN) / 8, STEP = NS > 0 ? (LIMIT - 2) / NS : 1;
          if constexpr (NS > 0 && m >= 2 && (m - 2) % STEP == 0 && (m - 2) / STEP < NS)
            chunk_store(std::integral_constant<int, Q0 + (m - 2) / STEP>{});
        }
        if constexpr (ZC) mfma_bf16_zero(acc[i][j], cb[j], ca[i]);
        else mfma_bf16(acc[i][j], cb[j], ca[i]);
      }(), ...);
    }(std::make_integer_sequence<int, TM * TN>{});
  };
  constexpr auto F = std::false_type{};
  constexpr auto T = std::true_type{};

  // ---- first tile: its bias piece (oldest in the queue), K-tiles 0 .. 3, the fragments of K-tile 0
  int tm, tn;
  decode(vb, tm, tn);
  unsigned long long curA, curW, nxtA = 0, nxtW = 0;
  tile_base(tm, tn, curA, curW);
  int par = 0;
  glds_piece(bias_piece(tn, par));
  st_par = tn & 1;
#pragma unroll
  for (int s_ = 0; s_ < STAGES; ++s_) {
    st_lds = __builtin_amdgcn_readfirstlane(lds0 + s_ * STAGE_BYTES + wave * 1024);
    st_A = curA + (unsigned)(s_ * RB);
    st_W = curW + (unsigned)(s_ * RB);
    [&]<int... B>(std::integer_sequence<int, B...>) {
      (stage_slot(std::integral_constant<int, B>{}), ...);
    }(std::make_integer_sequence<int, 8>{});
  }
  u32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  wait_vmcnt<3 * PMAX>();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < TN; ++j) fb0[j] = *reinterpret_cast<const u32x4 *>(smem + offB + j * 16 * RB);
#pragma unroll
  for (int i = 0; i < TM; ++i) fa0[i] = *reinterpret_cast<const u32x4 *>(smem + offA + i * 16 * RB);
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
  ct_loop0 = __builtin_amdgcn_s_memtime();
#endif

  int b = 0;                  // ring buffer of the stream's current K-tile
  // VMEM operations of iteration t of a tile besides its PMAX pieces: the previous tile's chunk stores, the bias piece
  auto extras = [](int t) constexpr {
    if (t < 0) return 0;
    return (t < NSI ? ((t + 1) * ST) / NSI - (t * ST) / NSI : 0) + (t == 1 ? 1 : 0);
  };
  for (;;) {
    const int m0 = tm * BM, n0 = tn * BN;
    // the tile after this one (its first four K-tiles are requested by this tile's last four iterations)
    const int nvb = next_valid(vb + G);
    const int has_next = nvb >= 0 ? 1 : 0;
    int ntm = tm, ntn = tn;
    if (has_next) decode(nvb, ntm, ntn);
    tile_base(ntm, ntn, nxtA, nxtW);
    const QuadPiece nbias = bias_piece(ntn, par ^ 1);
    // K-tile t of this tile requests K-tile t + 4: of this tile, or K-tile t + 4 - nkt of the next one.  No predicate: at
    // the end of the stream (no next tile: nxtA / nxtW are this tile's) the last four K-tiles re-request K-tiles 0 .. 3
    // into ring buffers nobody reads any more; the kernel waits for them before it ends.  (A skipped piece is a taken
    // branch, and taken branches are what a one-wave-per-SIMD loop cannot afford: see DESIGN 4.1.)
    auto stage_begin = [&](int t) __attribute__((always_inline)) {
      const int kt = t + 4, wrap = kt >= nkt ? 1 : 0;
      const int k2 = wrap ? kt - nkt : kt;
      st_lds = __builtin_amdgcn_readfirstlane(lds0 + b * STAGE_BYTES + wave * 1024);
      st_A = (wrap ? nxtA : curA) + (unsigned)(k2 * RB);
      st_W = (wrap ? nxtW : curW) + (unsigned)(k2 * RB);
      st_par = (wrap ? ntn : tn) & 1;
    };
    // -- iterations 0 .. NSI + 1, unrolled: each carries its share of the previous tile's chunk stores (compile-time
    // registers) and waits for 2 PMAX + what the two iterations before it issued besides their pieces
    [&]<int... TI>(std::integer_sequence<int, TI...>) {
      ([&] {
        wait_sel_barrier<P2 + extras(TI - 2) + extras(TI - 1), P2 + extras(TI - 2) + extras(TI - 1)>(1);
        stage_begin(TI);
        if constexpr ((TI & 1) == 0)
          iter(fa0, fb0, fa1, fb1, (b + 1) & 3, std::bool_constant<TI == 0>{}, std::bool_constant<TI == 1>{}, nbias,
               std::integral_constant<int, TI>{});
        else
          iter(fa1, fb1, fa0, fb0, (b + 1) & 3, std::bool_constant<TI == 0>{}, std::bool_constant<TI == 1>{}, nbias,
               std::integral_constant<int, TI>{});
        b = (b + 1) & 3;
      }(), ...);
    }(std::make_integer_sequence<int, NSI + 2>{});
    // -- steady loop: no branch but its own back edge
    for (int t = NSI + 2; t < nkt; t += 2) {
      wait_sel_barrier<P2, P2>(1);
      stage_begin(t);
      iter(fa0, fb0, fa1, fb1, (b + 1) & 3, F, F, nbias, std::integral_constant<int, -1>{});
      b = (b + 1) & 3;
      wait_sel_barrier<P2, P2>(1);
      stage_begin(t + 1);
      iter(fa1, fb1, fa0, fb0, (b + 1) & 3, F, F, nbias, std::integral_constant<int, -1>{});
      b = (b + 1) & 3;
      asm volatile("s_nop 11" ::: "memory");      // MFMA -> v_accvgpr_read hazard behind the loop (see gemm_quad_kernel)
    }
#ifdef PP_GEMM_TIMELINE
    const unsigned long long cc0 = __builtin_amdgcn_s_memtime();
#endif
    // -- the finished tile, per wave, no LDS, no barrier: accumulators -> packed 16-byte chunks in registers; they leave from
    // the next tile's first NSI K-tiles, two or three per K-tile (3 % fewer cycles per workgroup than storing them here;
    // a real store takes ~236 cycles out of its wave wherever it is issued: DESIGN 4.1).
    // A lane (frow, fq) owns 4 consecutive columns of row frow in every 16 x 16 MFMA tile.  Two v_permlane16_swap_b32 per
    // pair of column tiles (j, j + 1) hand every lane 8 consecutive columns instead: even fq gets columns 4 fq .. 4 fq + 7 of
    // tile j (its own 4 + those of lane fq + 1), odd fq columns 4 (fq - 1) .. + 7 of tile j + 1: one 16-byte store per lane,
    // 64 contiguous bytes per row and store.
    {
      const float *lbias = reinterpret_cast<const float *>(smem + BIAS_OFF + par * 4096 + wave * 1024);
      const bool has_bias = (epi & PP_EPI_BIAS) != 0, headmajor = (epi & PP_EPI_HEADMAJOR) != 0;
      float4 b4[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // this wave's own piece, issued a whole tile ago and older than everything a later counted wait retired
        if (has_bias) b4[j] = *reinterpret_cast<const float4 *>(lbias + j * 16 + fq * 4);
      }
      const int ld_out = headmajor ? p.hm_HW : p.ldc;
      const unsigned row0 = (unsigned)(m0 + wm * 16 * TM + frow) * (unsigned)ld_out * 2u;
      block_step = 16u * (unsigned)ld_out * 2u;
      auto col_off = [&](int n) -> unsigned {     // byte offset of column n inside a row (head-major: of its head's slab)
        if (!headmajor) return (unsigned)n * 2u;
        const int hh = n / p.hm_HW, d = n - hh * p.hm_HW;   // [3][heads][M][head_dim]: 8 columns never straddle a head
        return (unsigned)(((size_t)hh * p.M * p.hm_HW + d) * 2);
      };
      const int lane_col = (fq & 1) ? 16 + (fq - 1) * 4 : fq * 4;
      // ROWPAIR (TN a multiple of 4): what a store costs is the number of 128-byte lines it touches (~4.5 cycles per line
      // and CU, a half-written line like a whole one: tools/ubench/store_bw.hip), and 16 rows x 64 B is 16 lines per KiB.
      // Neighbouring rows (lanes frow, frow ^ 1) therefore trade chunks so that one store holds 128 contiguous bytes of 8
      // rows: for the column-pair pair (2 pp, 2 pp + 1), store X carries the even row -- pair 2 pp from the even lane, pair
      // 2 pp + 1 from the odd lane -- and store Y the odd row.  192 x 256 tiles: every such 128 bytes is one aligned line.
      constexpr bool ROWPAIR = TN % 4 == 0 || PERM;
      const int wcol0 = PERM ? col_tile(tn & 1, wn, 0) * 16 : wn * 16 * TN;      // the wave's first (grouped) column in the tile
      if constexpr (ROWPAIR && (NPAIR & 1))      // an unpaired last pair of column tiles (TN 6: tiles 4, 5): 64-byte segments
        cp[NPAIR - 1] = row0 + col_off(n0 + (PERM ? col_tile(tn & 1, wn, NPAIR * 2 - 2) * 16 : wn * 16 * TN + (NPAIR - 1) * 32) + lane_col);
      if constexpr (ROWPAIR) {
        const unsigned ld2 = (unsigned)ld_out * 2u;
        const unsigned row_even = row0 - (unsigned)(frow & 1) * ld2;
#pragma unroll
        for (int pp = 0; pp < NPAIR / 2; ++pp) {
          const unsigned col = col_off(n0 + wcol0 + (2 * pp + (frow & 1)) * 32 + lane_col);
          cp[2 * pp] = row_even + col;
          cp[2 * pp + 1] = row_even + ld2 + col;
        }
      } else {
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) cp[jp] = row0 + col_off(n0 + wn * 16 * TN + jp * 32 + lane_col);
      }
      if constexpr (TN & 1)
        cp[NPAIR] = row0 + col_off(n0 + (PERM ? col_tile(tn & 1, wn, 8) * 16 : wn * 16 * TN + (TN - 1) * 16) + fq * 4);
      auto blocks = [&](auto now_c) __attribute__((always_inline)) {
      constexpr bool NOW = decltype(now_c)::value;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // a whole 16-row block at once: 4 TN values per lane, the activation over all of them side by side
        float v[4 * TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          v[4 * j] = acc[i][j][0] + b4[j].x;
          v[4 * j + 1] = acc[i][j][1] + b4[j].y;
          v[4 * j + 2] = acc[i][j][2] + b4[j].z;
          v[4 * j + 3] = acc[i][j][3] + b4[j].w;
        }
        if constexpr (ACT == 1) gelu_fast_n<4 * TN>(v);
        if constexpr (ACT == 2) {
#pragma unroll
          for (int e = 0; e < 4 * TN; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        unsigned w[2 * TN];
#pragma unroll
        for (int e = 0; e < 2 * TN; ++e) w[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
#pragma unroll
        for (int jp = 0; jp < NPAIR; ++jp) {          // tiles 2 jp (x) and 2 jp + 1 (y): dwords w[4 jp .. 4 jp + 3]
          const auto r0 = __builtin_amdgcn_permlane16_swap(w[4 * jp], w[4 * jp + 2], false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(w[4 * jp + 1], w[4 * jp + 3], false, false);
          pk16[i][jp] = u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
        if constexpr (ROWPAIR) {
          const bool odd = (frow & 1) != 0;
#pragma unroll
          for (int pp = 0; pp < NPAIR / 2; ++pp) {
            const u32x4 P = pk16[i][2 * pp], Q = pk16[i][2 * pp + 1];
            u32x4 S;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned give = odd ? P[e] : Q[e];          // what lane frow ^ 1 needs from this one
              S[e] = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              pk16[i][2 * pp][e] = odd ? S[e] : P[e];            // store X: the even row
              pk16[i][2 * pp + 1][e] = odd ? Q[e] : S[e];        // store Y: the odd row
            }
          }
        }
        if constexpr (TN & 1) pk8[i] = u32x2{w[2 * TN - 2], w[2 * TN - 1]};
        if constexpr (NOW) {          // the stream's last tile: nothing left to hide its stores under, so block i's go out
          [&]<int... J>(std::integer_sequence<int, J...>) {   // while block i + 1 is converted
            ([&] {
              // the compile-time block index: this loop is unrolled, i is a constant in every copy
              if (i == 0) chunk_store(std::integral_constant<int, 0 * NIT + J>{});
              if constexpr (TM > 1) if (i == 1) chunk_store(std::integral_constant<int, 1 * NIT + J>{});
              if constexpr (TM > 2) if (i == 2) chunk_store(std::integral_constant<int, 2 * NIT + J>{});
              if constexpr (TM > 3) if (i == 3) chunk_store(std::integral_constant<int, 3 * NIT + J>{});
              if constexpr (TM > 4) if (i == 4) chunk_store(std::integral_constant<int, 4 * NIT + J>{});
              if constexpr (TM > 5) if (i == 5) chunk_store(std::integral_constant<int, 5 * NIT + J>{});
              if constexpr (TM > 6) if (i == 6) chunk_store(std::integral_constant<int, 6 * NIT + J>{});
              if constexpr (TM > 7) if (i == 7) chunk_store(std::integral_constant<int, 7 * NIT + J>{});
            }(), ...);
          }(std::make_integer_sequence<int, NIT>{});
        }
      }
      };
      if (has_next) blocks(std::false_type{});
      else blocks(std::true_type{});
    }
#ifdef PP_GEMM_TIMELINE
    ct_conv += __builtin_amdgcn_s_memtime() - cc0;
#endif
    if (!has_next) break;
    vb = nvb; tm = ntm; tn = ntn; curA = nxtA; curW = nxtW; par ^= 1;
  }
#ifdef PP_GEMM_TIMELINE
  rt_loop1 = __builtin_amdgcn_s_memrealtime();
  ct_loop1 = __builtin_amdgcn_s_memtime();
#endif
  // ---- the stream's last (unused) K-tile requests must land before the LDS is freed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PP_GEMM_TIMELINE
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = rt_entry; o[1] = rt_loop0; o[2] = rt_loop1; o[3] = rt_end;
    o[4] = __builtin_amdgcn_s_getreg(63492);
    o[5] = __builtin_amdgcn_s_getreg(63508);
    o[6] = ct_loop1 - ct_loop0 + 1; o[7] = ct_conv;
  }
#endif
}

#ifdef PP_GEMM_LAB
template <int TM, int TN>
static int quad_launch_shape(const GemmParams &p, dim3 grid, hipStream_t s) {
  constexpr int lds = quad_lds_bytes(TM, TN);
  static thread_local unsigned long long attr_mask = 0;
  int dev_ = 0;
  if (attr_needed(attr_mask, dev_)) {
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_kernel<TM, TN, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_kernel<TM, TN, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_kernel<TM, TN, 2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  if (p.epilogue & PP_EPI_GELU) hipLaunchKernelGGL((gemm_quad_kernel<TM, TN, 1>), grid, dim3(256), lds, s, p);
  else if (p.epilogue & PP_EPI_RELU) hipLaunchKernelGGL((gemm_quad_kernel<TM, TN, 2>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((gemm_quad_kernel<TM, TN, 0>), grid, dim3(256), lds, s, p);
  PP_CHECK_LAUNCH("gemm_quad_kernel");
  return 0;
}

#endif

template <int TM, int TN>
static int quad_stream_launch_shape(const GemmParams &p, dim3 grid, hipStream_t s) {
  constexpr int lds = quad_stream_lds_bytes(TM, TN);
  static thread_local unsigned long long attr_mask = 0;
  static int ncu = 0;
  int dev_ = 0;
  if (attr_needed(attr_mask, dev_)) {
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_stream_kernel<TM, TN, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_stream_kernel<TM, TN, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_quad_stream_kernel<TM, TN, 2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  if (ncu == 0) {
    int dev = 0, n = 0;
    PP_CHECK_HIP(hipGetDevice(&dev));
    PP_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    ncu = n > 0 ? n : 256;
  }
  const int vblocks = (int)grid.x;
  int wgs = std::min(vblocks, ncu);
  if (wgs >= 8) wgs &= ~7;                      // a multiple of 8 keeps every workgroup's tiles on its XCD
  if (p.epilogue & PP_EPI_GELU) hipLaunchKernelGGL((gemm_quad_stream_kernel<TM, TN, 1>), dim3(wgs), dim3(256), lds, s, p, vblocks);
  else if (p.epilogue & PP_EPI_RELU) hipLaunchKernelGGL((gemm_quad_stream_kernel<TM, TN, 2>), dim3(wgs), dim3(256), lds, s, p, vblocks);
  else hipLaunchKernelGGL((gemm_quad_stream_kernel<TM, TN, 0>), dim3(wgs), dim3(256), lds, s, p, vblocks);
  PP_CHECK_LAUNCH("gemm_quad_stream_kernel");
  return 0;
}

// tiles 15 - 20 of pp_gemm (argument checks are the caller's: pp_gemm.hip)
int gemm_quad_launch(const GemmParams &p, int cfg, dim3 grid, hipStream_t s) {
  switch (cfg) {
#ifdef PP_GEMM_LAB
    case 15: return quad_launch_shape<8, 8>(p, grid, s);    // 256 x 256, per launch
    case 16: return quad_launch_shape<8, 6>(p, grid, s);    // 256 x 192
    case 17: return quad_launch_shape<6, 9>(p, grid, s);    // 192 x 288
#endif
    case 18: return quad_stream_launch_shape<8, 6>(p, grid, s);   // 256 x 192, stream
    case 19: return quad_stream_launch_shape<6, 9>(p, grid, s);   // 192 x 288, stream
    case 20: return quad_stream_launch_shape<6, 8>(p, grid, s);   // 192 x 256, stream
    default: return fail("pp_gemm: tile %d (the per-launch four-wave forms 15 - 17) exists in lab builds only", cfg);
  }
}

void gemm_quad_tile_shape(int cfg, int *bm, int *bn) {
  *bm = (cfg == 17 || cfg == 19 || cfg == 20) ? 192 : 256;
  *bn = (cfg == 15 || cfg == 20) ? 256 : ((cfg == 16 || cfg == 18) ? 192 : 288);
}

}  // namespace pp
