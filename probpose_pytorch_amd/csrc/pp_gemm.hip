// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,Kd] * W[N,Kd]^T), fp32 accumulate.
//
// One kernel family serves every dense contraction of the path (patch-embed,
// qkv, proj, fc1, fc2, 3x3 aux convolutions, the four stride-2 deconvolution
// parities, the final 1x1 heatmap layer): both operands are K-contiguous rows,
// A rows optionally addressed through a gather table (implicit GEMM, no im2col
// buffer), C rows optionally scattered through a row map.
//
// Structure (CDNA4): BM x BN output tile per workgroup of WGM x WGN waves, each wave TM x TN MFMA 16x16
// tiles.  Instantiations (pp_gemm's `tile` selector): 128x128, 192x96 (4 waves, 2 LDS stages, two workgroups
// per CU), 192x192 / 192x128 (8 waves, 3 stages, one workgroup per CU), 384x128, 192x384, 256x256, 192x256
// (8 waves, 2 stages).  192 rows = one crop's tokens, so at B = 64 the four ViT GEMMs are whole rounds.
// K-tiles of 128 B per row (64 bf16 / 32 fp32 / 128 fp8) are staged HBM/L2 -> LDS by global_load_lds_dwordx4
// (no VGPR round trip) into a 2- or 3-deep ring: counted s_waitcnt vmcnt(N) + one raw s_barrier per K-tile,
// the DMA pieces issued between the MFMA groups; the 16-B chunks are XOR-swizzled on the SOURCE address
// so the LDS image stays lane-linear for the DMA while ds_read_b128 fragment reads are bank-conflict free.
// bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity mode): v_mfma_f32_16x16x4_f32, bit-for-bit an fp32 FMA chain;
// fp8 (e4m3): v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, dequantised per column in the epilogue.
// Epilogues: bias / row bias / GELU / ReLU / in-place fp32 residual / NCHW heatmap store, the C tile staged
// through the dead K-loop buffers and stored as whole rows.  Forms: per-launch (gemm_kernel: plain, ping-pong order,
// wave-specialised producers), persistent stream (gemm_persist_kernel), two workgroups per CU (gemm_duo_kernel).
#include <type_traits>
#include <utility>

#include <algorithm>

#include "pp_common.h"
#include "pp_gemm_shared.h"

namespace pp {


constexpr int ROW_BYTES = 128;                 // bytes of K per staged row
constexpr int FUSE_MAPS = 32;                   // keypoint maps the fused final 1x1 layer serves (two 16-wide MFMA tiles)
constexpr int gemm_lds_bytes(int BM, int BN, int stages) {
  const int ring = stages * (BM + BN) * ROW_BYTES;
  // 192 x 256 (the N = 256 deconvolution layers): C staging + row map + the final layer's weight image (PP_EPI_FUSE_FINAL)
  const int fused = BM * (BN * 2 + 16) + BM * 4 + FUSE_MAPS * (BN * 2 + 16);
  return (BM == 192 && BN == 256 && fused > ring) ? fused : ring;
}

// Diagnostic build only (-DPP_GEMM_STAMPS, never shipped): per-wave cycle shares of the K-loop
// phases are written to the buffer passed in GemmParams::rowbias when epilogue bit 30 is set.
#ifdef PP_GEMM_STAMPS
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PP_STAMP(var) const unsigned long long var = stamp()
#define PP_ACC(dst, a, b) dst += (b) - (a)
#else
#define PP_STAMP(var)
#define PP_ACC(dst, a, b)
#endif


// Zero rows (W rows beyond N, zero-padding taps of the implicit convolutions) are DMA'd from a 64 KiB
// zero region, each lane/workgroup at a different 128-B line: one shared line would funnel every
// such request of the chip through a single L2 channel (measured: the K=17 heatmap layer, whose W
// tile is 82 % zero rows, ran 6x slower from a 256-B zero page).
constexpr int ZERO_REGION = 64 * 1024;
__device__ __attribute__((aligned(256))) unsigned char g_zero_page[ZERO_REGION];




// OCP e4m3 (gfx950): 4 floats -> 4 bytes, round-to-nearest-even, saturating at +-448 (NaN stays NaN)
typedef unsigned char fp8_t;
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f);
  b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
  c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f);
  d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}

// WGM x WGN waves per workgroup; STAGES LDS buffers (prefetch distance STAGES-1, counted vmcnt).
// GATHER: A rows addressed through the row-offset table (implicit convolution).  VEC: N and ldc are
// multiples of 4 and the epilogue is not the NCHW heatmap store (4-wide vector bias/residual/stores);
// the ragged variant (VEC = false) keeps the element-wise paths.  Both are compile-time so the hot
// plain-GEMM instantiation carries none of the gather / scalar code or its registers.
//
// NWP > 0 selects the wave-specialised form: WGM x WGN consumer waves that only read fragments and
// issue MFMAs, plus NWP producer waves that only issue the LDS-DMA pieces.  A global_load_lds
// blocks its wave for ~100-200 cycles while the CU's address unit drains (measured); with the DMA
// on the MFMA-issuing waves that stall came straight out of the matrix pipe's issue time.
//
// PINGPONG (8 waves, 3 stages): the two waves that share a SIMD (w and w + 4) run the K-tile interval in
// opposite order.  Waves 0-3: read the whole K-tile's fragments into registers, issue their DMA pieces, then
// TM*TN*2 bare MFMAs.  Waves 4-7: the MFMAs of the fragments read in the PREVIOUS interval first, then the
// fragment reads and DMA pieces for the next one.  One barrier per K-tile as before; between two barriers each
// SIMD always has one wave in its matrix segment and the other in its LDS / DMA segment (in the plain form both
// partners reach their DMA pieces together and the matrix pipe idles while the address unit drains).
template <typename T, int BM, int BN, int WGM, int WGN, int STAGES, bool GATHER, bool VEC, int NWP = 0,
          bool PINGPONG = false>
__global__ __launch_bounds__(64 * (WGM * WGN + NWP), (WGM * WGN + NWP + 3) / 4 < 2 ? 2 : (WGM * WGN + NWP + 3) / 4) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PP_GEMM_TIMELINE   // diagnostic build: wall-clock (100 MHz) marks per wave + where it ran
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long rt_loop0 = 0, rt_loop1 = 0, ct_loop0 = 0, ct_loop1 = 0;   // ct_*: shader cycles (s_memtime)
#endif
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = ROW_BYTES / ES;  // elements of K per tile
  constexpr int NW = WGM * WGN;                         // consumer (MFMA) waves
  constexpr int NWS = NWP > 0 ? NWP : NW;               // waves that issue the DMA
  constexpr int NTHREADS = 64 * (NW + NWP);
  constexpr int NTHREADS_EPI = 64 * NW;   // producer waves have exited by then
  constexpr int PA = BM / 8 / NWS, PB = BN / 8 / NWS;  // 1-KiB DMA pieces (8 rows) per staging wave per K-tile
  constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;  // 16x16 MFMA tiles per wave
  constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;
  static_assert(BM % (8 * NWS) == 0 && BN % (8 * NWS) == 0 && BM % (16 * WGM) == 0 && BN % (16 * WGN) == 0, "tile");

  // ---- tile assignment, XCD-aware (speed only, never correctness): workgroups b and b+8 share an
  // XCD and its 4 MiB L2.  The tile grid is cut into blocks of 8 x RN tiles (one block = one round
  // of an XCD's 32 CUs); block g goes to XCD g % 8 and blocks are numbered N-fastest, so an XCD
  // keeps working on the same few W column-slices (L2-resident) while A row-panels stream through,
  // each panel shared by the RN tiles of the block.  Out-of-range slots of partial blocks exit.
  constexpr int RM = 8;
  const int RN = p.rn, RT = RM * RN;
  int tm, tn;
  if (p.blocked) {
    const int nbm = (p.tiles_m + RM - 1) / RM, nbn = (p.tiles_n + RN - 1) / RN;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = (j / RT) * 8 + x, idx = j % RT;
    if (g >= nbm * nbn) return;
    const int bmi = g / nbn, bni = g - bmi * nbn;
    tm = bmi * RM + idx / RN;
    tn = bni * RN + idx % RN;
    if (tm >= p.tiles_m || tn >= p.tiles_n) return;
  } else {  // few tiles: one workgroup per tile in launch order, spread over all XCDs
    tm = blockIdx.x / p.tiles_n;
    tn = blockIdx.x - tm * p.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  // grid.y = batch entry x K split: split zs multiplies the K range [zs * Kd, (zs + 1) * Kd) into its own f32 partial
  const int zs = p.splitk > 1 ? (int)(blockIdx.y % p.splitk) : 0;
  const int z = p.splitk > 1 ? (int)(blockIdx.y / p.splitk) : (int)blockIdx.y;
  const char *Ab = p.A + ((size_t)z * p.strideA + (size_t)zs * p.strideA_k) * ES;
  const char *Wb = p.W + ((size_t)z * p.strideW + (size_t)zs * p.strideW_k) * ES;
  const int32_t *rowoff = GATHER ? p.rowoff + (size_t)z * p.strideRowoff + (size_t)zs * p.strideRowoff_k : nullptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_producer = NWP > 0 && __builtin_amdgcn_readfirstlane(wave) >= NW;
  const int cw = is_producer ? 0 : wave;               // consumer index (producers never use wm / wn)
  const int wm = cw / WGN, wn = cw - wm * WGN;
  const int sw = NWP > 0 ? (is_producer ? wave - NW : 0) : wave;  // staging-wave index

  // ---- staging geometry: each wave issues PA A-pieces + PB W-pieces (8 rows x 128 B) per K-tile.
  // lane -> (row in piece, physical 16-B chunk); logical chunk = physical ^ (row & 7).
  const int prow = lane >> 3, pchunk = lane & 7;
  // this lane's line inside the zero region: spread over workgroups, waves and piece rows
  const char *zero_line = (const char *)g_zero_page +
                          ((((blockIdx.x * 29 + blockIdx.y * 7 + sw) * 8 + prow) & 7) * 128 + pchunk * 16) +
                          (((blockIdx.x * 13 + sw * 5) & 7) * 8192);
  int a_row[PA], w_row[PB];
  const char *a_src[PA];
  const char *w_src[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (sw * PA + j) * 8 + prow;  // row inside the BM-row tile
    a_row[j] = min(m0 + r, p.M - 1);           // tail rows re-read the last row (never stored)
    a_src[j] = Ab + (size_t)a_row[j] * p.lda * ES + (pchunk ^ prow) * 16;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int r = (sw * PB + j) * 8 + prow;
    w_row[j] = n0 + r;
    w_src[j] = (w_row[j] < p.N) ? Wb + (size_t)w_row[j] * p.ldw * ES + (pchunk ^ prow) * 16
                                : zero_line + ((j * 5 + 3) & 7) * 1024;
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
  if constexpr (GATHER) {
#pragma unroll
    for (int j = 0; j < PA; ++j) cur_off[j] = rowoff[a_row[j]];
  }
  // One K-tile's DMA is split into its PA + PB pieces so the pieces can be issued BETWEEN MFMA
  // groups: a global_load_lds blocks the issuing wave for ~100-200 cycles while the CU's address
  // unit drains its queue (measured); issued back to back right after the barrier, all waves stall
  // there together and the matrix pipe idles for ~45 % of the K-loop.
  int st_kin = 0;
  unsigned st_ldsA = 0, st_ldsB = 0;
  size_t st_koff = 0;
  auto stage_begin = [&](int kt, int buf) {
    st_ldsA = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + sw * PA * 1024);
    st_ldsB = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + A_BYTES + sw * PB * 1024);
    st_koff = (size_t)kt * ROW_BYTES;
    if constexpr (GATHER) {
      const int k0 = kt * BK;
      const int seg = k0 / p.seg_len;
      st_kin = k0 - seg * p.seg_len;
      if (st_kin == 0 && kt > 0) {
#pragma unroll
        for (int j = 0; j < PA; ++j) cur_off[j] = nxt_off[j];
      }
    }
  };
  auto stage_piece = [&](auto qc) {   // qc: compile-time piece index 0 .. PA+PB-1
    constexpr int q = decltype(qc)::value;
    if constexpr (q < PA) {
      const char *src;
      if constexpr (GATHER)
        src = cur_off[q] >= 0 ? Ab + ((size_t)cur_off[q] + st_kin) * ES + lchunk_off
                              : zero_line + (q & 7) * 1024;
      else
        src = a_src[q] + st_koff;
      glds16(src, st_ldsA + q * 1024);
    } else {
      constexpr int j = q - PA;
      glds16(w_row[j] < p.N ? w_src[j] + st_koff : w_src[j], st_ldsB + j * 1024);
    }
  };
  auto stage_end = [&](int kt) {
    if constexpr (GATHER) {
      const int k0 = kt * BK;
      const int seg = k0 / p.seg_len;
      if (st_kin + BK == p.seg_len && k0 + BK < p.Kd) {
#pragma unroll
        for (int j = 0; j < PA; ++j) nxt_off[j] = rowoff[(size_t)(seg + 1) * p.M + a_row[j]];
      }
    }
  };
  auto stage_all = [&](int kt, int buf) {
    stage_begin(kt, buf);
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      (stage_piece(std::integral_constant<int, Q>{}), ...);
    }(std::make_integer_sequence<int, PA + PB>{});
    stage_end(kt);
  };

  f32x4 acc[TM][TN];

  // fragment read addresses: row = tile row (lane & 15), logical chunk = 4*s + (lane >> 4)
  const int frow = lane & 15, fq = lane >> 4;
  constexpr int PIECES = PA + PB;
  constexpr int GROUPS = 2 * TM;  // MFMA groups per K-tile (one A row-tile x TN column tiles each)

  // compute K-tile `buf`; when DO_STAGE, interleave the DMA pieces of the tile being prefetched
  auto compute = [&](int buf, auto do_stage_c) {
    constexpr bool DO_STAGE = decltype(do_stage_c)::value;
    const char *ldsA = smem + buf * STAGE_BYTES;
    const char *ldsB = ldsA + A_BYTES;
    if constexpr (ES == 1) {
      // fp8: one block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit scales 2^0) consumes the whole 128-deep
      // K-tile of a 16x16 output tile at twice the bf16 rate.  Lane (row, fq) supplies its two 16-B chunks
      // 4*0 + fq and 4*1 + fq = 32 k values; the k order inside the tile is the same permutation on both operands.
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      uint4 a8[TM][2], b8[TN][2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ra = wm * (BM / WGM) + i * 16 + frow;
          a8[i][hh] = *reinterpret_cast<const uint4 *>(ldsA + ra * ROW_BYTES + (((4 * hh + fq) ^ (ra & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int rb = wn * (BN / WGN) + j * 16 + frow;
          b8[j][hh] = *reinterpret_cast<const uint4 *>(ldsB + rb * ROW_BYTES + (((4 * hh + fq) ^ (rb & 7)) << 4));
        }
      }
      constexpr int UNIT = 0x7f7f7f7f;   // E8M0 exponent 127 = 2^0 in every byte
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
          constexpr int i = I;
          if constexpr (DO_STAGE) {
            [&]<int... Q>(std::integer_sequence<int, Q...>) {
              ([&] {
                if constexpr ((Q * TM) / PIECES == i) {
                  __builtin_amdgcn_sched_barrier(0);
                  stage_piece(std::integral_constant<int, Q>{});
                  __builtin_amdgcn_sched_barrier(0);
                }
              }(), ...);
            }(std::make_integer_sequence<int, PIECES>{});
          }
          const i32x8 av = {(int)a8[i][0].x, (int)a8[i][0].y, (int)a8[i][0].z, (int)a8[i][0].w,
                            (int)a8[i][1].x, (int)a8[i][1].y, (int)a8[i][1].z, (int)a8[i][1].w};
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const i32x8 bv = {(int)b8[j][0].x, (int)b8[j][0].y, (int)b8[j][0].z, (int)b8[j][0].w,
                              (int)b8[j][1].x, (int)b8[j][1].y, (int)b8[j][1].z, (int)b8[j][1].w};
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[i][j], 0, 0, 0, UNIT, 0, UNIT);
          }
        }(), ...);
      }(std::make_integer_sequence<int, TM>{});
      return;
    }
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int s = S;
        uint4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ra = wm * (BM / WGM) + i * 16 + frow;
          af[i] = *reinterpret_cast<const uint4 *>(ldsA + ra * ROW_BYTES + (((4 * s + fq) ^ (ra & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int rb = wn * (BN / WGN) + j * 16 + frow;
          bf[j] = *reinterpret_cast<const uint4 *>(ldsB + rb * ROW_BYTES + (((4 * s + fq) ^ (rb & 7)) << 4));
        }
        [&]<int... I>(std::integer_sequence<int, I...>) {
          ([&] {
            constexpr int i = I;
            constexpr int grp = s * TM + i;
            if constexpr (DO_STAGE) {
              // pieces whose slot floor(q * GROUPS / PIECES) equals this group
              [&]<int... Q>(std::integer_sequence<int, Q...>) {
                ([&] {
                  if constexpr ((Q * GROUPS) / PIECES == grp) {
                    __builtin_amdgcn_sched_barrier(0);
                    stage_piece(std::integral_constant<int, Q>{});
                    __builtin_amdgcn_sched_barrier(0);
                  }
                }(), ...);
              }(std::make_integer_sequence<int, PIECES>{});
            }
            if constexpr (sizeof(T) == 2) {
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    *reinterpret_cast<bf16x8 *>(&bf[j]), *reinterpret_cast<bf16x8 *>(&af[i]), acc[i][j], 0, 0, 0);
            } else if constexpr (sizeof(T) == 1) {
              // fp8 (e4m3): the 16-B chunk holds 16 consecutive k = two 8-byte MFMA operands; step h takes half h
              // of every lane's chunk (k slots 16 * (4s + fq) + 8h .. +7: the same permutation on both operands).
#pragma unroll
              for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(
                      reinterpret_cast<const long *>(&bf[j])[hh], reinterpret_cast<const long *>(&af[i])[hh],
                      acc[i][j], 0, 0, 0);
            } else {
              // fp32: the chunk holds 4 consecutive k; MFMA step e takes element e of every lane's
              // chunk (k slots 16s + 4*fq + e, the same permutation on both operands).
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                      __uint_as_float(reinterpret_cast<const unsigned *>(&bf[j])[e]),
                      __uint_as_float(reinterpret_cast<const unsigned *>(&af[i])[e]), acc[i][j], 0, 0, 0);
            }
          }(), ...);
        }(std::make_integer_sequence<int, TM>{});
      }(), ...);
    }(std::make_integer_sequence<int, 2>{});
  };

  // ---- ping-pong form: a whole K-tile of fragments in registers (2 x (TM + TN) x 16 B per lane)
  uint4 pfa[PINGPONG ? 2 : 1][PINGPONG ? TM : 1], pfb[PINGPONG ? 2 : 1][PINGPONG ? TN : 1];
  // per-lane fragment offsets inside a stage: row (w? * tile + frow), chunk (4s + fq) ^ (row & 7); the row-tile
  // index i / j only adds a multiple of 16 rows (an immediate offset of the ds_read, (row & 7) unchanged)
  unsigned pp_offA[2], pp_offB[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    pp_offA[s] = (wm * (BM / WGM) + frow) * ROW_BYTES + (((4 * s + fq) ^ (frow & 7)) << 4);
    pp_offB[s] = A_BYTES + (wn * (BN / WGN) + frow) * ROW_BYTES + (((4 * s + fq) ^ (frow & 7)) << 4);
  }
  auto pp_load = [&](int buf) {
    const char *sb = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char *pa = sb + pp_offA[s];
      const char *pb = sb + pp_offB[s];
#pragma unroll
      for (int i = 0; i < TM; ++i) pfa[s][i] = *reinterpret_cast<const uint4 *>(pa + i * 16 * ROW_BYTES);
#pragma unroll
      for (int j = 0; j < TN; ++j) pfb[s][j] = *reinterpret_cast<const uint4 *>(pb + j * 16 * ROW_BYTES);
    }
  };
  auto pp_mma = [&]() {
    if constexpr (ES == 1) {
      typedef int i32x8 __attribute__((ext_vector_type(8)));
      constexpr int UNIT = 0x7f7f7f7f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const i32x8 av = {(int)pfa[0][i].x, (int)pfa[0][i].y, (int)pfa[0][i].z, (int)pfa[0][i].w,
                          (int)pfa[1][i].x, (int)pfa[1][i].y, (int)pfa[1][i].z, (int)pfa[1][i].w};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const i32x8 bv = {(int)pfb[0][j].x, (int)pfb[0][j].y, (int)pfb[0][j].z, (int)pfb[0][j].w,
                            (int)pfb[1][j].x, (int)pfb[1][j].y, (int)pfb[1][j].z, (int)pfb[1][j].w};
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[i][j], 0, 0, 0, UNIT, 0, UNIT);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (ES == 2) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8 *>(&pfb[s][j]),
                                                                  *reinterpret_cast<bf16x8 *>(&pfa[s][i]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    __uint_as_float(reinterpret_cast<const unsigned *>(&pfb[s][j])[e]),
                    __uint_as_float(reinterpret_cast<const unsigned *>(&pfa[s][i])[e]), acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  // ---- pipeline fill first: the DMA of the first STAGES-1 K-tiles is in flight while the epilogue
  // operands below (residual / row bias / bias) are fetched.  Those loads are
  // YOUNGER than the fill, so the first counted wait of the K-loop over-waits (it also retires tile 1):
  // safe, and tile 1 was issued together with tile 0 anyway.
  if constexpr (NWP == 0) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nkt) stage_all(s, s);
  }

  // ---- wave-specialised form: the producer waves run their whole K-loop here and EXIT (s_barrier counts only the
  // surviving waves of a workgroup), so the accumulators, fragments and epilogue operands below are live in the
  // consumer waves only and the register allocation is not the union of both roles.
  if constexpr (NWP > 0) {
    if (is_producer) {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < nkt) stage_all(s, s);
      int buf = 0;
      for (int kt = 0; kt < nkt; ++kt) {
        if (kt + STAGES - 1 <= nkt) {
          wait_vmcnt<(STAGES - 2) * PIECES>();
        } else {
          wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < nkt) {
          int nb = buf + STAGES - 1;
          if (nb >= STAGES) nb -= STAGES;
          stage_all(kt + STAGES - 1, nb);
        }
        if (++buf == STAGES) buf = 0;
      }
      return;
    }
  }

  // ---- every additive epilogue term (bias, pos-embed row bias, the fp32 residual that the proj/fc2
  // GEMMs update in place) is loaded straight INTO the accumulators before the K-loop: the loads are
  // older than every DMA piece, their latency hides under the pipeline fill, and the epilogue needs no
  // global loads at all (the in-place residual aliases C, so hipcc could never hoist those loads).
  const int epi = p.epilogue;
  const float *__restrict__ bias = p.bias ? p.bias + (size_t)z * p.strideBias : nullptr;
  const int32_t *__restrict__ rowmap = p.out_rowmap ? p.out_rowmap + (size_t)z * p.strideRowmap : nullptr;
  const float *Rb = p.residual ? p.residual + (size_t)z * p.strideC : nullptr;
  constexpr bool vec_ok = VEC;
  int out_row[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WGM) + i * 16 + frow;
    out_row[i] = (rowmap && m < p.M) ? rowmap[m] : m;
  }
  // one row-shaped term goes straight into the accumulator registers (no extra live registers):
  // the residual (proj / fc2) or the pos-embed row bias (patch embed); the column bias is added
  // from TN float4 registers in the epilogue.
  const float *init_base =
      is_producer ? nullptr : ((epi & PP_EPI_RESIDUAL) ? Rb : ((epi & PP_EPI_ROWBIAS) ? p.rowbias : nullptr));
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WGM) + i * 16 + frow;
    const size_t rowbase = (epi & PP_EPI_RESIDUAL) ? (size_t)out_row[i] * p.ldc
                                                   : (size_t)(m % (p.rowbias_period > 0 ? p.rowbias_period : 1)) * p.ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WGN) + j * 16 + fq * 4;
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
      if (init_base && m < p.M) {
        if constexpr (vec_ok) {
          if (n < p.N) {
            const float4 t = *reinterpret_cast<const float4 *>(init_base + rowbase + n);
            c = f32x4{t.x, t.y, t.z, t.w};
            if constexpr (ES == 1) {   // the fp8 epilogue multiplies the accumulator by colscale[n]: pre-divide
              const float4 sc = *reinterpret_cast<const float4 *>(p.colsum + (size_t)z * p.strideBias + n);
              c = f32x4{t.x / sc.x, t.y / sc.y, t.z / sc.z, t.w / sc.w};
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) c[e] = init_base[rowbase + n + e];
        }
      }
      acc[i][j] = c;
    }
  }
  float4 bias4[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / WGN) + j * 16 + fq * 4;
    bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((epi & PP_EPI_BIAS) && !is_producer) {
      if constexpr (vec_ok) {
        if (n < p.N) bias4[j] = *reinterpret_cast<const float4 *>(bias + n);
      } else {
        if (n + 0 < p.N) bias4[j].x = bias[n + 0];
        if (n + 1 < p.N) bias4[j].y = bias[n + 1];
        if (n + 2 < p.N) bias4[j].z = bias[n + 2];
        if (n + 3 < p.N) bias4[j].w = bias[n + 3];
      }
    }
  }

  float4 cs4[TN];     // fp8: per-column dequantisation scales
#pragma unroll
  for (int j = 0; j < TN; ++j) cs4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (VEC && ES == 1) {   // fp8: per-column dequantisation scale (activation scale x weight-row scale)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WGN) + j * 16 + fq * 4;
      if (n < p.N) cs4[j] = *reinterpret_cast<const float4 *>(p.colsum + (size_t)z * p.strideBias + n);
    }
  }

  // ---- main loop.  STAGES-1 K-tiles of DMA stay in flight across the barrier: the wait that
  // retires tile t is a COUNTED vmcnt (never 0 in steady state), then one raw s_barrier makes every
  // wave's pieces of tile t visible and proves every wave finished reading the buffer that tile
  // t+STAGES-1 is about to overwrite (it was consumed in iteration t-1).
#ifdef PP_GEMM_STAMPS
  unsigned long long c_wait = 0, c_bar = 0, c_stage = 0, c_comp = 0;
#endif
  PP_STAMP(t_begin);
#ifdef PP_GEMM_STAMPS
  unsigned long long t_pro_v = t_begin;
#endif
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
  ct_loop0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  if constexpr (NWP > 0) {
    static_assert(!PINGPONG, "the ping-pong order is a form of the non-specialised kernel");
    // ---- wave-specialised K-loop: same barrier protocol, the two halves of each iteration on
    // different waves.  Every wave executes exactly nkt barriers.
    {
      int buf = 0;
      for (int kt = 0; kt < nkt; ++kt) {
        PP_STAMP(tb);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PP_STAMP(tc);
        compute(buf, std::false_type{});
        PP_STAMP(te);
        PP_ACC(c_bar, tb, tc);
        PP_ACC(c_comp, tc, te);
        if (++buf == STAGES) buf = 0;
      }
    }
  } else {
  // The epilogue operands above (residual / row bias into the accumulators, bias, row map) are ordinary global
  // loads whose first USE is inside the K-loop (the accumulators are MFMA operands).  hipcc does not see the
  // inline-asm DMA, so left alone it puts its own `s_waitcnt vmcnt(0)` in front of the loop's first MFMA - in
  // EVERY iteration (loop-carried scoreboard) - which drains the whole LDS-DMA prefetch ring each K-tile
  // (found in the round-1 kernel's ISA: the counted vmcnt(PIECES) was followed by a compiler vmcnt(0)).
  // Retire them once here with a wait the compiler models: from now on only DMA pieces are outstanding and
  // the loop keeps nothing but the counted waits.  Cost: the first iteration also waits for K-tile 1.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
#ifdef PP_GEMM_STAMPS
  t_pro_v = stamp();
#endif
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
  ct_loop0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  if constexpr (PINGPONG) {
    static_assert(STAGES == 3 && NW == 8, "ping-pong form: 8 waves, 3 LDS stages");
    // interval kt (between two barriers) reads K-tile kt and prefetches K-tile kt + 2 into the buffer that
    // interval kt - 1 read.  The two roles are whole separate loops (wave-uniform branch): one code path per
    // loop body, every wave passes exactly nkt barriers.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
    int buf = 0, kt = 0;
    auto advance = [&]() { if (++buf == STAGES) buf = 0; };
    auto nbuf = [&]() { int nb = buf + STAGES - 1; return nb >= STAGES ? nb - STAGES : nb; };
    if (!late) {
      for (; kt + 2 < nkt; ++kt) {
        PP_STAMP(ta);
        wait_vmcnt<PIECES>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(tb);
        pp_load(buf);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(tc);
        stage_all(kt + 2, nbuf());
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(td);
        pp_mma();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(te);
        PP_ACC(c_bar, ta, tb);
        PP_ACC(c_wait, tb, tc);
        PP_ACC(c_stage, tc, td);
        PP_ACC(c_comp, td, te);
        advance();
      }
      for (; kt < nkt; ++kt) {
        if (kt + 1 < nkt) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pp_load(buf);
        pp_mma();
        __builtin_amdgcn_sched_barrier(0);
        advance();
      }
    } else {
      // first interval: nothing to multiply yet
      if (nkt > 1) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      pp_load(buf);
      __builtin_amdgcn_sched_barrier(0);
      if (2 < nkt) stage_all(2, nbuf());
      advance();
      kt = 1;
      for (; kt + 2 < nkt; ++kt) {
        PP_STAMP(ta);
        wait_vmcnt<PIECES>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(tb);
        pp_mma();
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(tc);
        pp_load(buf);
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(td);
        stage_all(kt + 2, nbuf());
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(te);
        PP_ACC(c_bar, ta, tb);
        PP_ACC(c_comp, tb, tc);
        PP_ACC(c_wait, tc, td);
        PP_ACC(c_stage, td, te);
        advance();
      }
      for (; kt < nkt; ++kt) {
        if (kt + 1 < nkt) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        pp_mma();
        __builtin_amdgcn_sched_barrier(0);
        pp_load(buf);
        __builtin_amdgcn_sched_barrier(0);
        advance();
      }
      pp_mma();
    }
  } else {
  int buf = 0, kt = 0;
  // steady state: every iteration prefetches tile kt + STAGES - 1 (one code path in the loop body,
  // so the accumulators stay in place across iterations)
  for (; kt + STAGES - 1 < nkt; ++kt) {
    PP_STAMP(ta);
    wait_vmcnt<(STAGES - 2) * PIECES>();          // STAGES-2 younger tiles may still fly
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(tb);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(tc);
    int nb = buf + STAGES - 1;
    if (nb >= STAGES) nb -= STAGES;
    stage_begin(kt + STAGES - 1, nb);
    compute(buf, std::true_type{});
    stage_end(kt + STAGES - 1);
    PP_STAMP(te);
    PP_ACC(c_bar, tb, tc);
    PP_ACC(c_wait, ta, tb);
    PP_ACC(c_comp, tc, te);
    if (++buf == STAGES) buf = 0;
  }
  // tail: the last STAGES-1 tiles are already in flight, nothing left to prefetch
  for (; kt < nkt; ++kt) {
    PP_STAMP(ta);
    if (kt + STAGES - 1 <= nkt) {
      wait_vmcnt<(STAGES - 2) * PIECES>();
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(tb);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(tc);
    compute(buf, std::false_type{});
    PP_STAMP(te);
    PP_ACC(c_bar, tb, tc);
    PP_ACC(c_wait, ta, tb);
    PP_ACC(c_comp, tc, te);
    if (++buf == STAGES) buf = 0;
  }
  }
  }
  PP_STAMP(t_loop);
#ifdef PP_GEMM_TIMELINE
  rt_loop1 = __builtin_amdgcn_s_memrealtime();
  ct_loop1 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue.  The W fragment is the MFMA "A" operand and the activation fragment the "B"
  // operand, so a 16x16 accumulator tile holds C^T: lane (frow, fq) owns output row m = .. + frow and
  // the 4 CONSECUTIVE columns n = .. + 4*fq + e -> 8-byte (bf16) / 16-byte (fp32) vector stores.
  // All additive terms are already in the accumulators: activation, convert, store.
  char *Cb = p.C + ((size_t)z * p.strideC + (size_t)zs * p.strideC_k) *
                       ((epi & (PP_EPI_OUT_F32 | PP_EPI_HEATMAP)) ? 4 : (ES == 1 ? ((epi & PP_EPI_OUT_FP8) ? 1 : 2) : ES));
  bool stored = false;
  if constexpr (VEC) {
    // Stage the C tile through LDS (the K-loop buffers are dead) and store whole rows: 16 B per lane
    // and PBN*OES contiguous bytes per row instead of 8/16-B pieces of 16 different rows per
    // wave-instruction (the direct path's store tail cost ~24 % of a K = 768 tile).  Tiles wider than
    // the staging buffers go through in NPASS column passes.  OES = output element size.
    // ACT (0 none, 1 GELU, 2 ReLU) is a compile-time mode: with the flag tested per output quad the branches kept
    // the compiler from interleaving the exp / rcp chains of neighbouring quads (the GELU epilogue is issue-bound)
    auto lds_epilogue = [&]<int OES, int ACT>(std::integral_constant<int, OES>, std::integral_constant<int, ACT>) {
      constexpr int BUDGET = STAGES * STAGE_BYTES;
      constexpr auto fits = [](int np) {
        return BM * (BN / np * OES + 16) + BM * 4 <= BUDGET && WGN % np == 0;
      };
      constexpr int NPASS = fits(1) ? 1 : (fits(2) ? 2 : 4);
      static_assert(fits(NPASS), "C tile must fit the staging buffers");
      constexpr int PBN = BN / NPASS;                  // columns per pass
      constexpr int CS = PBN * OES + 16;               // padded row stride: keeps 16-B alignment, spreads banks
      int *rows_lds = reinterpret_cast<int *>(smem + BM * CS);
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        __syncthreads();                             // K-loop reads (pass 0) / previous pass's row stores done
        const bool mine = !is_producer && (wn / (WGN / NPASS)) == pass;
        const int wn_in = wn - pass * (WGN / NPASS);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (!mine) break;
          const int lr = wm * (BM / WGM) + i * 16 + frow;
          if (wn_in == 0 && fq == 0) rows_lds[lr] = (m0 + lr < p.M) ? out_row[i] : -1;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v[4];
            if constexpr (sizeof(T) == 1) {
              v[0] = fmaf(acc[i][j][0], cs4[j].x, bias4[j].x); v[1] = fmaf(acc[i][j][1], cs4[j].y, bias4[j].y);
              v[2] = fmaf(acc[i][j][2], cs4[j].z, bias4[j].z); v[3] = fmaf(acc[i][j][3], cs4[j].w, bias4[j].w);
            } else {
              v[0] = acc[i][j][0] + bias4[j].x; v[1] = acc[i][j][1] + bias4[j].y;
              v[2] = acc[i][j][2] + bias4[j].z; v[3] = acc[i][j][3] + bias4[j].w;
            }
            if constexpr (ACT == 1) gelu4<T>(v);
            if constexpr (ACT == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            char *dst = smem + lr * CS + (wn_in * (BN / WGN) + j * 16 + fq * 4) * OES;
            if constexpr (OES == 4) {
              *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else if constexpr (OES == 1) {
              const float q = p.out_scale;
              *reinterpret_cast<unsigned *>(dst) = pack_fp8x4(v[0] * q, v[1] * q, v[2] * q, v[3] * q);
            } else {
              uint2 pk;
              pk.x = pack_bf16x2(v[0], v[1]);
              pk.y = pack_bf16x2(v[2], v[3]);
              *reinterpret_cast<uint2 *>(dst) = pk;
            }
          }
        }
        if constexpr (BM == 192 && BN == 256 && OES == 2 && sizeof(T) == 2 && NPASS == 1 && NWP == 0) {
          if (epi & PP_EPI_FUSE_FINAL) {
            // ---- fused final 1x1 layer (head.py:525-532): the ReLU'd bf16 tile just staged (192 pixels x all 256
            // channels) is multiplied by the K <= 32 keypoint maps' weights right here -- a 192 x 32 x 256 product on
            // the matrix cores, A = pixels from the staging rows, B = the weight image loaded beside them -- then
            // bias, / temperature, clamp and the NCHW float32 store.  The 256-channel 64x48 map (100 MB at bs 64) is
            // never written to HBM nor read back, and the separate final-layer launch disappears.  Same k order and
            // same bf16-rounded operands as final_heatmap_mfma_kernel: bit-identical heatmaps.
            constexpr int WFS = BN * 2 + 16;
            char *wimg = smem + BM * CS + BM * 4;
            for (int c = tid; c < FUSE_MAPS * (BN / 8); c += NTHREADS_EPI) {
              const int k = c / (BN / 8), ch = c - k * (BN / 8);
              uint4 v = make_uint4(0, 0, 0, 0);
              if (k < p.hm_K) v = *reinterpret_cast<const uint4 *>(p.final_w + ((size_t)k * BN + ch * 8) * 2);
              *reinterpret_cast<uint4 *>(wimg + k * WFS + ch * 16) = v;
            }
            __syncthreads();
            const int prow_ = lane & 15, kq = lane >> 4;
            float *heat = reinterpret_cast<float *>(p.C);
            const int HW = p.hm_HW, KK = p.hm_K;
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
              const int t = wave * 3 + tt, rt = t >> 1, ct = t & 1;
              f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int s2 = 0; s2 < BN / 32; ++s2) {
                const uint4 av = *reinterpret_cast<const uint4 *>(smem + (rt * 16 + prow_) * CS + (4 * s2 + kq) * 16);
                const uint4 bv = *reinterpret_cast<const uint4 *>(wimg + (ct * 16 + prow_) * WFS + (4 * s2 + kq) * 16);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(&av),
                                                             *reinterpret_cast<const bf16x8 *>(&bv), a2, 0, 0, 0);
              }
              const int k = ct * 16 + prow_;            // accumulator: pixels 4 kq + e (rows) of map k (column)
              if (k < KK) {
                const float fbk = p.final_b[k];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int r = rows_lds[rt * 16 + 4 * kq + e];
                  if (r < 0) continue;
                  float x = (a2[e] + fbk) / p.hm_temperature;
                  if (!(epi & PP_EPI_NOCLAMP)) x = fminf(fmaxf(x, 0.f), 1.f);
                  const int b_ = r / HW, hw = r - b_ * HW;
                  heat[((size_t)b_ * KK + k) * HW + hw] = x;
                }
              }
            }
            continue;       // nothing else to store for this tile (NPASS == 1: leaves the pass loop)
          }
        }
        __syncthreads();
        constexpr int CPR = PBN * OES / 16;           // 16-B chunks per row
        const int ncols16 = max(0, min(CPR, (p.N - n0 - pass * PBN) * OES / 16));
        for (int c = tid; c < BM * CPR; c += NTHREADS_EPI) {
          const int lr = c / CPR, cc = c - lr * CPR;
          const int r = rows_lds[lr];
          if (r < 0 || cc >= ncols16) continue;
          const uint4 v = *reinterpret_cast<const uint4 *>(smem + lr * CS + cc * 16);
          if constexpr (OES == 2) {
            if (epi & PP_EPI_HEADMAJOR) {
              // qkv projection written head-major, [3][heads][M][head_dim]: one head's rows are contiguous for the
              // attention kernel (a 160-byte head row at a 7 680-byte stride touches 2 - 3 cache lines, ViT-H)
              const int n = n0 + pass * PBN + cc * 8, Cc = p.hm_K * p.hm_HW;
              const int which = n / Cc, rem = n - which * Cc, head = rem / p.hm_HW, d = rem - head * p.hm_HW;
              *reinterpret_cast<uint4 *>(Cb + ((((size_t)which * p.hm_K + head) * p.M + r) * p.hm_HW + d) * 2) = v;
              continue;
            }
          }
          *reinterpret_cast<uint4 *>(Cb + ((size_t)r * p.ldc + n0 + pass * PBN) * OES + cc * 16) = v;
        }
      }
    };
    if (p.lds_epilogue) {
      auto with_act = [&](auto oes) __attribute__((always_inline)) {
        if (epi & PP_EPI_GELU) lds_epilogue(oes, std::integral_constant<int, 1>{});
        else if (epi & PP_EPI_RELU) lds_epilogue(oes, std::integral_constant<int, 2>{});
        else lds_epilogue(oes, std::integral_constant<int, 0>{});
      };
      if (sizeof(T) == 4 || (epi & PP_EPI_OUT_F32))
        with_act(std::integral_constant<int, 4>{});
      else if (sizeof(T) == 1 && (epi & PP_EPI_OUT_FP8))
        with_act(std::integral_constant<int, 1>{});
      else
        with_act(std::integral_constant<int, 2>{});
      stored = true;
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / WGM) + i * 16 + frow;
    if (stored || is_producer || m >= p.M) continue;
    const int r = out_row[i];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WGN) + j * 16 + fq * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0] + bias4[j].x, acc[i][j][1] + bias4[j].y, acc[i][j][2] + bias4[j].z,
                    acc[i][j][3] + bias4[j].w};
      if (epi & PP_EPI_GELU) {
        gelu4<T>(v);
      }
      if (epi & PP_EPI_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (vec_ok) {
        const size_t idx = (size_t)r * p.ldc + n;
        if ((epi & PP_EPI_OUT_F32) || sizeof(T) == 4) {
          *reinterpret_cast<float4 *>(reinterpret_cast<float *>(Cb) + idx) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2 *>(reinterpret_cast<bf16_t *>(Cb) + idx) = pk;
        }
      } else {
      // scalar path: ragged N (e.g. the K=17 heatmap layer) or unaligned ldc
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ne = n + e;
        if (ne >= p.N) continue;
        float x = v[e];
        if (epi & PP_EPI_HEATMAP) {
          x = x / p.hm_temperature;
          if (!(epi & PP_EPI_NOCLAMP)) x = fminf(fmaxf(x, 0.f), 1.f);
          const int b = r / p.hm_HW, hw = r - b * p.hm_HW;
          reinterpret_cast<float *>(Cb)[((size_t)b * p.hm_K + ne) * p.hm_HW + hw] = x;
          continue;
        }
        const size_t idx = (size_t)r * p.ldc + ne;
        if (epi & PP_EPI_OUT_F32)
          reinterpret_cast<float *>(Cb)[idx] = x;
        else if constexpr (sizeof(T) != 1)   // fp8 launches always take the LDS epilogue (checked on the host)
          Store<T>::st(reinterpret_cast<T *>(Cb) + idx, x);
      }
      }
    }
  }
#ifdef PP_GEMM_TIMELINE
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile's stores have been accepted
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = rt_entry; o[1] = rt_loop0; o[2] = rt_loop1; o[3] = rt_end;
    o[4] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
    o[5] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    o[6] = ct_loop1 - ct_loop0 + 1;            // K-loop shader cycles (> 0: also the "this wave ran" mark)
    o[7] = (unsigned long long)tm << 32 | (unsigned)tn;
  }
#endif
#ifdef PP_GEMM_STAMPS
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    PP_STAMP(t_end);
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * NW + wave) * 8;
    o[0] = t_pro_v - t_begin; o[1] = c_wait; o[2] = c_bar; o[3] = c_stage; o[4] = c_comp;  // stage now inside compute
    o[5] = t_end - t_loop; o[6] = t_end - t_begin; o[7] = t_begin;
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------
// Persistent form of the 192x192 / 8-wave / 3-stage configuration for the plain bf16 -> bf16 layers (qkv, fc1).
//
// Measured on the per-launch form (tools/gemm_timeline.py, ViT-B bs 64, round 3): a workgroup of qkv lives 17.6 us, of
// which the K-loop is 12.0; 2.9 us go to the pipeline fill -- every CU of the chip asks for its first two K-tiles in
// the same microsecond, a cold burst of ~25 MB -- 2.7 us to the epilogue (4.7 with GELU), and 0.4 us pass between one
// workgroup's end and the next one's start on a CU.  With 3 (qkv) or 4 (fc1) tiles per CU the fill + gap are paid 3 - 4
// times per launch.  Here ONE workgroup per CU walks its tiles as ONE continuous stream of K-tiles through the LDS
// ring: the DMA of the next tile's first two K-tiles is issued during the current tile's last two iterations, so they
// land underneath the epilogue, which runs right away out of the accumulators: bias / activation / bf16 -> the ring
// buffer the last K-tile was read from (the other two hold the next tile's K-tiles) in two column passes -> whole-row
// 16-byte stores.  (Round 2 built this stream with the epilogue DEFERRED into the next tile's K-loop, 9 slices of
// VALU + 8-byte stores between the MFMA groups: it tied the per-launch form, the K-loop is issue-bound and took the
// slices at full price.  Here the K-loop is exactly the per-launch form's.)
// No compiler-visible global load or store exists in this kernel: the bias arrives by one small DMA per tile, the C
// rows leave through inline-asm stores, so hipcc places no vmcnt wait of its own and the counted waits below are the
// only ones.  vmcnt is in order: the first wait after an epilogue also retires all but PIECES of its stores.
// ---------------------------------------------------------------------------------------------------------
template <int ACT>   // 0 none, 1 GELU, 2 ReLU (compile-time: see lds_epilogue of gemm_kernel)
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(GemmParams p, int vblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PP_GEMM_TIMELINE
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long rt_loop0 = 0, rt_loop1 = 0, ct_loop0 = 0, ct_loop1 = 0, ct_epi = 0;
#endif
  constexpr int BM = 192, BN = 192, WGN = 4, STAGES = 3, BK = 64;
  constexpr int PA = 3, PB = 3, PIECES = PA + PB, TM = 6, TN = 3;
  constexpr int A_BYTES = BM * ROW_BYTES, STAGE_BYTES = (BM + BN) * ROW_BYTES;
  constexpr int BIAS_OFF = STAGES * STAGE_BYTES, BIAS_SLOT = BN * 4;   // 3 slots of 192 floats behind the ring
  constexpr int CPASS = 2, PBM = BM / CPASS, CS = BN * 2 + 16, CPR = BN * 2 / 16;   // C staging: 96 whole rows per pass
  static_assert(PBM * CS <= STAGE_BYTES && PBM == BM / 2, "a C pass (the rows of one wave row) must fit one ring buffer");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int prow = lane >> 3, pchunk = lane & 7, frow = lane & 15, fq = lane >> 4;
  const int nkt = p.Kd / BK;
  const int epi = p.epilogue;

  // virtual block id -> tile, exactly the per-launch kernel's XCD-aware order (ids b and b + 8 share an XCD; the
  // stride gridDim.x is a multiple of 8, so a workgroup's tiles keep its XCD's slice of the order)
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
    for (; vb < vblocks; vb += (int)gridDim.x)
      if (decode(vb, tm, tn)) return vb;
    return -1;
  };
  if (next_valid((int)blockIdx.x) < 0) return;

  const char *zero_line = (const char *)g_zero_page +
                          ((((blockIdx.x * 29 + wave) * 8 + prow) & 7) * 128 + pchunk * 16) +
                          (((blockIdx.x * 13 + wave * 5) & 7) * 8192);
  const unsigned lds0 = lds_offset_of(smem);
  const int lchunk_off = (pchunk ^ prow) * 16;

  // ---- staging stream: the next K-tile to stage is (s_vb, s_kt)
  int s_vb = next_valid((int)blockIdx.x), s_kt = 0, s_slot = 0, s_n0 = 0;
  const char *a_src[PA];
  const char *w_src[PB];
  bool w_ok[PB];
  auto stage_setup = [&]() __attribute__((always_inline)) {
    int tm, tn;
    decode(s_vb, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    s_n0 = n0;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int r = (wave * PA + j) * 8 + prow;
      a_src[j] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda * 2 + lchunk_off;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int r = n0 + (wave * PB + j) * 8 + prow;
      w_ok[j] = r < p.N;
      w_src[j] = w_ok[j] ? p.W + (size_t)r * p.ldw * 2 + lchunk_off : zero_line + ((j * 5 + 3) & 7) * 1024;
    }
  };
  stage_setup();
  unsigned st_ldsA = 0, st_ldsB = 0;
  size_t st_koff = 0;
  // Every iteration of the stream stages one K-tile, so that every iteration is the same straight-line code and
  // the counted vmcnt(PIECES) holds to the end: once the workgroup's tiles are exhausted (the last two
  // iterations) the pieces re-read the last K-tile into the two buffers nobody will read again.
  auto stage_begin = [&](int buf) __attribute__((always_inline)) {      // latch (s_vb, s_kt) for the pieces of this iteration
    const bool st_live = s_vb >= 0;
    st_ldsA = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * PA * 1024);
    st_ldsB = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + A_BYTES + wave * PB * 1024);
    st_koff = (size_t)s_kt * ROW_BYTES;
    if (st_live && s_kt == 0 && wave == 0 && (epi & PP_EPI_BIAS)) {
      // this tile's bias segment -> LDS slot (one 768-byte DMA by wave 0; lanes beyond N stay unwritten, never stored)
      if (lane < BN / 4 && s_n0 + lane * 4 < p.N)
        glds16(p.bias + s_n0 + lane * 4, __builtin_amdgcn_readfirstlane(lds0 + BIAS_OFF + s_slot * BIAS_SLOT));
    }
  };
  auto stage_piece = [&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < PA) {
      glds16(a_src[q] + st_koff, st_ldsA + q * 1024);
    } else {
      constexpr int j = q - PA;
      glds16(w_ok[j] ? w_src[j] + st_koff : w_src[j], st_ldsB + j * 1024);
    }
  };
  auto stage_end = [&]() __attribute__((always_inline)) {               // advance; crossing into the next tile re-derives the source rows
    if (s_vb < 0) return;
    if (s_kt + 1 == nkt) {
      const int nv = next_valid(s_vb + (int)gridDim.x);
      if (nv < 0) {          // exhausted: keep the sources of the last K-tile (dummy re-reads from here on)
        s_vb = -1;
        return;
      }
      s_vb = nv;
      s_kt = 0;
      s_slot = s_slot == 2 ? 0 : s_slot + 1;
      stage_setup();
    } else {
      ++s_kt;
    }
  };
  auto stage_all = [&](int buf) __attribute__((always_inline)) {
    stage_begin(buf);
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      (stage_piece(std::integral_constant<int, Q>{}), ...);
    }(std::make_integer_sequence<int, PIECES>{});
    stage_end();
  };

  f32x4 acc[TM][TN];
  constexpr int GROUPS = 2 * TM;
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char *ldsA = smem + buf * STAGE_BYTES;
    const char *ldsB = ldsA + A_BYTES;
    [&]<int... S>(std::integer_sequence<int, S...>) {
      ([&] {
        constexpr int s = S;
        uint4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int ra = wm * (BM / 2) + i * 16 + frow;
          af[i] = *reinterpret_cast<const uint4 *>(ldsA + ra * ROW_BYTES + (((4 * s + fq) ^ (ra & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int rb = wn * (BN / WGN) + j * 16 + frow;
          bf[j] = *reinterpret_cast<const uint4 *>(ldsB + rb * ROW_BYTES + (((4 * s + fq) ^ (rb & 7)) << 4));
        }
        [&]<int... I>(std::integer_sequence<int, I...>) {
          ([&] {
            constexpr int i = I;
            constexpr int grp = s * TM + i;
            [&]<int... Q>(std::integer_sequence<int, Q...>) {
              ([&] {
                if constexpr ((Q * GROUPS) / PIECES == grp) {
                  __builtin_amdgcn_sched_barrier(0);
                  stage_piece(std::integral_constant<int, Q>{});
                  __builtin_amdgcn_sched_barrier(0);
                }
              }(), ...);
            }(std::make_integer_sequence<int, PIECES>{});
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                  *reinterpret_cast<bf16x8 *>(&bf[j]), *reinterpret_cast<bf16x8 *>(&af[i]), acc[i][j], 0, 0, 0);
          }(), ...);
        }(std::make_integer_sequence<int, TM>{});
      }(), ...);
    }(std::make_integer_sequence<int, 2>{});
  };

  // ---- epilogue of the tile just accumulated, through ring buffer `buf` (the one its last K-tile was read from: no
  // DMA targets it until the next iteration of the stream, the other two buffers hold the next tile's first K-tiles).
  constexpr int ST_PER_PASS = (PBM * CPR + 511) / 512;           // store instructions per wave and row pass
  constexpr int ST_PER_TILE = CPASS * ST_PER_PASS;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long c_base = (unsigned long long)p.C;
  // raw buffer descriptor of C (stride 0, byte-granular range check): offsets at or beyond M * ldc * 2 bytes are dropped
  const i32x4 c_srd = {(int)(unsigned)c_base, (int)(unsigned)((c_base >> 32) & 0xFFFFu),
                       (int)(unsigned)((size_t)p.M * p.ldc * 2), 0x00020000};
  auto epilogue = [&](int buf, int m0, int n0, int slot) __attribute__((always_inline)) {
    char *cst = smem + buf * STAGE_BYTES;
    // every wave first turns its accumulators into packed bf16 (bias, activation: the VALU part runs on all 8 waves at
    // once), then the two wave rows take turns through the one free ring buffer: 96 whole rows (384 B each) per pass
    uint2 pk[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (epi & PP_EPI_BIAS)
        b4 = *reinterpret_cast<const float4 *>(smem + BIAS_OFF + slot * BIAS_SLOT + (wn * (BN / WGN) + j * 16 + fq * 4) * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
        if constexpr (ACT == 1) gelu4<bf16_t>(v);
        if constexpr (ACT == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        pk[i][j].x = pack_bf16x2(v[0], v[1]);
        pk[i][j].y = pack_bf16x2(v[2], v[3]);
      }
    }
    const int ncols16 = max(0, min(CPR, (p.N - n0) * 2 / 16));
#pragma unroll
    for (int pass = 0; pass < CPASS; ++pass) {
      __syncthreads();                   // every wave has finished reading the K-tile (pass 0) / storing the previous pass
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            *reinterpret_cast<uint2 *>(cst + (i * 16 + frow) * CS + (wn * (BN / WGN) + j * 16 + fq * 4) * 2) = pk[i][j];
      }
      __syncthreads();
      // EXACTLY ST_PER_PASS store instructions per wave and pass, whatever the tile's edges: rows / columns outside the
      // matrix get an out-of-range buffer offset (the descriptor's range check drops them) instead of a branch around
      // the store, because the counted waits of the next tile's first iterations rely on the number (see the stream)
#pragma unroll
      for (int it = 0; it < ST_PER_PASS; ++it) {
        const int c = tid + it * 512;
        const int lr = c / CPR, cc = c - lr * CPR;
        const int row = m0 + pass * PBM + lr;
        const bool ok = c < PBM * CPR && row < p.M && cc < ncols16;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = *reinterpret_cast<const u32x4 *>(cst + (c < PBM * CPR ? lr * CS + cc * 16 : 0));
        unsigned off = 0xFFFFFFF0u;
        if (ok) {
          if (epi & PP_EPI_HEADMAJOR) {          // [3][heads][M][head_dim], see gemm_kernel
            const int n = n0 + cc * 8, Cc = p.hm_K * p.hm_HW;
            const int which = n / Cc, rem = n - which * Cc, head = rem / p.hm_HW, d = rem - head * p.hm_HW;
            off = (unsigned)(((((size_t)which * p.hm_K + head) * p.M + row) * p.hm_HW + d) * 2);
          } else {
            off = (unsigned)(((size_t)row * p.ldc + n0 + cc * 8) * 2);
          }
        }
        // invisible to hipcc's vmcnt bookkeeping on purpose (a compiler-counted store ahead of the K-loop's back edge
        // makes it guard the loop with vmcnt(0), which drains the DMA ring every K-tile)
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(c_srd) : "memory");
      }
    }
  };

  // ---- the stream: fill two K-tiles, then one identical iteration per K-tile, tile after tile
  stage_all(0);
  stage_all(1);
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
  ct_loop0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  int buf = 0, c_slot = 0, post_store = 0;
  for (int t_vb = next_valid((int)blockIdx.x); t_vb >= 0; t_vb = next_valid(t_vb + (int)gridDim.x)) {
    int tm, tn;
    decode(t_vb, tm, tn);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int last = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      // vmcnt counts loads, LDS-DMA and stores together, in issue order.  Right after an epilogue the queue of a wave
      // reads [K-tile 0'] [K-tile 1'] [ST_PER_TILE stores]: retiring K-tile 0' must not wait for the stores behind it
      // (their acknowledgements take microseconds when every CU writes), so the first two waits of a tile leave the
      // stores outstanding as well; from the third iteration on the stores are older than what is waited for.
      if (post_store > 0) {
        wait_vmcnt<PIECES + ST_PER_TILE>();
        --post_store;
      } else {
        wait_vmcnt<PIECES>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      stage_begin(nb);
      compute(buf);
      stage_end();
      last = buf;
      if (++buf == STAGES) buf = 0;
    }
#ifdef PP_GEMM_TIMELINE
    const unsigned long long ce0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    epilogue(last, tm * BM, tn * BN, c_slot);
#ifdef PP_GEMM_TIMELINE
    ct_epi += __builtin_amdgcn_s_memtime() - ce0;
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    post_store = 2;
    c_slot = c_slot == 2 ? 0 : c_slot + 1;
  }
#ifdef PP_GEMM_TIMELINE
  rt_loop1 = __builtin_amdgcn_s_memrealtime();
  ct_loop1 = __builtin_amdgcn_s_memtime();
#endif
  wait_vmcnt<0>();      // the dummy pieces of the last two iterations must not outlive the workgroup's LDS allocation
#ifdef PP_GEMM_TIMELINE
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = rt_entry; o[1] = rt_loop0; o[2] = rt_loop1; o[3] = rt_end;
    o[4] = __builtin_amdgcn_s_getreg(63492);
    o[5] = __builtin_amdgcn_s_getreg(63508);
    o[6] = ct_loop1 - ct_loop0 + 1; o[7] = ct_epi;      // stream cycles, of which epilogues
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------
// "Duo" form: 192x192 tile, FOUR waves (2 x 2, 96x96 per wave), K-tiles of 32 (64-byte rows), 3 LDS stages of
// 24 KB = 72 KB per workgroup, <= 256 VGPRs: TWO workgroups are resident per CU (one wave of each per SIMD).
//
// Why: the per-launch timeline (tools/gemm_timeline.py) shows a workgroup of the 8-wave form spending 35-50 % of its
// life outside the K-loop -- pipeline fill, the residual / bias fetch, and an epilogue in which all 256 CUs store in
// the same phase -- while nothing else can run on its CU (144 KB of LDS).  Two half-size workgroups per CU drift out
// of phase on their own: one's fill / epilogue runs under the other's K-loop, and inside the K-loop one's DMA issue
// and LDS reads run under the other's MFMAs.  96x96 wave tiles also read a third less LDS per flop than 96x48.
// Plain layers only (no gather / row map / LayerNorm modes / fp8); epilogue straight from the accumulators
// (8-byte bf16 or 16-byte fp32 stores per lane: no LDS staging to wait for, its inefficiency hides under the
// neighbour workgroup).  Swizzle for 64-byte rows: physical 16-B chunk = logical ^ perm[(row >> 2) & 3],
// perm = {0, 3, 2, 1}: every 16-lane group of a ds_read_b128 (rows r .. r+15, chunk = lane >> 4) covers all 16
// bank slots of the 256-byte LDS row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PP_GEMM_TIMELINE
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
  unsigned long long rt_loop0 = 0, rt_loop1 = 0;
#endif
#ifndef PP_DUO_DELAY
#define PP_DUO_DELAY 0       /* units of s_sleep(127) ~ 3.9 us; measured: 0 best (fc1 87.2 / 90.9 / 101 us at 0 / 2 / 4) */
#endif
  // Optional start stagger of the second slot of every CU (block ids b and b + 256 share a CU on a >= 512-workgroup
  // launch; speed only, never correctness).  Tried because two workgroups that start together run the same phases at
  // the same time; measured a loss at every delay, so it is off.
  if (PP_DUO_DELAY > 0 && gridDim.x >= 512 && ((blockIdx.x >> 8) & 1)) {
    for (int d = 0; d < PP_DUO_DELAY; ++d) __builtin_amdgcn_s_sleep(127);
  }
  constexpr int BM = 192, BN = 192, STAGES = 3, BK = 32, RB = 64;          // RB: bytes of K per staged row
  constexpr int PA = 3, PB = 3, PIECES = PA + PB, TM = 6, TN = 6;           // 1-KiB pieces (16 rows) per wave per K-tile
  constexpr int A_BYTES = BM * RB, STAGE_BYTES = (BM + BN) * RB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int epi = p.epilogue;

  int tm, tn;
  if (p.blocked) {
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
  const int nkt = p.Kd / BK;

  // ---- staging: lane -> (row in piece = lane >> 2, physical chunk = lane & 3); logical chunk = physical ^ perm
  const int prow = lane >> 2, pchunk = lane & 3;
  const int perm_p = (0x1230 >> (4 * ((prow >> 2) & 3))) & 3;      // {0, 3, 2, 1}[(row >> 2) & 3]
  const int lchunk_off = (pchunk ^ perm_p) * 16;
  const char *zero_line = (const char *)g_zero_page + (((blockIdx.x * 29 + wave) * 16 + prow) & 63) * 128 + pchunk * 16;
  const char *a_src[PA];
  const char *w_src[PB];
  bool w_ok[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (wave * PA + j) * 16 + prow;
    a_src[j] = p.A + (size_t)min(m0 + r, p.M - 1) * p.lda * 2 + lchunk_off;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int r = n0 + (wave * PB + j) * 16 + prow;
    w_ok[j] = r < p.N;
    w_src[j] = w_ok[j] ? p.W + (size_t)r * p.ldw * 2 + lchunk_off : zero_line;
  }
  const unsigned lds0 = lds_offset_of(smem);
  unsigned st_ldsA = 0, st_ldsB = 0;
  size_t st_koff = 0;
  auto stage_begin = [&](int kt, int buf) __attribute__((always_inline)) {
    st_ldsA = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + wave * PA * 1024);
    st_ldsB = __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + A_BYTES + wave * PB * 1024);
    st_koff = (size_t)kt * RB;
  };
  auto stage_piece = [&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < PA) {
      glds16(a_src[q] + st_koff, st_ldsA + q * 1024);
    } else {
      constexpr int j = q - PA;
      glds16(w_ok[j] ? w_src[j] + st_koff : w_src[j], st_ldsB + j * 1024);
    }
  };
  auto stage_all = [&](int kt, int buf) __attribute__((always_inline)) {
    stage_begin(kt, buf);
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      (stage_piece(std::integral_constant<int, Q>{}), ...);
    }(std::make_integer_sequence<int, PIECES>{});
  };

  f32x4 acc[TM][TN];
  // fragment offsets: row (w? * 96 + 16 i + frow), chunk fq ^ perm[(frow >> 2) & 3] (i only adds multiples of 16 rows)
  const int perm_f = (0x1230 >> (4 * ((frow >> 2) & 3))) & 3;
  const unsigned offA = (wm * 96 + frow) * RB + ((fq ^ perm_f) << 4);
  const unsigned offB = A_BYTES + (wn * 96 + frow) * RB + ((fq ^ perm_f) << 4);
  auto compute = [&](int buf, auto do_stage_c) __attribute__((always_inline)) {
    constexpr bool DO_STAGE = decltype(do_stage_c)::value;
    const char *sb = smem + buf * STAGE_BYTES;
    uint4 bf[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const uint4 *>(sb + offB + j * 16 * RB);
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        constexpr int i = I;
        const uint4 af = *reinterpret_cast<const uint4 *>(sb + offA + i * 16 * RB);
        if constexpr (DO_STAGE) {
          __builtin_amdgcn_sched_barrier(0);
          stage_piece(std::integral_constant<int, i>{});        // PIECES == TM: one piece per MFMA group
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8 *>(&bf[j]),
                                                              *reinterpret_cast<const bf16x8 *>(&af), acc[i][j], 0, 0, 0);
      }(), ...);
    }(std::make_integer_sequence<int, TM>{});
  };
  static_assert(PIECES == TM, "one DMA piece per MFMA group");

  // ---- pipeline fill, then the additive epilogue terms straight into the accumulators (residual / row bias)
#pragma unroll
  for (int s_ = 0; s_ < STAGES - 1; ++s_)
    if (s_ < nkt) stage_all(s_, s_);
  const float *Rb = p.residual;
  const float *init_base = (epi & PP_EPI_RESIDUAL) ? Rb : ((epi & PP_EPI_ROWBIAS) ? p.rowbias : nullptr);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 96 + i * 16 + frow;
    const size_t rowbase = (epi & PP_EPI_RESIDUAL) ? (size_t)m * p.ldc
                                                   : (size_t)(m % (p.rowbias_period > 0 ? p.rowbias_period : 1)) * p.ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 96 + j * 16 + fq * 4;
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
      if (init_base && m < p.M && n < p.N) {
        const float4 t = *reinterpret_cast<const float4 *>(init_base + rowbase + n);
        c = f32x4{t.x, t.y, t.z, t.w};
      }
      acc[i][j] = c;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);     // retire them with a wait hipcc models (see gemm_kernel): no vmcnt(0) in the loop
#ifdef PP_GEMM_TIMELINE
  rt_loop0 = __builtin_amdgcn_s_memrealtime();
#endif

  int buf = 0, kt = 0;
  for (; kt + STAGES - 1 < nkt; ++kt) {
    wait_vmcnt<(STAGES - 2) * PIECES>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int nb = buf + STAGES - 1;
    if (nb >= STAGES) nb -= STAGES;
    stage_begin(kt + STAGES - 1, nb);
    compute(buf, std::true_type{});
    if (++buf == STAGES) buf = 0;
  }
  for (; kt < nkt; ++kt) {
    if (kt + STAGES - 1 <= nkt) {
      wait_vmcnt<(STAGES - 2) * PIECES>();
    } else {
      wait_vmcnt<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(buf, std::false_type{});
    if (++buf == STAGES) buf = 0;
  }

#ifdef PP_GEMM_TIMELINE
  rt_loop1 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- epilogue: the tile goes through the (now idle) LDS ring in column passes and leaves as whole rows (16 B per
  // lane, contiguous per row); lane (frow, fq) of a wave owns row m and 4 consecutive columns of every 16-wide tile.
  const bool out_f32 = (epi & PP_EPI_OUT_F32) != 0;
  auto staged = [&]<int OES>(std::integral_constant<int, OES>) __attribute__((always_inline)) {
    constexpr int NPASS = OES == 2 ? 2 : 4;          // 96 bf16 or 48 f32 columns per pass: 192 x 208 B = 39 KB
    constexpr int PBN = BN / NPASS, CS = PBN * OES + 16, CPR = PBN * OES / 16;
    constexpr int JPP = PBN / 16;                    // 16-wide tiles of one wave per pass (6 or 3)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      __syncthreads();
      const int wn_of_pass = pass * PBN / 96;        // which wave column owns this pass
      if (wn == wn_of_pass) {
        const int j0 = (pass * PBN - wn_of_pass * 96) / 16;
#pragma unroll
        for (int jj = 0; jj < JPP; ++jj) {
          const int j = j0 + jj;
          const int n = n0 + wn * 96 + j * 16 + fq * 4;
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((epi & PP_EPI_BIAS) && n < p.N) b4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            // the accumulator index must be a compile-time constant: select by unrolled comparison
            f32x4 c = acc[i][0];
#pragma unroll
            for (int q = 1; q < TN; ++q) c = (j == q) ? acc[i][q] : c;
            float v[4] = {c[0] + b4.x, c[1] + b4.y, c[2] + b4.z, c[3] + b4.w};
            if (epi & PP_EPI_GELU) gelu4<bf16_t>(v);
            if (epi & PP_EPI_RELU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            char *dst = smem + (wm * 96 + i * 16 + frow) * CS + (jj * 16 + fq * 4) * OES;
            if constexpr (OES == 4) {
              *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              uint2 pk;
              pk.x = pack_bf16x2(v[0], v[1]);
              pk.y = pack_bf16x2(v[2], v[3]);
              *reinterpret_cast<uint2 *>(dst) = pk;
            }
          }
        }
      }
      __syncthreads();
      const int ncols16 = max(0, min(CPR, (p.N - n0 - pass * PBN) * OES / 16));
      for (int c = tid; c < BM * CPR; c += 256) {
        const int lr = c / CPR, cc = c - lr * CPR;
        if (m0 + lr >= p.M || cc >= ncols16) continue;
        const uint4 v = *reinterpret_cast<const uint4 *>(smem + lr * CS + cc * 16);
        *reinterpret_cast<uint4 *>(p.C + ((size_t)(m0 + lr) * p.ldc + n0 + pass * PBN) * OES + cc * 16) = v;
      }
    }
  };
  if (out_f32) staged(std::integral_constant<int, 4>{});
  else staged(std::integral_constant<int, 2>{});
#ifdef PP_GEMM_TIMELINE
  if ((p.epilogue & (1 << 30)) && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = reinterpret_cast<unsigned long long *>(const_cast<float *>(p.rowbias)) +
                            ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = rt_entry; o[1] = rt_loop0; o[2] = rt_loop1; o[3] = rt_end;
    o[4] = __builtin_amdgcn_s_getreg(63492);
    o[5] = __builtin_amdgcn_s_getreg(63508);
    o[6] = 1; o[7] = (unsigned long long)tm << 32 | (unsigned)tn;
  }
#endif
}

// pp_gemm_quad.hip: tiles 18 - 20 (four waves, one per SIMD, 128x96 / 96x144 / 96x128 wave tiles, persistent stream;
// 15 - 17: the per-launch forms, lab builds only)
int gemm_quad_launch(const GemmParams &p, int cfg, dim3 grid, hipStream_t s);
void gemm_quad_tile_shape(int cfg, int *bm, int *bn);

}  // namespace pp

#ifndef PP_CFG5_VS_CFG3
#define PP_CFG5_VS_CFG3 1.0
#endif
#ifndef PP_CFG3_SPEEDUP
#define PP_CFG3_SPEEDUP 1.3  // measured: one 8-wave 3-stage 192x192 tile per CU vs two co-resident 4-wave tiles
#endif

extern "C" int pp_gemm(const pp_gemm_args *a, void *stream) {
  using namespace pp;
  PP_REQUIRE(a, "pp_gemm: null args");
  PP_REQUIRE(a->dtype == PP_F32 || a->dtype == PP_BF16 || a->dtype == PP_FP8, "pp_gemm: bad dtype %d", a->dtype);
  const int es = a->dtype == PP_BF16 ? 2 : (a->dtype == PP_FP8 ? 1 : 4);
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
  PP_REQUIRE(!(a->epilogue & (PP_EPI_ROWSTATS | PP_EPI_LNFOLD)),
             "pp_gemm: the LayerNorm-fusion epilogues (PP_EPI_ROWSTATS / PP_EPI_LNFOLD) were removed in round 2: measured "
             "+11..16 us per fused GEMM against the 12 us LayerNorm launch they replace");
  if (a->epilogue & PP_EPI_HEATMAP)
    PP_REQUIRE(a->hm_K >= a->N && a->hm_HW > 0 && a->hm_temperature != 0.f, "pp_gemm: bad heatmap epilogue");
  if (a->epilogue & PP_EPI_HEADMAJOR)
    PP_REQUIRE(a->dtype == PP_BF16 && a->hm_K > 0 && a->hm_HW > 0 && a->hm_HW % 8 == 0 && a->N == 3 * a->hm_K * a->hm_HW &&
                   a->ldc == a->N && (a->N & 7) == 0 && ((uintptr_t)a->C & 15) == 0 && !a->out_rowmap &&
                   a->batch <= 1 && a->splitk <= 1 && a->tile != 14 &&
                   !(a->epilogue & (PP_EPI_OUT_F32 | PP_EPI_HEATMAP | PP_EPI_RESIDUAL | PP_EPI_FUSE_FINAL | PP_EPI_OUT_FP8)),
               "pp_gemm: PP_EPI_HEADMAJOR serves the bf16 qkv projection: N = 3 * heads (hm_K) * head_dim (hm_HW), head_dim a "
               "multiple of 8, C 16-byte aligned, a plain single launch (not tile 14)");
  if (a->epilogue & PP_EPI_FUSE_FINAL) {
    PP_REQUIRE(a->dtype == PP_BF16 && a->N == 256 && a->final_w && a->final_b && a->hm_K > 0 && a->hm_K <= 32 &&
                   a->hm_HW > 0 && a->hm_temperature != 0.f && (a->tile == 9 || a->tile == 0) &&
                   !(a->epilogue & (PP_EPI_OUT_F32 | PP_EPI_HEATMAP | PP_EPI_RESIDUAL)) &&
                   ((uintptr_t)a->final_w & 15) == 0,
               "pp_gemm: PP_EPI_FUSE_FINAL serves bf16 layers with N = 256 outputs (tile 9), K <= 32 keypoint maps");
  }
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
  p.splitk = a->splitk > 1 ? a->splitk : 1;
  p.strideA_k = a->strideA_k; p.strideW_k = a->strideW_k; p.strideC_k = a->strideC_k; p.strideRowoff_k = a->strideRowoff_k;
  if (a->epilogue & PP_EPI_HEADMAJOR) { p.hm_K = a->hm_K; p.hm_HW = a->hm_HW; }
  if (p.splitk > 1) {
    PP_REQUIRE(a->epilogue == PP_EPI_OUT_F32 && a->dtype != PP_FP8 && !a->out_rowmap,
               "pp_gemm: split-K launches write plain f32 partials (epilogue must be PP_EPI_OUT_F32 alone)");
    PP_REQUIRE(!a->rowoff || a->Kd % (a->seg_len > 0 ? a->seg_len : a->Kd) == 0,
               "pp_gemm: split-K depth %d must be a whole number of gather segments (%d)", a->Kd, a->seg_len);
  }
  p.epilogue = a->epilogue;
  p.hm_K = a->hm_K; p.hm_HW = a->hm_HW; p.hm_temperature = a->hm_temperature;
  p.colsum = a->colsum;
  p.out_scale = a->out_scale;
  p.final_w = (const char *)a->final_w; p.final_b = a->final_b;
  if (a->dtype == PP_FP8) {
    PP_REQUIRE(a->colsum && !a->rowoff && !a->out_rowmap && !(a->epilogue & (PP_EPI_ROWBIAS | PP_EPI_HEATMAP)),
               "pp_gemm: fp8 needs colsum = per-column dequantisation scales and a plain (non-gather, non-fused) GEMM");
    if (a->epilogue & PP_EPI_OUT_FP8)
      PP_REQUIRE(a->out_scale > 0.f && !(a->epilogue & PP_EPI_OUT_F32), "pp_gemm: PP_EPI_OUT_FP8 needs out_scale > 0");
  } else {
    PP_REQUIRE(!(a->epilogue & PP_EPI_OUT_FP8), "pp_gemm: PP_EPI_OUT_FP8 is an fp8-GEMM epilogue");
  }
  const int batch = (a->batch > 0 ? a->batch : 1) * (a->splitk > 1 ? a->splitk : 1);   // grid.y
  PP_REQUIRE(batch <= 65535, "pp_gemm: batch too large");
  // Tile configuration.  0 = auto, 1 = 128x128 (4 waves, 2 stages), 2 = 192x96 (4 waves, 2 stages),
  // 3 = 192x192 (8 waves, 3 stages, one workgroup per CU), 4 = 192x128 (8 waves, 3 stages),
  // 5 = 384x128 (8 waves, 2 stages; the N = 256 deconvolution layers), 6 = 192x192 wave-specialised
  // (8 consumer + 4 producer waves, 3 stages), 7 = 192x384 (8 waves, 2 stages; wide-N layers such as fc1),
  // 8 = 256x256 (8 waves, 2 stages), 9 = 192x256 (8 waves, 2 stages; N = 256 layers: one column tile, A read once).  Auto: cost = rounds of co-resident workgroups x padded tile area / relative per-CU
  // throughput of the configuration.
  PP_REQUIRE(a->tile >= 0 && a->tile <= 20 && a->tile != 11 && a->tile != 12,
             "pp_gemm: bad tile selector %d (11 / 12: round-2 experiments, removed)", a->tile);
  auto rounds = [&](int bm, int bn, int slots) {
    const long long tiles = (long long)cdiv(a->M, bm) * cdiv(a->N, bn) * batch;
    return (tiles + slots - 1) / slots;
  };
  const bool vec = (a->N & 3) == 0 && (a->ldc & 3) == 0 && !(a->epilogue & PP_EPI_HEATMAP);
  int cfg = vec ? a->tile : 1;
  if (a->epilogue & PP_EPI_FUSE_FINAL) cfg = 9;     // the one configuration that holds all 256 channels of a pixel
  if (cfg == 0) {
    const double c1 = (double)rounds(128, 128, 512) * 128 * 128 * 2;   // 2 workgroups share a CU
    const double c2 = (double)rounds(192, 96, 512) * 192 * 96 * 2;
    const double c3 = (double)rounds(192, 192, 256) * 192 * 192 / PP_CFG3_SPEEDUP;
    const double c4 = (double)rounds(192, 128, 256) * 192 * 128 / (PP_CFG3_SPEEDUP * 0.9);
    cfg = 1;
    double best = c1;
    if (c2 < best) { best = c2; cfg = 2; }
    if (c4 < best) { best = c4; cfg = 4; }
    if (c3 <= best) { best = c3; cfg = 3; }
    if (a->N <= 256) {  // narrow outputs (deconvolution layers): a taller tile restores the flop/byte ratio
      const double c5 = (double)rounds(384, 128, 256) * 384 * 128 / (PP_CFG3_SPEEDUP * PP_CFG5_VS_CFG3);
      if (c5 < best) { best = c5; cfg = 5; }
    }
  }
  int bm = cfg == 1 ? 128 : (cfg == 5 ? 384 : (cfg == 8 ? 256 : 192));
  int bn = cfg == 1 ? 128 : (cfg == 2 ? 96 : ((cfg == 3 || cfg == 6 || cfg == 10 || cfg == 13 || cfg == 14) ? 192 : (cfg == 7 ? 384 : ((cfg == 8 || cfg == 9) ? 256 : 128))));
  if (cfg >= 15) gemm_quad_tile_shape(cfg, &bm, &bn);
  p.tiles_m = cdiv(a->M, bm);
  p.tiles_n = cdiv(a->N, bn);
  dim3 grid;
  auto set_grid = [&](int c) {   // c = tile configuration; tiles_m / tiles_n are set
    const int rn_ = std::min((c >= 3) ? 4 : 8, p.tiles_n);
    const long long nblk = (long long)cdiv(p.tiles_m, 8) * cdiv(p.tiles_n, rn_);
    p.rn = rn_;
    p.blocked = nblk >= 16 ? 1 : 0;
    grid = dim3(p.blocked ? (unsigned)(((nblk + 7) / 8) * 8 * 8 * rn_) : (unsigned)(p.tiles_m * p.tiles_n), batch);
  };
  set_grid(cfg);
  {
    const int oes = (a->dtype == PP_F32 || (a->epilogue & PP_EPI_OUT_F32)) ? 4 : ((a->epilogue & PP_EPI_OUT_FP8) ? 1 : 2),
              per16 = 16 / oes;
    p.lds_epilogue = (vec && a->N % per16 == 0 && a->ldc % per16 == 0 && ((uintptr_t)a->C & 15) == 0 &&
                      (a->strideC % per16) == 0)
                         ? 1
                         : 0;
  }
  if (a->epilogue & PP_EPI_FUSE_FINAL)
    PP_REQUIRE(p.lds_epilogue && vec, "pp_gemm: PP_EPI_FUSE_FINAL runs inside the LDS epilogue: C must be 16-byte aligned "
                                      "(and N = ldc = 256)");
  hipStream_t s = (hipStream_t)stream;
  if (cfg >= 15) {
    // quad forms (pp_gemm_quad.hip): plain bf16 -> bf16 layers, K-tiles of 32 walked in pairs behind a 4-deep ring
    PP_REQUIRE(a->dtype == PP_BF16 && !a->rowoff && !a->out_rowmap && vec && p.lds_epilogue && batch == 1 &&
                   a->Kd % 64 == 0 && a->Kd >= 128 && (a->N & 7) == 0 &&
                   !(a->epilogue & ~(PP_EPI_BIAS | PP_EPI_GELU | PP_EPI_RELU | PP_EPI_HEADMAJOR | (1 << 30))),
               "pp_gemm: tiles 15 - 20 (four-wave forms) serve plain bf16 -> bf16 GEMMs with bias / GELU / ReLU epilogues, "
               "K >= 128");
    if (cfg >= 18)
      PP_REQUIRE(a->M % bm == 0 && a->N % bn == 0 && a->Kd >= 512 && (unsigned long long)a->M * a->ldc * 2 < 0xFFFFFFF0ull,
                 "pp_gemm: tiles 18 - 20 (four-wave stream) need M %% %d == 0, N %% %d == 0, K >= 512 and C below 4 GiB", bm, bn);
    return gemm_quad_launch(p, cfg, grid, s);
  }
  if (cfg == 14) {
    // duo form (gemm_duo_kernel): plain bf16 layers (K a multiple of 64 like every bf16 tile; it stages 32-deep K-tiles)
    PP_REQUIRE(a->dtype == PP_BF16 && !a->rowoff && !a->out_rowmap && vec && batch == 1 && a->Kd % 64 == 0 &&
                   !(a->epilogue & ~(PP_EPI_BIAS | PP_EPI_GELU | PP_EPI_RELU | PP_EPI_RESIDUAL | PP_EPI_OUT_F32 |
                                     PP_EPI_ROWBIAS | (1 << 30))) &&
                   (!(a->epilogue & (PP_EPI_RESIDUAL | PP_EPI_ROWBIAS)) || (a->epilogue & PP_EPI_OUT_F32)) &&
                   p.lds_epilogue,
               "pp_gemm: tile 14 (two workgroups per CU) serves plain bf16 GEMMs (bias / GELU / ReLU / f32 residual) whose "
               "output rows are whole 16-byte chunks (N, ldc multiples of 8 for bf16 / 4 for f32 outputs, C 16-byte aligned)");
    constexpr int lds = 3 * (192 + 192) * 64;
    static thread_local unsigned long long attr_mask = 0;
    int dev_ = 0;
    if (attr_needed(attr_mask, dev_))
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_duo_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(gemm_duo_kernel, grid, dim3(256), lds, s, p);
    PP_CHECK_LAUNCH("gemm_duo_kernel");
    return 0;
  }
  if (cfg == 13) {
    // persistent 192x192 form (gemm_persist_kernel): plain bf16 -> bf16 layers only
    PP_REQUIRE(a->dtype == PP_BF16 && !a->rowoff && !a->out_rowmap && vec && p.lds_epilogue && batch == 1 &&
                   !(a->epilogue & ~(PP_EPI_BIAS | PP_EPI_GELU | PP_EPI_RELU | PP_EPI_HEADMAJOR | (1 << 30))) &&
                   (unsigned long long)a->M * a->ldc * 2 < 0xFFFFFFF0ull,
               "pp_gemm: tile 13 (persistent) serves plain bf16 -> bf16 GEMMs with bias / GELU / ReLU epilogues only "
               "(C below 4 GiB: its rows leave through 32-bit buffer offsets)");
    static int ncu = 0;
    if (ncu == 0) {
      int dev = 0, n = 0;
      PP_CHECK_HIP(hipGetDevice(&dev));
      PP_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
      ncu = n > 0 ? n : 256;
    }
    const int vblocks = (int)grid.x;
    int wgs = std::min(vblocks, ncu);
    if (wgs >= 8) wgs &= ~7;                      // a multiple of 8 keeps every workgroup's tiles on its XCD
    constexpr int lds = 3 * (192 + 192) * ROW_BYTES + 3 * 192 * 4;
    static thread_local unsigned long long attr_mask = 0;
    int dev_ = 0;
    if (attr_needed(attr_mask, dev_)) {
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_persist_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_persist_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      PP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_persist_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    if (a->epilogue & PP_EPI_GELU) hipLaunchKernelGGL(gemm_persist_kernel<1>, dim3(wgs), dim3(512), lds, s, p, vblocks);
    else if (a->epilogue & PP_EPI_RELU) hipLaunchKernelGGL(gemm_persist_kernel<2>, dim3(wgs), dim3(512), lds, s, p, vblocks);
    else hipLaunchKernelGGL(gemm_persist_kernel<0>, dim3(wgs), dim3(512), lds, s, p, vblocks);
    PP_CHECK_LAUNCH("gemm_persist_kernel");
    return 0;
  }
#define PP_LAUNCH_GEMM_P(T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, NWP_, PP_)                         \
  do {                                                                                                \
    constexpr int lds = gemm_lds_bytes(BM_, BN_, ST_);                                                \
    static thread_local unsigned long long attr_mask = 0;                                             \
    int dev_ = 0;                                                                                     \
    if (attr_needed(attr_mask, dev_))                                                                 \
      PP_CHECK_HIP(hipFuncSetAttribute(                                                               \
          reinterpret_cast<const void *>(gemm_kernel<T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, NWP_, PP_>), \
          hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                          \
    hipLaunchKernelGGL((gemm_kernel<T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, NWP_, PP_>), grid,     \
                       dim3(64 * (WGM_ * WGN_ + NWP_)), lds, s, p);                                   \
  } while (0)
#define PP_LAUNCH_GEMM_PP(T, BM_, BN_, WGM_, WGN_, ST_)                                               \
  do {                                                                                                \
    if (gather) PP_LAUNCH_GEMM_P(T, BM_, BN_, WGM_, WGN_, ST_, true, true, 0, true);               \
    else PP_LAUNCH_GEMM_P(T, BM_, BN_, WGM_, WGN_, ST_, false, true, 0, true);                     \
  } while (0)
#define PP_LAUNCH_GEMM_W(T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, NWP_) PP_LAUNCH_GEMM_P(T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, NWP_, false)
#define PP_LAUNCH_GEMM_V(T, BM_, BN_, WGM_, WGN_, ST_, G_, V_) PP_LAUNCH_GEMM_W(T, BM_, BN_, WGM_, WGN_, ST_, G_, V_, 0)
#define PP_LAUNCH_GEMM(T, BM_, BN_, WGM_, WGN_, ST_)                                                  \
  do {                                                                                                \
    if (gather) PP_LAUNCH_GEMM_V(T, BM_, BN_, WGM_, WGN_, ST_, true, true);                           \
    else PP_LAUNCH_GEMM_V(T, BM_, BN_, WGM_, WGN_, ST_, false, true);                                 \
  } while (0)
  const bool gather = a->rowoff != nullptr;
#ifdef PP_GEMM_LAB   // experiment builds (tools/build_lab.sh): only the plain bf16 192x192 forms, compiles in seconds
  if (a->dtype != PP_BF16 || gather || !vec || !(cfg == 3 || cfg == 6 || cfg == 10))
    return fail("pp_gemm (lab build): only plain bf16 tiles 3, 6, 10 (and 13 / 14, dispatched above)");
  if (cfg == 3) PP_LAUNCH_GEMM_V(bf16_t, 192, 192, 2, 4, 3, false, true);
  else if (cfg == 6) PP_LAUNCH_GEMM_W(bf16_t, 192, 192, 2, 4, 3, false, true, 4);
  else PP_LAUNCH_GEMM_P(bf16_t, 192, 192, 2, 4, 3, false, true, 0, true);
#else
  if (a->dtype == PP_FP8) {
    PP_REQUIRE(vec && p.lds_epilogue, "pp_gemm: fp8 needs the vector / LDS epilogue path (aligned N, ldc, C)");
    if (!(cfg == 2 || cfg == 3 || cfg == 10)) {
      PP_REQUIRE(a->tile == 0, "pp_gemm: fp8 is built for tiles 2, 3 and 10, got tile %d", cfg);
      cfg = 3;
      p.tiles_m = cdiv(a->M, 192);
      p.tiles_n = cdiv(a->N, 192);
      set_grid(3);
    }
    if (cfg == 2) PP_LAUNCH_GEMM_V(fp8_t, 192, 96, 2, 2, 2, false, true);
    else if (cfg == 10) PP_LAUNCH_GEMM_P(fp8_t, 192, 192, 2, 4, 3, false, true, 0, true);
    else PP_LAUNCH_GEMM_V(fp8_t, 192, 192, 2, 4, 3, false, true);
  } else if (!vec) {  // ragged N / heatmap epilogue: element-wise variant, 128x128 only
    if (a->dtype == PP_BF16) {
      if (gather) PP_LAUNCH_GEMM_V(bf16_t, 128, 128, 2, 2, 2, true, false);
      else PP_LAUNCH_GEMM_V(bf16_t, 128, 128, 2, 2, 2, false, false);
    } else {
      if (gather) PP_LAUNCH_GEMM_V(float, 128, 128, 2, 2, 2, true, false);
      else PP_LAUNCH_GEMM_V(float, 128, 128, 2, 2, 2, false, false);
    }
  } else if (a->dtype == PP_BF16) {
    if (cfg == 1) PP_LAUNCH_GEMM(bf16_t, 128, 128, 2, 2, 2);
    else if (cfg == 2) PP_LAUNCH_GEMM(bf16_t, 192, 96, 2, 2, 2);
    else if (cfg == 3) PP_LAUNCH_GEMM(bf16_t, 192, 192, 2, 4, 3);
    else if (cfg == 4) PP_LAUNCH_GEMM(bf16_t, 192, 128, 2, 4, 3);
    else if (cfg == 5) PP_LAUNCH_GEMM(bf16_t, 384, 128, 2, 4, 2);
    else if (cfg == 7) PP_LAUNCH_GEMM(bf16_t, 192, 384, 2, 4, 2);
    else if (cfg == 8) PP_LAUNCH_GEMM(bf16_t, 256, 256, 2, 4, 2);
    else if (cfg == 9) PP_LAUNCH_GEMM(bf16_t, 192, 256, 2, 4, 2);
    else if (cfg == 10) PP_LAUNCH_GEMM_PP(bf16_t, 192, 192, 2, 4, 3);
    else if (gather) PP_LAUNCH_GEMM_W(bf16_t, 192, 192, 2, 4, 3, true, true, 4);
    else PP_LAUNCH_GEMM_W(bf16_t, 192, 192, 2, 4, 3, false, true, 4);
  } else {
    if (cfg == 1) PP_LAUNCH_GEMM(float, 128, 128, 2, 2, 2);
    else if (cfg == 2) PP_LAUNCH_GEMM(float, 192, 96, 2, 2, 2);
    else if (cfg == 3) PP_LAUNCH_GEMM(float, 192, 192, 2, 4, 3);
    else if (cfg == 4) PP_LAUNCH_GEMM(float, 192, 128, 2, 4, 3);
    else if (cfg == 5) PP_LAUNCH_GEMM(float, 384, 128, 2, 4, 2);
    else if (cfg == 7) PP_LAUNCH_GEMM(float, 192, 384, 2, 4, 2);
    else if (cfg == 8 || cfg == 9) return fail("pp_gemm: the 256-wide tiles are built for bf16 only (fp32 fragments do not fit the register file)");
    else if (cfg == 10) PP_LAUNCH_GEMM_PP(float, 192, 192, 2, 4, 3);
    else if (gather) PP_LAUNCH_GEMM_W(float, 192, 192, 2, 4, 3, true, true, 4);
    else PP_LAUNCH_GEMM_W(float, 192, 192, 2, 4, 3, false, true, 4);
  }
#endif
#undef PP_LAUNCH_GEMM_V
#undef PP_LAUNCH_GEMM_W
#undef PP_LAUNCH_GEMM_L
#undef PP_LAUNCH_GEMM_P
#undef PP_LAUNCH_GEMM_PP
#undef PP_LAUNCH_GEMM
  PP_CHECK_LAUNCH("gemm_kernel");
  return 0;
}
