// score_tiles_f16t.h -- the fp16 filter engine of score_tiles_f16.h fed from a TILE-ORDERED fp16 copy
// of the rows: no LDS staging of the rows at all.
//
// Layout of the copy ("T layout"): blocks of 32 consecutive rows; inside a block the 16-byte pieces are
// stored in the order v_mfma_f32_32x32x16_f16 wants its B operand,
//
//     piece (kb, lane) at byte  blk * 32 * DM * 2 + kb * 1024 + lane * 16,   lane = j + 32 g
//       = row 32 blk + j of the pass's rows, columns 16 kb + 8 g .. + 7      (kb < DM / 16)
//
// so ONE global_load_dwordx4 per k-block (1 KiB contiguous per wave instruction -- the shape that streams
// fastest, tools/probes/read_patterns3.hip: 7.0 TB/s against 5.5 for 32-byte row segments of a row-major
// copy) puts the B operand of that k-block straight into registers.  Against the row-major engine a
// wave-tile saves its 32 ds_write_b64 + 17 ds_read_b128 of row traffic and the write -> read round trip
// through the window (the LDS array was as busy as the matrix pipe there: 4.3 K cycles per 256 rows per
// CU each); what is left in LDS is the table (hi / lo planes), read as A operands.
// Rows of the tail columns and the per-row error stay in xt[row] (8 bytes per row, row-major).
// The caller guarantees: crow0 % 32 == 0, the copy holds kHalfSlackRows readable rows past the last row.
#pragma once
#include "score_tiles_f16.h"

namespace hsgk {

template <int MB = 2, int PLANES = 2>
__host__ __device__ constexpr size_t half_t_lds_bytes(int d) {
  return (size_t)PLANES * 32 * MB * (half_main_cols(d) + 16 + 8) * 2 + 16;
}

// table block -> fp16 hi / lo planes [32 MB][RS] (zero padded), the layout of score_tiles_half
template <int NW, int MB, int PLANES>
__device__ __forceinline__ void stage_half_planes(unsigned char *lds_raw, const float *__restrict__ table, int d,
                                                  int kvalid) {
  constexpr int TR = 32 * MB;
  const int DM = half_main_cols(d), RS = DM + 16 + 8;
  uint32_t *ch32 = reinterpret_cast<uint32_t *>(lds_raw);
  uint32_t *cl32 = ch32 + (PLANES == 2 ? TR * (RS >> 1) : 0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int RS2 = RS >> 1, dp = d >> 1;
  constexpr int PPL = 4;
  for (int k0 = w; k0 < TR; k0 += 4 * NW) {
    float2 v[4][PPL];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u * NW, kvalid - 1);
#pragma unroll
      for (int i = 0; i < PPL; ++i)
        v[u][i] = *reinterpret_cast<const float2 *>(table + (int64_t)k * d + 2 * min(lane + 64 * i, dp - 1));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u * NW;
      if (k < TR) {
        const bool live = k < kvalid;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
          const int pr = lane + 64 * i;
          if (pr < RS2) {
            uint32_t hi = 0u, lo = 0u;
            if (live && pr < dp) f16_split2(v[u][i].x, v[u][i].y, hi, lo);
            ch32[k * RS2 + pr] = hi;
            if constexpr (PLANES == 2) cl32[k * RS2 + pr] = lo;
          }
        }
      }
    }
  }
}

// Epi(tile, acc, err) as in score_tiles_half.  NFULL = d / 64 (2 or 4) at compile time: a whole wave-tile
// (NFULL sets of four 1-KiB loads) is in flight ahead of its use.
template <int NW, int NFULL, class Epi, int MB = 2, int PLANES = 2>
__device__ __forceinline__ void score_tiles_half_t(const _Float16 *__restrict__ xmT, const uint2 *__restrict__ xt,
                                                   int d, const float *__restrict__ table, int kvalid,
                                                   int64_t crow0, int nrows, unsigned char *lds_raw, Epi &epi,
                                                   bool stage_table = true) {
  static_assert(NFULL == 2 || NFULL == 4, "d / 64 must be 2 or 4");
  constexpr int TPX = NW * 32;
  constexpr int DM = NFULL * 64;
  constexpr int RS = DM + 16 + 8;
  constexpr int TR = 32 * MB;
  const uint16_t *chs = reinterpret_cast<const uint16_t *>(lds_raw);
  const uint16_t *cls = chs + (PLANES == 2 ? TR * RS : 0);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  const bool has_tail = d > DM;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int ntile = max(0, (nrows - wu * 32 + TPX - 1) / TPX);

  // ---- the row stream: asm loads, counted waits (see score_tiles_f16.h for why)
  constexpr int64_t blk_bytes = (int64_t)32 * DM * 2;
  const uint32_t voff = (uint32_t)lane * 16u;
  const char *wbase = reinterpret_cast<const char *>(xmT) + ((crow0 >> 5) + wu) * blk_bytes;
  int ld_tile = 0, ld_s = 0;
  auto load_next = [&](u32x4 (&pre)[4]) {
    const char *tb = wbase + (int64_t)ld_tile * (NW * blk_bytes) + ld_s * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      // "+v": the set keeps ITS registers from load to load.  With "=v" the compiler may give the next tile's loads
      // fresh registers and copy them into place at the loop's back edge -- before the counted wait, i.e. before
      // the data has landed (it takes an asm output for available at once).
      asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(pre[i]) : "v"(voff), "s"(tb + i * 1024));
    const bool wrap = ld_s + 1 == NFULL;
    ld_s = wrap ? 0 : ld_s + 1;
    ld_tile += wrap ? 1 : 0;
  };
#define HSGK_TWAIT4(N, P) \
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N))

  f32x16 acc[MB];
  struct Ops { f16x8 ah[MB]; f16x8 al[PLANES == 2 ? MB : 1]; };
  auto load_table_ops = [&](int col0, Ops &o) {
#if defined(HSGK_T_DEBUG) && HSGK_T_DEBUG == 3                               // probe: no table reads
    const f16x8 c = {(_Float16)(col0 * 0.001f), 1, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < MB; ++m) { o.ah[m] = c; if constexpr (PLANES == 2) o.al[m] = c; }
    return;
#endif
    const uint16_t *hp = chs + j * RS + col0 + 8 * g;
#pragma unroll
    for (int m = 0; m < MB; ++m) o.ah[m] = *reinterpret_cast<const f16x8 *>(hp + m * 32 * RS);
    if constexpr (PLANES == 2) {
      const uint16_t *lp = cls + j * RS + col0 + 8 * g;
#pragma unroll
      for (int m = 0; m < MB; ++m) o.al[m] = *reinterpret_cast<const f16x8 *>(lp + m * 32 * RS);
    }
  };
  auto mfma_ops = [&](const Ops &o, const f16x8 &b, bool first) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if defined(HSGK_T_DEBUG) && (HSGK_T_DEBUG == 1 || HSGK_T_DEBUG == 4)      // elimination probes: no matrix work
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (first) acc[m] = zero;
      acc[m][0] += (float)o.ah[m][0] * (float)b[0] + (float)o.al[0][1] * (float)b[7];
    }
    return;
#endif
#pragma unroll
    for (int m = 0; m < MB; ++m)
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.ah[m], b, first ? zero : acc[m], 0, 0, 0);
    if constexpr (PLANES == 2) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.al[m], b, acc[m], 0, 0, 0);
    }
  };
  // one set = four k-blocks whose B operands are the four loaded registers; the table operands of
  // k-block n + 1 are read before the MFMAs of k-block n (two operand sets)
  auto compute_set = [&](const u32x4 (&pre)[4], int s, bool first) {
    Ops o0, o1;
    load_table_ops(s * 64, o0);
    __builtin_amdgcn_sched_barrier(0);
    load_table_ops(s * 64 + 16, o1);
    mfma_ops(o0, __builtin_bit_cast(f16x8, pre[0]), first);
    __builtin_amdgcn_sched_barrier(0);
    load_table_ops(s * 64 + 32, o0);
    mfma_ops(o1, __builtin_bit_cast(f16x8, pre[1]), false);
    __builtin_amdgcn_sched_barrier(0);
    load_table_ops(s * 64 + 48, o1);
    mfma_ops(o0, __builtin_bit_cast(f16x8, pre[2]), false);
    __builtin_amdgcn_sched_barrier(0);
    mfma_ops(o1, __builtin_bit_cast(f16x8, pre[3]), false);
  };
  const uint32_t toff = (uint32_t)(w * 32 + j) * 8u;
  auto load_tail = [&](int tile, uint2 &v) {
    const char *tb = reinterpret_cast<const char *>(xt + crow0 + (int64_t)tile * TPX);
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(toff), "s"(tb));
  };

  u32x4 preA[4], preB[4], preC[NFULL == 4 ? 4 : 1], preD[NFULL == 4 ? 4 : 1];
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) { preA[i] = z; preB[i] = z; }
    if constexpr (NFULL == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { preC[i] = z; preD[i] = z; }
    }
  }
  if (ntile > 0) {
    load_next(preA);
    load_next(preB);
    if constexpr (NFULL == 4) { load_next(preC); load_next(preD); }
  }
  if (stage_table) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // nobody still reads the previous table
    stage_half_planes<NW, MB, PLANES>(lds_raw, table, d, kvalid);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (ntile <= 0) return;
  uint2 tailv = {0u, 0u};
  // The landed set moves to `cur` (16 register moves) so that the NEXT tile's loads into the same set are issued
  // BEFORE the set's 16 MFMAs, not after them: the wave's share of the memory queue stays full while it computes.
  u32x4 cur[4];
#define HSGK_T_STEP(PRE, S, FIRST)                                            \
  HSGK_TWAIT4(4 * (NFULL - 1) + 1, PRE);                                      \
  cur[0] = PRE[0]; cur[1] = PRE[1]; cur[2] = PRE[2]; cur[3] = PRE[3];         \
  asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));  \
  __builtin_amdgcn_sched_barrier(0);                                          \
  load_next(PRE);                                                             \
  __builtin_amdgcn_sched_barrier(0);                                          \
  compute_set(cur, S, FIRST);                                                 \
  __builtin_amdgcn_sched_barrier(0);
  for (int tile = 0; tile < ntile; ++tile) {
    load_tail(tile, tailv);
    HSGK_T_STEP(preA, 0, true)
    HSGK_T_STEP(preB, 1, false)
    if constexpr (NFULL == 4) {
      HSGK_T_STEP(preC, 2, false)
      HSGK_T_STEP(preD, 3, false)
    }
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tailv) : "n"(4 * NFULL));
    if (has_tail) {
      const u32x4 tv = {g == 0 ? tailv.x : 0u, 0u, 0u, 0u};
      Ops o;
      load_table_ops(DM, o);
      mfma_ops(o, __builtin_bit_cast(f16x8, tv), false);
    }
#if defined(HSGK_T_DEBUG) && (HSGK_T_DEBUG == 2 || HSGK_T_DEBUG == 4)      // probe: no epilogue
    if (acc[0][0] + acc[1][1] + __uint_as_float(tailv.y) == 12345.678f) epi(tile, acc, 0.0f);
#else
    epi(tile, acc, __uint_as_float(tailv.y));
#endif
  }
  HSGK_TWAIT4(0, preA);
  HSGK_TWAIT4(0, preB);
  if constexpr (NFULL == 4) { HSGK_TWAIT4(0, preC); HSGK_TWAIT4(0, preD); }
#undef HSGK_T_STEP
#undef HSGK_TWAIT4
}

}  // namespace hsgk
