// score_tiles_f16.h -- first-level FILTER of the E-step: scores from an fp16 copy of
// the rows (half the HBM bytes of the fp32 rows) against an fp16 hi/lo split of the
// table, two v_mfma_f32_32x32x16_f16 per 16 columns:
//
//     acc += ch * xh + cl * xh          xh = fp16(x),  c = ch + cl + ec
//
// It decides every row whose two best approximate scores are far enough apart and
// hands the rest (a few per cent) to the bf16x3 filter (score_tiles_bf16.h, on the
// fp32 rows), which in turn hands its ambiguous rows to the exact fp32 chain -- the
// labels stay bit-identical to the canonical arithmetic.
//
// Error of the approximate score against the canonical fp32 chain, for unit-norm
// rows and centroids (|x_i|, |c_i| <= 1, |c| <= 1 + 1.6e-5):
//
//     row rounding   |sum c_i (x_i - xh_i)| <= |c| |x - xh|  =  |c| * err(row)
//        err(row) = |x - xh|_2 is MEASURED when the copy is made (fp32 sum of the exact
//        per-element residuals, inflated by 1.0001 against its own rounding) and stored
//        per row; fp16 RNE gives about 2^-12.3 |x| = 2.0e-4, 2.4x below the worst case
//        2^-11 |x| = 4.88e-4 (subnormal residuals are part of the measurement;
//        v_cvt_f16_f32 and the MFMA keep fp16 subnormals, tools/probes/mfma_f16_probe.hip)
//     table residual sum |ec_i| |xh_i|      <= 2^-22 + 2^-25 sum|x_i|          = 7.2e-7
//     fp16 MFMA accumulation, 2*21 instr.   <= 42 * 2^-21 * 1.01               = 2.02e-5
//        (products of two fp16 are exact in fp32; measured |D - exact| <= 2^-22.4 sum|terms|
//         per instruction on gfx950, same probe; 2^-21 used; d <= 322)
//     fp32 chain of the oracle vs the real number  gamma_322                   = 1.92e-5
//                                           E1(row) <= 1.00002 err(row) + 4.02e-5
//
// half_gap(err) = 2.0002 err + 8.1e-5 > 2 E1(row): a row whose two best approximate
// scores differ by more has a strictly unique exact argmax (typically gap = 4.8e-4; with
// the worst-case constant it would be 1.06e-3 and twice as many rows would be undecided).
//
// Layout of the fp16 copy: the first DM = 64 * (d / 64) columns of every row in
// xm[n][DM] -- rows are whole 128-byte lines, which streams 8 % faster than the
// 1032 / 528-byte strides (tools/probes/read_patterns2.hip: 6.3 vs 5.8 TB/s) -- and
// the remaining d - DM <= 2 columns (the location features of emb_loc) packed in
// xt[n] = {two fp16 tail columns, err(row) as fp32 bits} (8 bytes per row: a 32-row
// tile reads two lines instead of 32).
// Table block as two fp16 planes [64][RS] in LDS; every wave streams 32 rows x 64
// columns (128 B per row) per chunk through a private double-buffered window
// [32][72] -- no conversion, no barrier in the column loop, four chunks (16 KiB per
// wave) in flight.
#pragma once
#include "common.h"
#include "score_tiles.h"

namespace hsgk {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// decision threshold of a row whose copy has rounding error err = |x - xh|_2
__device__ inline float half_gap(float err) { return fmaf(2.0002f, err, 8.1e-5f); }
// err of a row whose residual was not measured: 2^-11 |x| + the subnormal floor
constexpr float kHalfWorstErr = 4.89e-4f;

// rows the engine may read past the end of a pass (partial last tile + look-ahead):
// the fp16 copy is allocated with this much slack
constexpr int kHalfSlackRows = 2 * 8 * 32 + 64;

// columns of a row kept in the main array / in the packed tail word
__host__ __device__ constexpr int half_main_cols(int d) { return d & ~63; }

// (a, b) -> packed fp16 pair and the packed fp16 pair of the residuals (RNE; the
// residual subtraction is exact)
__device__ inline void f16_split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// LDS of the engine: PLANES table planes of MB*32 rows, NW windows of NBUF buffers
template <int NW, int MB = 2, int PLANES = 2, int NBUF = 2>
__host__ __device__ constexpr size_t half_lds_bytes(int d) {
  return (size_t)PLANES * 32 * MB * (half_main_cols(d) + 16 + 8) * 2 + (size_t)NW * NBUF * 32 * 72 * 2 + 16;
}

// shapes the fp16 engine accepts: an even number of 64-column chunks (prefetch depth
// 2 or 4), at most two tail columns (d = C + 2 with C % 64 == 0), and few enough
// columns for the error bound above (gamma_d, MFMA count)
__host__ __device__ inline bool half_shape_ok(int d) {
  const int nfull = d / 64;
  return d >= 128 && d <= 322 && (nfull & 1) == 0 && d - nfull * 64 <= 2;
}
// the hi-plane-only variant (64 < K <= 128, see HalfWide in kmeans.hip) has half the MFMA
// instructions per column block and its own constant: d up to 450
__host__ __device__ inline bool half_wide_shape_ok(int d) {
  const int nfull = d / 64;
  return d >= 128 && d <= 450 && (nfull & 1) == 0 && d - nfull * 64 <= 2;
}

// The table planes written from registers (a caller that has just produced the table rows in
// LDS / registers skips the global round trip of the engine's own staging and then passes
// stage_table = false).  Same layout, same row ownership as the staging loop of
// score_tiles_half: wave w owns table rows w + NW*u, lane l the column pairs l + 64*i;
// cv[u][i] = the two fp32 values of that pair (anything where the row or pair is out of range).
// d <= 322 (half_shape_ok): three pairs per lane cover a plane row.  The caller separates this
// from the last read of whatever the plane region held before with a workgroup barrier.
constexpr int kHalfRegPairs = 3;
template <int NW, int MB = 2, int PLANES = 2>
__device__ __forceinline__ void half_planes_from_regs(unsigned char *lds_raw, int d, int kvalid,
                                                      const float2 (&cv)[32 * MB / NW][kHalfRegPairs]) {
  constexpr int TR = 32 * MB;
  const int DM = half_main_cols(d), RS = DM + 16 + 8, RS2 = RS >> 1, dp = d >> 1;
  uint32_t *ch32 = reinterpret_cast<uint32_t *>(lds_raw);
  uint32_t *cl32 = ch32 + (PLANES == 2 ? TR * RS2 : 0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < TR / NW; ++u) {
    const int k = w + NW * u;
    const bool live = k < kvalid;
#pragma unroll
    for (int i = 0; i < kHalfRegPairs; ++i) {
      const int pr = lane + 64 * i;
      if (pr < RS2) {
        uint32_t hi = 0u, lo = 0u;
        if (live && pr < dp) f16_split2(cv[u][i].x, cv[u][i].y, hi, lo);
        ch32[k * RS2 + pr] = hi;
        if constexpr (PLANES == 2) cl32[k * RS2 + pr] = lo;
      }
    }
  }
}

#ifdef HSGK_SMALL_TIMING                   // tools/probes/small_timing.py: cycles per phase, workgroup 0
__device__ unsigned long long g_small_ts[12];
#define HSGK_ETS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
    atomicAdd(&g_small_ts[i], now_ - ets_); ets_ = now_; } } while (0)
#else
#define HSGK_ETS(i) do { } while (0)
#endif

// Epi(tile, acc, err): lane (j, h) holds acc[m][r] = approximate score of table row
// m*32 + (r&3) + 8*(r>>2) + 4*h for row tile*NW*32 + w*32 + j of the pass, and err =
// the measured rounding error of that row's copy.
// MB: table blocks of 32 rows (2 or 4); PLANES: 2 = hi + lo planes of the table, 1 = hi
// plane only (the caller's bound then carries the table's rounding error); NBUF: window
// buffers per wave (1 is enough for correctness -- a wave's LDS operations execute in
// order -- and saves LDS for the wide variants).
// AUX: a 16-byte per-row record aux[row] (state left by an earlier pass over the same rows)
// is prefetched with the tail and handed to Epi(tile, acc, err, aux).
// ZERO_C: the first MFMA of a tile takes the constant 0 as its C operand instead of a zeroed accumulator: no
// 16 MB accumulator writes per tile and no loop-carried accumulator values (with eight accumulator sets the
// loop phis cost a register-to-register copy of all 128 of them per tile).  It needs the number of 64-column
// chunks of a row at compile time (NFULL_CT = d / 64; 0 = taken from d at run time, ZERO_C then unavailable).
template <int NW, int DEPTH, class Epi, int MB = 2, int PLANES = 2, int NBUF = 2, bool AUX = false, bool ZERO_C = false,
          int NFULL_CT = 0>
__device__ __forceinline__ void score_tiles_half(const _Float16 *__restrict__ xm,
                                                 const uint2 *__restrict__ xt, int d,
                                                 const float *__restrict__ table, int kvalid,
                                                 int64_t crow0, int nrows, unsigned char *lds_raw,
                                                 Epi &epi, bool stage_table = true,
                                                 const uint4 *__restrict__ aux = nullptr) {
  constexpr int TPX = NW * 32;
  constexpr int KC = 64;               // columns per staged chunk (4 k-blocks)
  constexpr int XSB = 72;              // fp16 elements per staged row (64 + 8 pad: conflict-free b128 reads)
  constexpr int LOADS = 8;             // 8-byte loads per lane per chunk (4 rows x 128 B per instruction)
  const int DM = half_main_cols(d);
  const int RS = DM + 16 + 8;          // fp16 elements per table row (main + one tail k-block + pad)

  constexpr int TR = 32 * MB;          // table rows
  uint16_t *chs = reinterpret_cast<uint16_t *>(lds_raw);          // [TR][RS]
  uint16_t *cls = chs + (PLANES == 2 ? TR * RS : 0);               // [TR][RS] (PLANES == 2)
  uint16_t *xs = chs + PLANES * TR * RS;                           // [NW][NBUF][32][XSB]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
#ifdef HSGK_SMALL_TIMING
  unsigned long long ets_ = __builtin_readcyclecounter();
#endif

  static_assert(!ZERO_C || NFULL_CT > 0, "ZERO_C needs the chunk count at compile time");
  const int nfull = NFULL_CT > 0 ? NFULL_CT : DM / KC;
  const int tcol0 = DM;
  const bool has_tail = d > DM;                   // <= 2 tail columns (half_shape_ok), fed from xt
  const int wu = __builtin_amdgcn_readfirstlane(w);
  // tiles of THIS wave: a wave whose 32 rows of the (partial) last tile all lie past nrows stops
  // one tile early (no workgroup barrier below; with two waves per SIMD the other one then has
  // the matrix pipe to itself)
  const int ntile = max(0, (nrows - wu * 32 + TPX - 1) / TPX);
  const int nsteps = ntile * nfull;

  uint16_t *xw = xs + w * (NBUF * 32 * XSB);
  const int lpx = lane >> 4, lf = lane & 15;

  // ---- the row stream: explicit loads, explicit counted waits -----------------
  // Loads return in order, so "wait for the oldest of DEPTH sets" is s_waitcnt
  // vmcnt(8 * (DEPTH - 1)).  The compiler's own s_waitcnt insertion cannot be relied on
  // for that: it merges the outstanding-load state pessimistically at every control-flow
  // join (the tile loop, the epilogue's branches), and whether the steady-state loop ends
  // up with counted waits or with a full drain of the prefetch queue once per tile
  // (vmcnt(0): ~25 % slower) changed with unrelated edits.  So the chunk and tail loads
  // are inline asm (invisible to that pass) and the waits are written out by hand.
  //   * only asm loads are counted; compiler-visible VM ops in between (the epilogue's
  //     label store, the rare queue-overflow atomic) can only make a wait longer.
  //   * every wait names the registers it protects as "+v" operands, so their first use
  //     is ordered after it; a final vmcnt(0) keeps look-ahead loads from landing in
  //     registers the compiler has already handed to someone else.
  //   * every load is unconditional and unclamped: the caller guarantees kHalfSlackRows
  //     readable rows past the last row of xm / xt (the library's own buffers), so a
  //     partial last tile and the look-ahead chunks past the end simply read on; those
  //     rows' scores are never used (an x row only feeds its own accumulator column).
  // Address = wave-uniform 64-bit base (SGPR pair, advanced incrementally) + one constant
  // 32-bit lane offset.
  const uint32_t voff = (uint32_t)(lpx * DM + 4 * lf) * 2u;                   // bytes
  const char *wbase = reinterpret_cast<const char *>(xm + (crow0 + wu * 32) * DM);   // wave-uniform
  const int64_t tile_bytes = (int64_t)TPX * DM * 2, grp_bytes = (int64_t)4 * DM * 2;
  int ld_tile = 0, ld_q = 0;                                    // next chunk to load (uniform)
  auto load_next = [&](uint2 (&pre)[LOADS]) {
    const char *tb = wbase + ld_tile * tile_bytes + ld_q * (KC * 2);
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
      asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(pre[i]) : "v"(voff), "s"(tb + i * grp_bytes));
    const bool wrap = ld_q + 1 == nfull;
    ld_q = wrap ? 0 : ld_q + 1;
    ld_tile += wrap ? 1 : 0;
  };
#define HSGK_VMWAIT8(N, P)                                                              \
  asm volatile("s_waitcnt vmcnt(%8)"                                                    \
               : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), \
                 "+v"(P[6]), "+v"(P[7])                                                 \
               : "n"(N))
  auto store_chunk = [&](int buf, const uint2 (&pre)[LOADS]) {
    uint16_t *bp = xw + (buf % NBUF) * (32 * XSB);
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
      *reinterpret_cast<uint2 *>(bp + (lpx + 4 * i) * XSB + 4 * lf) = pre[i];
  };

  f32x16 acc[MB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
  };
  // operands of one k-block (16 columns): the row operand and the table operands of the
  // MB table blocks (hi, and lo when PLANES == 2).  The LDS reads of block n+1 are issued
  // before the MFMAs of block n (two register sets), so the per-wave chain is MFMA-bound,
  // not a sequence of exposed LDS latencies.
  struct Ops { f16x8 b; f16x8 ah[MB]; f16x8 al[PLANES == 2 ? MB : 1]; };
  auto load_table_ops = [&](int col0, Ops &o) {
    const uint16_t *hp = chs + j * RS + col0 + 8 * g;
#pragma unroll
    for (int m = 0; m < MB; ++m) o.ah[m] = *reinterpret_cast<const f16x8 *>(hp + m * 32 * RS);
    if constexpr (PLANES == 2) {
      const uint16_t *lp = cls + j * RS + col0 + 8 * g;
#pragma unroll
      for (int m = 0; m < MB; ++m) o.al[m] = *reinterpret_cast<const f16x8 *>(lp + m * 32 * RS);
    }
  };
  auto mfma_ops = [&](const Ops &o, bool first = false) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MB; ++m)
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.ah[m], o.b, first ? zero : acc[m], 0, 0, 0);
    if constexpr (PLANES == 2) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.al[m], o.b, acc[m], 0, 0, 0);
    }
  };
  auto kblock = [&](const f16x8 &b, int col0) {
    Ops o;
    o.b = b;
    load_table_ops(col0, o);
    mfma_ops(o);
  };
  auto compute_chunk = [&](int buf, int q, bool first = false) {
    const uint16_t *bp = xw + (buf % NBUF) * (32 * XSB) + j * XSB + 8 * g;
    Ops o0, o1;
    o0.b = *reinterpret_cast<const f16x8 *>(bp);
    load_table_ops(q * KC, o0);
    __builtin_amdgcn_sched_barrier(0);
    o1.b = *reinterpret_cast<const f16x8 *>(bp + 16);
    load_table_ops(q * KC + 16, o1);
    mfma_ops(o0, first);
    __builtin_amdgcn_sched_barrier(0);
    o0.b = *reinterpret_cast<const f16x8 *>(bp + 32);
    load_table_ops(q * KC + 32, o0);
    mfma_ops(o1);
    __builtin_amdgcn_sched_barrier(0);
    o1.b = *reinterpret_cast<const f16x8 *>(bp + 48);
    load_table_ops(q * KC + 48, o1);
    mfma_ops(o0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_ops(o1);
  };
  // tail k-block of a tile: the packed tail word of row j (k = 0, 1 of the block; the
  // rest of the block and the g = 1 half are zero), together with the row's err.  Two
  // lines per wave tile, issued at the top of the tile (nfull >= DEPTH chunk sets are
  // issued after it), so the epilogue never waits on memory.
  const uint32_t toff = (uint32_t)(w * 32 + j) * 8u;
  auto load_tail = [&](int tile, uint2 &v) {                  // unconditional, unclamped (see above)
    const char *tb = reinterpret_cast<const char *>(xt + crow0 + (int64_t)tile * TPX);
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(toff), "s"(tb));
  };
  const uint32_t aoff = (uint32_t)(w * 32 + j) * 16u;
  auto load_aux = [&](int tile, u32x4 &v) {
    const char *tb = reinterpret_cast<const char *>(aux + crow0 + (int64_t)tile * TPX);
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(aoff), "s"(tb));
  };
  auto finish_tile = [&](int tile, const uint2 &tw, const u32x4 &av) {
    if (has_tail) {
      const u32x4 tv = {g == 0 ? tw.x : 0u, 0u, 0u, 0u};
      kblock(__builtin_bit_cast(f16x8, tv), tcol0);
    }
    if constexpr (AUX) epi(tile, acc, __uint_as_float(tw.y), av);
    else epi(tile, acc, __uint_as_float(tw.y));
  };

  // DEPTH register sets rotate, each loaded DEPTH chunks (4 KiB per wave each) ahead
  // of its use; nfull % DEPTH == 0 (half_shape_ok + the launcher's choice), so a tile
  // always ends on the last set and there is ONE epilogue site.  The first sets are requested
  // before the table is staged: they do not depend on it (measured neutral so far: the staging
  // loads queue behind them).
  // DEPTH == 6 (four chunks per row only): SIX sets rotate over the four chunks of a tile, so a wave has one and a
  // half tiles (24 KiB) in flight and the set order repeats every three tiles (the tile loop is unrolled by three).
  // The time of a tile is the memory latency of a set plus whatever part of the wave's own work (16 MFMAs per set,
  // the epilogue) delays the re-issue of its slots; with the queue capped at one tile per wave that delay came
  // straight out of the stream (tools/probes/ab_tkernel.sh: 863 us, 766 without the epilogue, 700 for the bare loads).
  static_assert(DEPTH == 2 || DEPTH == 4 || (DEPTH == 6 && NFULL_CT == 4), "prefetch depth");
  uint2 preA[LOADS], preB[LOADS], preC[DEPTH >= 4 ? LOADS : 1], preD[DEPTH >= 4 ? LOADS : 1];
  uint2 preE[DEPTH == 6 ? LOADS : 1], preF[DEPTH == 6 ? LOADS : 1];
  if (nsteps > 0) {
    load_next(preA);
    load_next(preB);
    if constexpr (DEPTH >= 4) {
      load_next(preC);
      load_next(preD);
    }
    if constexpr (DEPTH == 6) {
      load_next(preE);
      load_next(preF);
    }
  }

  // ---- table block -> fp16 hi / lo planes (zero padded); a persistent caller
  //      skips this while consecutive passes use the same table
  if (stage_table) {
    // nobody still reads the previous table.  A bare barrier: only LDS reads are ordered here
    // (the caller has made the table itself visible), and __syncthreads() would also wait for the
    // row loads requested above.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // A wave converts whole table rows: lanes = column pairs (d is even, rows are 8-byte aligned:
    // one 8-byte load, one packed hi and one packed lo dword per pair), four rows of loads in flight;
    // the padding pairs of a row and the rows past kvalid are zeroed by the same lanes.  (The first
    // version -- one element per thread with a division by d and 2-byte LDS stores -- took 11 us of
    // the 29 us this kernel needs for a 256-row batch.)
    uint32_t *ch32 = reinterpret_cast<uint32_t *>(chs);
    uint32_t *cl32 = reinterpret_cast<uint32_t *>(cls);
    const int RS2 = RS >> 1, dp = d >> 1;                     // dwords per table row, column pairs per row
    constexpr int PPL = 4;                                    // pairs per lane and row (d <= 512)
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

  // A visible full wait: the staging loads above are consumed under per-lane conditions,
  // so on paths where a consumer block is skipped the compiler's model keeps them
  // "pending" -- and would protect their registers (reused for the accumulators) with
  // vmcnt(0) waits inside the streaming loop, draining the prefetch queue.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                         // table planes visible to all waves
  HSGK_ETS(5);
  if constexpr (!ZERO_C) zero_acc();
  if (nsteps <= 0) return;              // (wave-uniform)
  uint2 tailv = {0u, 0u};
  u32x4 auxv = {0u, 0u, 0u, 0u};
#define HSGK_HALF_STEP(BUF, PRE, QQ, FIRST)                                   \
  HSGK_VMWAIT8(8 * (DEPTH - 1), PRE);                                         \
  store_chunk(BUF, PRE);                                                      \
  __builtin_amdgcn_sched_barrier(0);                                          \
  load_next(PRE);                                                             \
  __builtin_amdgcn_sched_barrier(0);                                          \
  compute_chunk(BUF, QQ, FIRST);                                              \
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DEPTH == 6) {
    // (the tail / aux loads of a tile are older than the four sets issued during it: vmcnt(8 * 4) before the epilogue)
#define HSGK_HALF_TILE6(P0, P1, P2, P3)                                       \
    load_tail(tile, tailv);                                                   \
    if constexpr (AUX) load_aux(tile, auxv);                                  \
    HSGK_HALF_STEP(0, P0, 0, ZERO_C)                                          \
    HSGK_HALF_STEP(1, P1, 1, false)                                           \
    HSGK_HALF_STEP(0, P2, 2, false)                                           \
    HSGK_HALF_STEP(1, P3, 3, false)                                           \
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(tailv), "+v"(auxv) : "n"(8 * 4)); \
    finish_tile(tile, tailv, auxv);                                           \
    if constexpr (!ZERO_C) zero_acc();
    for (int tile = 0; tile < ntile;) {
      HSGK_HALF_TILE6(preA, preB, preC, preD)
      if (++tile >= ntile) break;
      HSGK_HALF_TILE6(preE, preF, preA, preB)
      if (++tile >= ntile) break;
      HSGK_HALF_TILE6(preC, preD, preE, preF)
      ++tile;
    }
#undef HSGK_HALF_TILE6
  } else
  for (int tile = 0; tile < ntile; ++tile) {
    load_tail(tile, tailv);
    if constexpr (AUX) load_aux(tile, auxv);
    if constexpr (NFULL_CT > 0) {
#pragma unroll
      for (int q = 0; q < NFULL_CT; q += DEPTH) {
        const bool first = ZERO_C && q == 0;          // (folds after unrolling)
        if constexpr (DEPTH == 4) {
          HSGK_HALF_STEP(0, preA, q, first)
          HSGK_HALF_STEP(1, preB, q + 1, false)
          HSGK_HALF_STEP(0, preC, q + 2, false)
          HSGK_HALF_STEP(1, preD, q + 3, false)
        } else {
          HSGK_HALF_STEP(0, preA, q, first)
          HSGK_HALF_STEP(1, preB, q + 1, false)
        }
      }
    } else {
      for (int q = 0; q < nfull; q += DEPTH) {
        if constexpr (DEPTH == 4) {
          HSGK_HALF_STEP(0, preA, q, false)
          HSGK_HALF_STEP(1, preB, q + 1, false)
          HSGK_HALF_STEP(0, preC, q + 2, false)
          HSGK_HALF_STEP(1, preD, q + 3, false)
        } else {
          HSGK_HALF_STEP(0, preA, q, false)
          HSGK_HALF_STEP(1, preB, q + 1, false)
        }
      }
    }
    // the tail (and aux) loads were issued before this tile's nfull >= DEPTH chunk sets
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(tailv), "+v"(auxv) : "n"(8 * DEPTH));
    finish_tile(tile, tailv, auxv);
    if constexpr (!ZERO_C) zero_acc();
  }
  HSGK_ETS(6);
  // drain the look-ahead sets; naming them keeps their registers reserved until here
  HSGK_VMWAIT8(0, preA);
  HSGK_VMWAIT8(0, preB);
  if constexpr (DEPTH >= 4) {
    HSGK_VMWAIT8(0, preC);
    HSGK_VMWAIT8(0, preD);
  }
  if constexpr (DEPTH == 6) {
    HSGK_VMWAIT8(0, preE);
    HSGK_VMWAIT8(0, preF);
  }
  HSGK_ETS(7);
#undef HSGK_HALF_STEP
#undef HSGK_VMWAIT8
}

}  // namespace hsgk
