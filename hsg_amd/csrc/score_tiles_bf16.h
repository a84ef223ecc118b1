// score_tiles_bf16.h -- bf16x3 "split" scoring engine: a fast, rigorously
// bounded FILTER in front of the exact fp32 engine (score_tiles.h).
//
// fp32 MFMA costs 64 cycles / instruction / SIMD and caps the E-step at the
// MFMA roof (~50 % of HBM peak at the clock the chip sustains).  Here every
// fp32 operand is split on the fly into two bf16 values, x = xh + xl + ex with
// xh = bf16(x), xl = bf16(x - xh) (|ex| <= 2^-16 |x|), and the score is
// approximated by three bf16 MFMAs per 16 columns
//
//     acc += ch * xh + cl * xh + ch * xl          (one fp32 accumulator)
//
// (v_mfma_f32_32x32x16_bf16: 32 cycles each -> 16/3 times cheaper than the fp32
// chain).  The approximation error against the canonical fp32 chain is bounded
// for unit-norm rows and centroids:
//
//     dropped products (xl cl, ex c, x ec)      <= 3.02 * 2^-16           = 4.62e-5
//     bf16 MFMA accumulation, 3*17 instructions <= 51 * 2^-22 * 1.01      = 1.23e-5
//        (measured on gfx950: |D - exact| <= 2^-23.1 * sum|terms| per instruction,
//         tools/probes/mfma_bf16_probe.hip; 2^-22 used)
//     fp32 chain of the oracle vs the real number  gamma_258               = 1.54e-5
//                                                              E           <= 7.4e-5
//
// A row whose two best approximate scores are more than kSplitGap = 1.6e-4 (> 2E)
// apart has a strictly unique exact argmax and gets its label here; every other
// row (a few per cent on i.i.d. data) is queued and re-scored EXACTLY by the fp32
// engine, so the labels are bit-identical to the canonical arithmetic.
//
// Layout: one workgroup (8 waves) per chunk part, table block (<= 64 rows) as
// two bf16 planes [64][RS] in LDS (RS = ceil16(d) + 8 elements: 16-byte aligned
// rows, conflict-free ds_read_b128), rows stream through wave-private double
// buffered hi/lo planes [32][40] exactly like the fp32 engine (no barrier in the
// column loop, global loads two chunks ahead).
#pragma once
#include "common.h"
#include "score_tiles.h"

namespace hsgk {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr float kSplitGap = 1.6e-4f;

// (a, b) -> packed hi pair and packed lo pair; v_cvt_pk_bf16_f32 rounds to
// nearest even, the residual subtraction is exact
__device__ inline void bf16_split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// fp16 variant of the split with SCALED residuals: x = xh + 2^-11 xl', xh = fp16(x), xl' = fp16((x - xh) 2^11)
// (fp16 carries 11 significant bits, so the two terms hold 22 -- bf16's two hold 16; the scaling keeps the
// residual in fp16's normal range).  Same three MFMAs per 16 columns, but the cross terms go to a second
// accumulator that is folded in with 2^-11 at the end of a tile:
//     acc  += ch * xh          acc2 += cl' * xh + ch * xl'          score = acc + 2^-11 acc2
// Error against the fp32 chain for |x|, |c| <= 1: dropped cl xl <= 2^-22, residual rounding 2 * 2^-22,
// MFMA accumulation 17 * 2^-22 (fp32 accumulate, measured per instruction) -- about 5e-6 in the worst case,
// the size of the fp32 chain's own rounding (gamma_258 = 1.5e-5); values must stay below fp16's 65504.
// Used where a TOLERANCE is the contract (the loss: exp(kappa s) amplifies a score error kappa-fold, and its
// backward divides by differences of those sums), not for the E-step filters, whose bounds are built on bf16.
__device__ inline void f16s_split2(float a, float b, uint32_t &hi, uint32_t &lo) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.0f;
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

template <int NW>
__host__ __device__ constexpr size_t split_lds_bytes(int d) {
  return (size_t)2 * 64 * (((d + 15) / 16) * 16 + 8) * 2 + (size_t)NW * 2 * 2 * 32 * 40 * 2 + 16;
}

// Epi(tile, acc): lane (j, h) holds acc[m][r] = approximate score of table row
// m*32 + (r&3) + 8*(r>>2) + 4*h for x row tile*NW*32 + w*32 + j.
//
// ROWS: the pass covers the rows x[rowlist[0 .. nrows)] (a gathered subset: the
// second filter level of the E-step) instead of x[crow0 .. crow0 + nrows).  The row
// ids of a wave's 32 rows are staged per tile in a wave-private LDS slot pair
// rl_lds[NW][2][32] (tile parity), loaded two tiles ahead: an id fetched from global
// right before its use would drain the whole prefetch queue (loads return in order).
// Epi may read the ids of tile t from the slot t & 1 inside its call.
// XPRE: x is not fp32 rows but the already split image of them IN OPERAND ORDER (loss.hip: loss_image_kernel) --
// for every 32 rows and 16-column block one KiB of hi words and one of lo words, lane l = 32 g + j holding the four
// words of row j's columns 8 g .. 8 g + 7: a wave's B operands of a k-block are two 16-byte buffer loads per lane
// straight into registers, contiguous per instruction, and the rows never pass through LDS (every pixel row is
// streamed once per prototype block, 48 times at P = 3 072: converting it each time was a third of that kernel's
// vector work, staging it through LDS half of its LDS instructions).  The caller guarantees d % 32 == 0 (no tail
// columns) and crow0 % 32 == 0; x points at the image of row 0.
// NFULLC > 0 (XPRE only; the caller guarantees d == 32 NFULLC): the tile loop is unrolled and, when the epilogue has
// the begin / piece / end form (loss.hip: LossFwdEpiFast), the per-score work of tile t - 1 is issued BETWEEN the
// matrix instructions of tile t -- an MFMA occupies the matrix pipe for 32 cycles during which the wave would issue
// nothing else; five vector instructions fit behind each one.  With two waves per SIMD (256 registers) the epilogue of
// one wave and the MFMAs of the other overlapped only by chance.
template <int NW, int DEPTH, class Epi, bool ROWS = false, bool F16S = false, bool XPRE = false, int NFULLC = 0>
__device__ __forceinline__ void score_tiles_split(const float *__restrict__ x, int d,
                                         const float *__restrict__ table, int kvalid,
                                         int64_t crow0, int nrows, unsigned char *lds_raw,
                                         Epi &epi, bool stage_table = true,
                                         const int32_t *__restrict__ rowlist = nullptr,
                                         int32_t *rl_lds = nullptr) {
  constexpr int NT = NW * 64;
  constexpr int TPX = NW * 32;
  constexpr int KC = 32;               // columns per staged chunk (2 k-blocks)
  constexpr int XSB = 40;              // bf16 elements per staged row (32 + 8 pad)
  constexpr int LOADS = 8;             // float2 per lane per chunk
  const int dk16 = ((d + 15) / 16) * 16;
  const int RS = dk16 + 8;             // bf16 elements per table row

  uint16_t *chs = reinterpret_cast<uint16_t *>(lds_raw);          // [64][RS]
  uint16_t *cls = chs + 64 * RS;                                   // [64][RS]
  uint16_t *xs = cls + 64 * RS;                                    // [NW][2 buf][2 plane][32][XSB]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  auto split2 = [](float a, float b, uint32_t &hi, uint32_t &lo) {
    if constexpr (F16S) f16s_split2(a, b, hi, lo); else bf16_split2(a, b, hi, lo);
  };

  // ---- table block -> bf16 hi / lo planes (zero padded); a persistent caller
  //      skips this while consecutive chunks use the same table
  if (stage_table) {
    __syncthreads();                       // nobody still reads the previous table
    uint32_t *z = reinterpret_cast<uint32_t *>(chs);
    for (int i = tid; i < 64 * RS; i += NT) z[i] = 0u;            // 2 planes * 64*RS*2 B / 4
    __syncthreads();
    const int half = d >> 1;
    const int total = kvalid * half;
    for (int f0 = 0; f0 < total; f0 += NT * 8) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = min(f0 + tid + NT * u, total - 1);
        v[u] = *reinterpret_cast<const float2 *>(table + 2 * (int64_t)f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + tid + NT * u;
        if (f < total) {
          const int k = f / half, col = 2 * (f - k * half);
          uint32_t hi, lo;
          split2(v[u].x, v[u].y, hi, lo);
          *reinterpret_cast<uint32_t *>(chs + k * RS + col) = hi;
          *reinterpret_cast<uint32_t *>(cls + k * RS + col) = lo;
        }
      }
    }
  }

  const int nfull = d / KC;
  const int tcol0 = nfull * KC;
  const int tblocks = (d - tcol0 + 15) / 16;      // tail k-blocks fed from global (<= 2)
  const int ntile = (nrows + TPX - 1) / TPX;
  const int nsteps = ntile * nfull;

  uint16_t *xw = xs + w * (2 * 2 * 32 * XSB);
  const int lpx = lane >> 4, lf2 = lane & 15;

  const int wu = __builtin_amdgcn_readfirstlane(w);
  int32_t *rlw = ROWS ? rl_lds + w * 64 : nullptr;            // [2][32] row ids of tiles t (slot t & 1)
  auto row_id = [&](int tile) -> int32_t {                     // id of this lane's row j in `tile` (clamped)
    return rowlist[max(min(tile * TPX + w * 32 + j, nrows - 1), 0)];
  };
  if constexpr (ROWS) {
    rlw[j] = row_id(0);
    rlw[32 + j] = row_id(1);
  }

  // Every load is issued UNCONDITIONALLY with clamped indices: the compiler's s_waitcnt
  // insertion merges the outstanding-load state pessimistically at control-flow joins,
  // and one conditional load_chunk in the steady-state loop turns "wait for the oldest
  // set" into vmcnt(0) -- a full drain of the prefetch queue once per tile.
  auto load_chunk = [&](int gidx, float2 (&pre)[LOADS]) {
    const int tile = gidx / nfull, q = gidx - tile * nfull;
    const int n = nrows - tile * TPX - wu * 32;
    if constexpr (ROWS) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        const int rid = rlw[(tile & 1) * 32 + lpx + 4 * i];      // (clamped ids: rows past the end re-read a valid row)
        pre[i] = *reinterpret_cast<const float2 *>(x + (int64_t)rid * d + q * KC + 2 * lf2);
      }
    } else {
      const float *tb = x + (crow0 + (int64_t)tile * TPX + wu * 32) * d + q * KC + 2 * lf2;   // wave-uniform + lane column
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        // rows past the end re-read a valid row (clamped address) and are never written back
        const int pxc = max(min(lpx + 4 * i, n - 1), -(tile * TPX + wu * 32));
        pre[i] = *reinterpret_cast<const float2 *>(tb + pxc * d);
      }
    }
  };
  // Streaming rows (not ROWS): the chunks are fetched strictly in order, four ahead of their use, so the address of a
  // load is (a wave-uniform base that moves with the column chunk) + (a per-lane byte offset that changes once per
  // tile).  load_chunk above recomputes tile, column and eight clamped row addresses for every chunk -- ~6 vector
  // instructions per load, a quarter of the vector work of the loss forward; this stream costs none per load and
  // ~40 per tile.
  uint32_t voff[LOADS];
  int pf_g = 0, pf_q = 0, pf_tile = 0;
  // (buffer loads: descriptor = the pass's rows, built from wave-uniform values made provably so; voffset = the
  //  lane's byte offset, soffset = the column chunk: no address arithmetic and no 64-bit address registers)
  const float *xbase = x + crow0 * (int64_t)d;
  const uint64_t xb = reinterpret_cast<uint64_t>(xbase);
  const uint32_t xb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb);
  const uint32_t xb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xb >> 32));
  const int xbytes = __builtin_amdgcn_readfirstlane((int)min((int64_t)max(nrows, 1) * d * 4, (int64_t)0x7FFFFFFF));
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void *>(((uint64_t)xb_hi << 32) | xb_lo), 0, xbytes, 0x00020000);
  auto set_stream_tile = [&](int tile) {
    const int n = nrows - tile * TPX - wu * 32;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int pxc = max(min(lpx + 4 * i, n - 1), -(tile * TPX + wu * 32));          // rows past the end: a valid row
      voff[i] = (uint32_t)((tile * TPX + wu * 32 + pxc) * d + 2 * lf2) * 4u;
    }
  };
  if constexpr (!ROWS) set_stream_tile(0);
  auto next_chunk = [&](float2 (&pre)[LOADS]) {
    const int soff = __builtin_amdgcn_readfirstlane(pf_q * KC * 4);
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, (int)voff[i], soff, 0);
      pre[i].x = __uint_as_float(v.x);
      pre[i].y = __uint_as_float(v.y);
    }
    if (pf_g < nsteps - 1) {                       // (the last chunk is re-read by the calls past the end)
      ++pf_g;
      if (++pf_q == nfull) { pf_q = 0; ++pf_tile; set_stream_tile(pf_tile); }
    }
  };
  auto store_chunk = [&](int buf, const float2 (&pre)[LOADS]) {
    uint16_t *hp = xw + buf * (2 * 32 * XSB);
    uint16_t *lp = hp + 32 * XSB;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int px = lpx + 4 * i;
      uint32_t hi, lo;
      if constexpr (XPRE) {
        hi = __float_as_uint(pre[i].x);
        lo = __float_as_uint(pre[i].y);
      } else {
        split2(pre[i].x, pre[i].y, hi, lo);
      }
      *reinterpret_cast<uint32_t *>(hp + px * XSB + 2 * lf2) = hi;
      *reinterpret_cast<uint32_t *>(lp + px * XSB + 2 * lf2) = lo;
    }
  };

  f32x16 acc[2];
  f32x16 acc2[F16S ? 2 : 1];                    // cross terms of the scaled fp16 split
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[m][r] = 0.0f;
        if constexpr (F16S) acc2[m][r] = 0.0f;
      }
  };
  auto kblock = [&](const bf16x8 &bh, const bf16x8 &bl, int col0) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(chs + (m * 32 + j) * RS + col0 + 8 * g);
      const bf16x8 al = *reinterpret_cast<const bf16x8 *>(cls + (m * 32 + j) * RS + col0 + 8 * g);
      if constexpr (F16S) {
        typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
        const h16x8 fah = __builtin_bit_cast(h16x8, ah), fal = __builtin_bit_cast(h16x8, al);
        const h16x8 fbh = __builtin_bit_cast(h16x8, bh), fbl = __builtin_bit_cast(h16x8, bl);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbh, acc[m], 0, 0, 0);
        acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, fbh, acc2[m], 0, 0, 0);
        acc2[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbl, acc2[m], 0, 0, 0);
      } else {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
      }
    }
  };
  auto compute_chunk = [&](int buf, int q) {
    const uint16_t *hp = xw + buf * (2 * 32 * XSB) + j * XSB + 8 * g;
    const uint16_t *lp = hp + 32 * XSB;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const bf16x8 bh = *reinterpret_cast<const bf16x8 *>(hp + kb * 16);
      const bf16x8 bl = *reinterpret_cast<const bf16x8 *>(lp + kb * 16);
      kblock(bh, bl, q * KC + kb * 16);
    }
  };
  // tail k-block kb of a tile: columns tcol0 + 16 kb + 8 g + 0..7 of row j, from global
  // (generic path: loaded where it is used, which drains the prefetch queue once per tile)
  auto tail_operands = [&](int tile, int kb, bf16x8 &bh, bf16x8 &bl) {
    const int n = nrows - tile * TPX - w * 32;
    const int jc = max(min(j, n - 1), -(tile * TPX + w * 32));
    const float *src = x + (crow0 + (int64_t)tile * TPX + w * 32 + jc) * d;
    if constexpr (ROWS) src = x + (int64_t)rlw[(tile & 1) * 32 + j] * d;
    const int c0 = tcol0 + 16 * kb + 8 * g;
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int ca = c0 + 2 * p;
      const float a = (j < n && ca < d) ? src[min(ca, d - 1)] : 0.0f;
      const float b = (j < n && ca + 1 < d) ? src[min(ca + 1, d - 1)] : 0.0f;
      split2(a, b, hw[p], lw[p]);
    }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 hv = {hw[0], hw[1], hw[2], hw[3]}, lv = {lw[0], lw[1], lw[2], lw[3]};
    bh = __builtin_bit_cast(bf16x8, hv);
    bl = __builtin_bit_cast(bf16x8, lv);
  };
  // two tail columns (d = C + 2, the shape that matters): one float2 per row, loaded a
  // whole tile ahead with the chunk loads -- the epilogue never waits on memory
  const bool two_tail = d - tcol0 == 2;
  auto load_tail2 = [&](int tile, float2 &v) {
    const int n = nrows - tile * TPX - w * 32;
    const int jc = max(min(j, n - 1), -(tile * TPX + w * 32));
    const float *src = x + (crow0 + (int64_t)tile * TPX + w * 32 + jc) * d + tcol0;
    if constexpr (ROWS) src = x + (int64_t)rlw[(tile & 1) * 32 + j] * d + tcol0;
    v = *reinterpret_cast<const float2 *>(src);
  };
  auto finish_tile = [&](int tile, const float2 &tv) {
    if (two_tail) {
      uint32_t hi, lo;
      split2(tv.x, tv.y, hi, lo);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 hv = {g == 0 ? hi : 0u, 0u, 0u, 0u}, lv = {g == 0 ? lo : 0u, 0u, 0u, 0u};
      kblock(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, lv), tcol0);
    } else {
      for (int kb = 0; kb < tblocks; ++kb) {
        bf16x8 bh, bl;
        tail_operands(tile, kb, bh, bl);
        kblock(bh, bl, tcol0 + 16 * kb);
      }
    }
    if constexpr (F16S) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = fmaf(acc2[m][r], 0.00048828125f, acc[m][r]);
    }
    epi(tile, acc);
  };

  // visible full wait: staging loads consumed under per-lane conditions stay "pending" in
  // the compiler's s_waitcnt model on the skipped paths, and it would then guard their
  // registers with vmcnt(0) inside the streaming loop (see score_tiles_f16.h)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                         // table planes visible to all waves
  zero_acc();
  // The caller guarantees nfull % 4 == 0 (split_shape_ok): four register sets
  // rotate, each loaded FOUR chunks (4 KiB per wave each) ahead of its use --
  // with the MFMA work this cheap the kernel is latency-bound unless ~100 KiB
  // per CU are in flight -- and a tile always ends on the last set, so there is
  // ONE epilogue site and the accumulators never move between code paths.
  static_assert(DEPTH == 2 || DEPTH == 4, "prefetch depth");
  if constexpr (XPRE && !ROWS) {
    // ---- operands straight from the operand-order image: four register sets of one chunk (two k-blocks) each
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (nsteps <= 0) return;
    const int nkb = d / 16;
    const int64_t img0 = (crow0 / 32) * (int64_t)nkb * 512;          // dwords: (32-row tile, k-block) -> 512
    const uint64_t ib = reinterpret_cast<uint64_t>(x + img0);
    const uint32_t ib_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ib);
    const uint32_t ib_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ib >> 32));
    const int ibytes = __builtin_amdgcn_readfirstlane((nrows + 31) / 32 * nkb * 2048);
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((uint64_t)ib_hi << 32) | ib_lo), 0, ibytes, 0x00020000);
    const int lane16 = lane * 16;
    int sg = 0, sq = 0, stile = 0;                                   // the fetch stream (sequential, clamped at the end)
    auto fetchx = [&](u32x4 (&h)[2], u32x4 (&l)[2]) {
      const int soff = __builtin_amdgcn_readfirstlane(((stile * (TPX / 32) + wu) * nkb + 2 * sq) * 2048);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {                               // (tiles past the end of the pass read zeros)
        h[kb] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, lane16, soff + kb * 2048, 0);
        l[kb] = __builtin_amdgcn_raw_buffer_load_b128(irsrc, lane16 + 1024, soff + kb * 2048, 0);
      }
      if (sg < nsteps - 1) {
        ++sg;
        if (++sq == nfull) { sq = 0; ++stile; }
      }
    };
    auto computex = [&](const u32x4 (&h)[2], const u32x4 (&l)[2], int q) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        kblock(__builtin_bit_cast(bf16x8, h[kb]), __builtin_bit_cast(bf16x8, l[kb]), q * KC + kb * 16);
    };
    u32x4 hA[2], lA[2], hB[2], lB[2], hC[2], lC[2], hD[2], lD[2];
    fetchx(hA, lA); fetchx(hB, lB); fetchx(hC, lC); fetchx(hD, lD);
    const float2 notail = {0.0f, 0.0f};
    if constexpr (NFULLC > 0 && F16S && requires { typename Epi::State; }) {
      static_assert(NFULLC == 4 || NFULLC == 8 || NFULLC == 16, "chunks per tile");
      typedef typename Epi::State EState;
      EState stp, stn;                      // rows of the tile whose scores are in `prev` / of the tile being scored
      epi.begin(0, stp);
      stp.valid = false; stp.pmax = 0;      // (nothing to fold before the first tile: scores of zero, nothing stored)
      f32x16 prev[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[m][r] = 0.0f;
      auto xstep = [&]<int Q>(u32x4 (&h)[2], u32x4 (&l)[2]) {
        computex(h, l, Q);
        constexpr int P0 = Q * 8 / NFULLC, P1 = (Q + 1) * 8 / NFULLC;
        if constexpr (P1 > P0) epi.template piece<P0 / 4, 4 * (P0 % 4), 4>(stp, prev[P0 / 4]);
        if constexpr (P1 > P0 + 1) epi.template piece<(P0 + 1) / 4, 4 * ((P0 + 1) % 4), 4>(stp, prev[(P0 + 1) / 4]);
        // one matrix instruction, then the vector work that fits behind it
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        fetchx(h, l);
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int tile = 0; tile < ntile; ++tile) {
        epi.begin(tile, stn);
        xstep.template operator()<0>(hA, lA);
        xstep.template operator()<1>(hB, lB);
        xstep.template operator()<2>(hC, lC);
        xstep.template operator()<3>(hD, lD);
        if constexpr (NFULLC >= 8) {
          xstep.template operator()<4>(hA, lA);
          xstep.template operator()<5>(hB, lB);
          xstep.template operator()<6>(hC, lC);
          xstep.template operator()<7>(hD, lD);
        }
        if constexpr (NFULLC >= 16) {
          xstep.template operator()<8>(hA, lA);
          xstep.template operator()<9>(hB, lB);
          xstep.template operator()<10>(hC, lC);
          xstep.template operator()<11>(hD, lD);
          xstep.template operator()<12>(hA, lA);
          xstep.template operator()<13>(hB, lB);
          xstep.template operator()<14>(hC, lC);
          xstep.template operator()<15>(hD, lD);
        }
        epi.end(stp);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[m][r] = fmaf(acc2[m][r], 0.00048828125f, acc[m][r]);
        stp = stn;
        zero_acc();
      }
      // the last tile's scores: nothing left to hide behind
      epi.template piece<0, 0, 4>(stp, prev[0]);
      epi.template piece<0, 4, 4>(stp, prev[0]);
      __builtin_amdgcn_sched_barrier(0);
      epi.template piece<0, 8, 4>(stp, prev[0]);
      epi.template piece<0, 12, 4>(stp, prev[0]);
      __builtin_amdgcn_sched_barrier(0);
      epi.template piece<1, 0, 4>(stp, prev[1]);
      epi.template piece<1, 4, 4>(stp, prev[1]);
      __builtin_amdgcn_sched_barrier(0);
      epi.template piece<1, 8, 4>(stp, prev[1]);
      epi.template piece<1, 12, 4>(stp, prev[1]);
      epi.end(stp);
      return;
    }
#define HSGK_SPLIT_XSTEP(H, L, QQ)                                            \
    computex(H, L, QQ);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                        \
    fetchx(H, L);                                                             \
    __builtin_amdgcn_sched_barrier(0);
    for (int tile = 0; tile < ntile; ++tile) {
      for (int q = 0; q < nfull; q += 4) {
        HSGK_SPLIT_XSTEP(hA, lA, q)
        HSGK_SPLIT_XSTEP(hB, lB, q + 1)
        HSGK_SPLIT_XSTEP(hC, lC, q + 2)
        HSGK_SPLIT_XSTEP(hD, lD, q + 3)
      }
      finish_tile(tile, notail);
      zero_acc();
    }
#undef HSGK_SPLIT_XSTEP
    return;
  }
  float2 preA[LOADS], preB[LOADS], preC[DEPTH == 4 ? LOADS : 1], preD[DEPTH == 4 ? LOADS : 1];
  if (nsteps <= 0) return;
  auto fetch = [&](int gi, float2 (&pre)[LOADS]) {
    if constexpr (ROWS) load_chunk(gi, pre); else next_chunk(pre);
  };
  fetch(0, preA);
  fetch(min(1, nsteps - 1), preB);
  if constexpr (DEPTH == 4) {
    fetch(min(2, nsteps - 1), preC);
    fetch(min(3, nsteps - 1), preD);
  }
  float2 tail_cur = {0.0f, 0.0f}, tail_next = {0.0f, 0.0f};
  if (two_tail) load_tail2(0, tail_cur);
  int gidx = 0;
#define HSGK_SPLIT_STEP(BUF, PRE, STEP, QQ)                                   \
  store_chunk(BUF, PRE);                                                      \
  __builtin_amdgcn_sched_barrier(0);                                          \
  fetch(min(gidx + (STEP) + DEPTH, nsteps - 1), PRE);                         \
  __builtin_amdgcn_sched_barrier(0);                                          \
  compute_chunk(BUF, QQ);                                                     \
  __builtin_amdgcn_sched_barrier(0);
  for (int tile = 0; tile < ntile; ++tile) {
    int32_t rid_next = 0;
    if constexpr (ROWS) rid_next = row_id(tile + 2);           // into the slot this tile is about to free
    if (two_tail) load_tail2(min(tile + 1, ntile - 1), tail_next);
    for (int q = 0; q < nfull; q += 4, gidx += 4) {
      if constexpr (DEPTH == 4) {
        HSGK_SPLIT_STEP(0, preA, 0, q)
        HSGK_SPLIT_STEP(1, preB, 1, q + 1)
        HSGK_SPLIT_STEP(0, preC, 2, q + 2)
        HSGK_SPLIT_STEP(1, preD, 3, q + 3)
      } else {
        HSGK_SPLIT_STEP(0, preA, 0, q)
        HSGK_SPLIT_STEP(1, preB, 1, q + 1)
        HSGK_SPLIT_STEP(0, preA, 2, q + 2)
        HSGK_SPLIT_STEP(1, preB, 3, q + 3)
      }
    }
    finish_tile(tile, tail_cur);
    tail_cur = tail_next;
    if constexpr (ROWS) rlw[(tile & 1) * 32 + j] = rid_next;
    zero_acc();
  }
#undef HSGK_SPLIT_STEP
}

// shapes the split engine accepts (number of 32-column chunks divisible by 4)
__host__ __device__ inline bool split_shape_ok(int d) {
  return (d & 1) == 0 && d >= 128 && ((d / 32) & 3) == 0;
}

}  // namespace hsgk
