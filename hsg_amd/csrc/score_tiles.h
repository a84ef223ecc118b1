// score_tiles.h -- the fp32-MFMA "rows x table" streaming engine shared by the
// k-means E-step (kmeans.hip: argmax epilogue) and the pixel-to-segment
// contrastive loss (loss.hip: exp / label-sum epilogue).
//
// One workgroup (NW waves) scores `nrows` consecutive rows of x[., d] against a
// block of up to KB (= 32 or 64) table rows.  The table block is staged ONCE in
// LDS with an odd row stride (bank-conflict-free A-operand reads).  Each wave
// owns 32-row tiles: rows sit on the MFMA N dimension, table rows on M, so a
// lane ends with all KB scores of ONE row split between lanes l and l^32.
//
// Rows stream through a WAVE-PRIVATE, double-buffered LDS window in column
// chunks of KC floats (padded to KC+1 per row): a wave parks and reads back
// only its own rows, so the column loop needs no workgroup barrier and the two
// waves of a SIMD overlap each other's staging with MFMAs.  Global loads run
// two chunks ahead of their use; LDS operand reads one k-step ahead.  Columns
// past the last full chunk (d mod KC: the 2 location channels when C is a
// multiple of KC) are fed straight from global memory, one float per lane per
// k-step.
//
// Every score is the canonical C1 chain: v_mfma_f32_32x32x2_f32 accumulates
// fma(a_k1, b_k1, fma(a_k0, b_k0, c)) in ascending k from +0.0f.
// Operand map: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; D[i][j] at lane
// (j, h = l>>5), register r <-> i = (r&3) + 8*(r>>2) + 4*h.
//
// Epilogue contract: epi(tile, acc) is called by every wave once per finished
// 32-row tile with `tile` = index of the NW*32-row tile inside the workgroup's
// row range; lane (j, h) holds acc[m][r] = score(table row m*32 + (r&3) +
// 8*(r>>2) + 4*h, x row tile*NW*32 + w*32 + j).
#pragma once
#include "common.h"

namespace hsgk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KB, int NW, int KC>
__host__ __device__ constexpr size_t score_tiles_lds_bytes(int d) {
  return ((size_t)KB * ((((d + 1) & ~1)) | 1) + (size_t)2 * NW * 32 * (KC + 1)) * 4;
}

// rowlist (nullable): when given, logical row r of the range is the physical
// row crow0 + rowlist[r] (used by the exact re-scoring pass of the bf16-split
// E-step, which only visits the queued rows of a chunk).  RL = int32_t with
// crow0 = 0: a list of absolute row ids (the dense pass over the rows that ask
// for all K centroids, kmeans.hip: assign_hard_rows_kernel).
template <int KB, int NW, int KC, bool EVEN_D, class Epi, class RL = uint16_t>
__device__ inline void score_tiles(const float *__restrict__ x, int d,
                                   const float *__restrict__ table, int kvalid,
                                   int64_t crow0, int nrows, float *lds, Epi &epi,
                                   const RL *__restrict__ rowlist = nullptr) {
  constexpr int NT = NW * 64;
  constexpr int TPX = NW * 32;
  constexpr int XS = KC + 1;
  constexpr int MB = KB / 32;
  constexpr int F2_PER_ROW = KC / 2;
  constexpr int LOADS = (32 * F2_PER_ROW) / 64;       // float2 per lane per chunk
  constexpr int ROWS_PER_LOAD = 64 / F2_PER_ROW;
  constexpr int MAXT = 4;                             // preloaded tail k-steps
  static_assert((32 * F2_PER_ROW) % 64 == 0 && 64 % F2_PER_ROW == 0, "staging must divide evenly");

  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  float *cent_s = lds;                     // [KB][DP]
  float *xs = lds + KB * DP;               // [NW][2][32][XS]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;

  // ---- stage the table block once (rows >= kvalid and the pad column are
  //      zero); batches of 8 independent 8-byte loads per thread
  {
    const float *src = table;
    if constexpr (EVEN_D) {
      const int half = d >> 1;
      const int total = KB * half;                     // float2 elements
      for (int f0 = 0; f0 < total; f0 += NT * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + tid + NT * u;
          const int k = f / half;
          v[u] = (f < total && k < kvalid) ? *reinterpret_cast<const float2 *>(src + 2 * (int64_t)f)
                                           : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + tid + NT * u;
          if (f < total) {
            const int k = f / half, col = 2 * (f - k * half);
            cent_s[k * DP + col] = v[u].x;
            cent_s[k * DP + col + 1] = v[u].y;
          }
        }
      }
    } else {
      for (int i = w; i < KB; i += NW)
        for (int dd = lane; dd < dpad; dd += 64)
          cent_s[i * DP + dd] = (i < kvalid && dd < d) ? src[(int64_t)i * d + dd] : 0.0f;
    }
  }

  const int nfull = d / KC;                 // full column chunks, staged via LDS
  const int tcol0 = nfull * KC;
  const int tsteps = (dpad - tcol0) / 2;    // tail k-steps, fed from global
  const int ntile = (nrows + TPX - 1) / TPX;
  const int nsteps = ntile * nfull;

  float *xw = xs + w * (2 * 32 * XS);
  const int lpx = lane / F2_PER_ROW, lf2 = lane % F2_PER_ROW;

  // issue the LOADS 8-byte loads of global chunk g (tile g / nfull, chunk g % nfull)
  auto load_chunk = [&](int g, float2 (&pre)[LOADS]) {
    const int tile = g / nfull, q = g - tile * nfull;
    const int n = nrows - tile * TPX - w * 32;       // rows this wave owns in the tile
    const float *tb = x + (crow0 + (int64_t)tile * TPX + w * 32) * d + q * KC;   // wave-uniform
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int px = lpx + ROWS_PER_LOAD * i;
      const int pxc = max(min(px, n - 1), -(tile * TPX + w * 32));   // stay inside the chunk
      const float *src = tb + pxc * d + 2 * lf2;
      if (rowlist)
        src = x + (crow0 + rowlist[tile * TPX + w * 32 + pxc]) * d + q * KC + 2 * lf2;
      float2 v;
      if constexpr (EVEN_D) {
        v = *reinterpret_cast<const float2 *>(src);
      } else {
        v.x = src[0];
        v.y = src[1];                                 // q*KC + 2*lf2 + 1 < tcol0 <= d
      }
      pre[i] = v;       // rows past the end re-read the last valid row and are never stored
    }
  };
  auto store_chunk = [&](int buf, const float2 (&pre)[LOADS]) {
    float *dst = xw + buf * (32 * XS);
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int px = lpx + ROWS_PER_LOAD * i;
      dst[px * XS + 2 * lf2] = pre[i].x;
      dst[px * XS + 2 * lf2 + 1] = pre[i].y;
    }
  };

  f32x16 acc[MB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
  };

  // tail operand of (tile, k-step st): column tcol0 + 2*st + h of row j
  auto load_tail = [&](int tile, int st) -> float {
    const int n = nrows - tile * TPX - w * 32;
    const int col = tcol0 + 2 * st + h;
    const int jc = max(min(j, n - 1), -(tile * TPX + w * 32));
    const int64_t lr = (int64_t)tile * TPX + w * 32 + jc;
    const float v = x[(crow0 + (rowlist ? (int64_t)rowlist[lr] : lr)) * d + min(col, d - 1)];
    return (j < n && col < d) ? v : 0.0f;
  };

  // all MFMAs of one staged chunk; LDS operands are read one k-step ahead
  auto compute_chunk = [&](int buf, int q) {
    const float *xb = xw + buf * (32 * XS) + j * XS + h;
    const float *cb = cent_s + j * DP + q * KC + h;
    float bc = xb[0], ac[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) ac[m] = cb[m * 32 * DP];
#pragma unroll
    for (int st = 0; st < KC / 2; ++st) {
      float bn = 0.0f, an[MB];
      if (st + 1 < KC / 2) {
        bn = xb[2 * st + 2];
#pragma unroll
        for (int m = 0; m < MB; ++m) an[m] = cb[m * 32 * DP + 2 * st + 2];
      }
      __builtin_amdgcn_sched_barrier(0);     // next operands are in flight ...
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m], bc, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);     // ... while these MFMAs occupy the pipe
      bc = bn;
#pragma unroll
      for (int m = 0; m < MB; ++m) ac[m] = an[m];
    }
  };

  // tail k-steps, then hand the finished tile to the epilogue
  auto finish_tile = [&](int tile, const float (&bt)[MAXT]) {
    const float *cb = cent_s + j * DP + tcol0 + h;
    for (int st = 0; st < tsteps; ++st) {
      const float bv = st < MAXT ? bt[st < MAXT ? st : 0] : load_tail(tile, st);
#pragma unroll
      for (int m = 0; m < MB; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[m * 32 * DP + 2 * st], bv, acc[m], 0, 0, 0);
    }
    epi(tile, acc);
  };

  // visible full wait: staging loads consumed under per-lane conditions stay "pending" in
  // the compiler's s_waitcnt model on the skipped paths, and it would then guard their
  // registers with vmcnt(0) inside the streaming loop (see score_tiles_f16.h)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();                         // table block visible to all waves
  zero_acc();

  float bt[MAXT];
  auto preload_tail = [&](int tile) {
#pragma unroll
    for (int st = 0; st < MAXT; ++st) bt[st] = st < tsteps ? load_tail(tile, st) : 0.0f;
  };

  if (nfull == 0) {                         // rows shorter than one chunk
    for (int tile = 0; tile < ntile; ++tile) {
      preload_tail(tile);
      finish_tile(tile, bt);
      zero_acc();
    }
    return;
  }

  // software pipeline over global chunk index g = tile * nfull + q: register
  // sets A / B alternate, each loaded two chunks ahead of its use.
  float2 preA[LOADS], preB[LOADS];
  // (loads are unconditional with clamped indices: a conditional load in the loop makes
  //  the compiler's s_waitcnt insertion drain the whole queue at the join)
  load_chunk(0, preA);
  load_chunk(min(1, nsteps - 1), preB);
  preload_tail(0);
  int tile = 0, q = 0;
  for (int g = 0; g < nsteps; g += 2) {
    // ---- even step: set A, LDS buffer 0
    store_chunk(0, preA);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(min(g + 2, nsteps - 1), preA);
    __builtin_amdgcn_sched_barrier(0);
    compute_chunk(0, q);
    if (++q == nfull) {
      finish_tile(tile, bt);
      zero_acc();
      q = 0;
      ++tile;
      if (tile < ntile) preload_tail(tile);
    }
    if (g + 1 >= nsteps) break;
    // ---- odd step: set B, LDS buffer 1
    store_chunk(1, preB);
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(min(g + 3, nsteps - 1), preB);
    __builtin_amdgcn_sched_barrier(0);
    compute_chunk(1, q);
    if (++q == nfull) {
      finish_tile(tile, bt);
      zero_acc();
      q = 0;
      ++tile;
      if (tile < ntile) preload_tail(tile);
    }
  }
}

}  // namespace hsgk
