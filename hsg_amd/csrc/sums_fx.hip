// sums_fx.hip -- M-step of the k-means loop with EXACT segment sums (canonical order C2x).
//
// Every element x of a unit-norm row is converted to the fixed-point integer
// q = rint(x * 2^40) (exact for |x| >= 2^-16, absolute error <= 2^-41 below) and the
// segment sums are 64-bit integer sums of the q, converted to fp32 with ONE rounding at
// the end.  Integer addition is associative and commutative, so
//   * the result does not depend on any summation order (LDS / global atomics are fine),
//   * it can be UPDATED: when a row moves from cluster a to cluster b, sums[a] -= q(row),
//     sums[b] += q(row) gives exactly the sums a full pass over the new labels would.
// A Lloyd iteration therefore only reads the rows whose label changed in the previous
// E-step (28 % -> 2.6 % of the rows per iteration on the i.i.d. bench input, a few per mille
// on converging data) instead of streaming all fp32 rows again; the first iteration adds
// every row once (prev label = -1).  Range: |x| <= 1 and <= 2^22 rows per segment keep
// every sum below 2^63 (segment_by_kmeans: unit rows, one image per table).
//
// update_sums_kernel: one workgroup (4 waves) per 512 rows (a quarter of a chunk: any
// partition is valid, the sums do not depend on it).  The changed rows are compacted into
// an LDS list; the clusters they touch (old or new label; a 512-row strip of an image meets
// one row of grid cells) get one of S slots of an int64 [S][d] LDS table (28 KiB for S = 12,
// d = 258: five workgroups per CU hide each other's latencies; a big [K][d] table would
// allow one); each changed row is loaded once (16-byte row loads, 4 rows per wave in
// flight, double buffered), converted, and added to / subtracted from its slots with
// ds_add_u64 (measured 2.5x faster than a plain 64-bit LDS read-add-write and 20x faster
// than ds_add_f32, tools/probes/lds_atomics.hip); the slots are flushed to the image's
// table with global 64-bit atomics.  More than S touched clusters: further rounds over the
// same list (the rows are re-read).
#include "common.h"

namespace hsgk {

constexpr int kFxRows = 512;                         // rows per workgroup

template <int NW, int UNROLL, int S>
__global__ __launch_bounds__(NW * 64) void update_sums_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ prev,
    const int32_t *__restrict__ cur, const int64_t *__restrict__ chunk_row0,
    const int32_t *__restrict__ chunk_rows, const int32_t *__restrict__ chunk_img, int K,
    unsigned long long *__restrict__ sumq, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned long long *tab = reinterpret_cast<unsigned long long *>(lds_raw);          // [S][min(d, 512)]
  uint32_t *list = reinterpret_cast<uint32_t *>(tab + (size_t)S * min(d, 515) + (d & 1));   // [kFxRows] + 2 (zeroing tail)
  int *wcount = reinterpret_cast<int *>(list + kFxRows + 2);                          // [NW], [NW] = touched clusters
  uint16_t *slot = reinterpret_cast<uint16_t *>(wcount + NW + 1);                     // [K] rank among the touched clusters (0xFFFF = untouched)
  constexpr int PARTS = HSGK_CHUNK / kFxRows;
  const int c = blockIdx.x / PARTS, part = blockIdx.x - c * PARTS;
  if (c >= meta->n_chunks) return;
  const int n = min(chunk_rows[c] - part * kFxRows, kFxRows);
  if (n <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t row0 = chunk_row0[c] + (int64_t)part * kFxRows;
  // ---- changed rows of this strip
  constexpr int PER = kFxRows / (NW * 64);
  int pl[PER], cl[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = (w * PER + i) * 64 + lane;                   // wave-contiguous rows
    const int rr = min(r, n - 1);
    pl[i] = get_label(prev, row0 + rr);
    cl[i] = get_label(cur, row0 + rr);
    if (r >= n) pl[i] = cl[i];                                 // past the end: unchanged
  }
  for (int i = tid; i < K; i += NW * 64) slot[i] = 0xFFFF;
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) cnt += __popcll(__ballot(pl[i] != cl[i]));
  if (lane == 0) wcount[w] = cnt;
  __syncthreads();
  int lbeg = 0, total = 0;
  for (int i = 0; i < NW; ++i) {
    if (i < w) lbeg += wcount[i];
    total += wcount[i];
  }
  if (total == 0) return;                                       // (uniform)
  {
    int pos = lbeg;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool ch = pl[i] != cl[i];
      const unsigned long long m = __ballot(ch);
      if (ch) {
        // row (9 bits) | new label + 1 (11 bits) | old label + 1 (11 bits, 0 = not added yet)
        list[pos + __popcll(m & ((1ull << lane) - 1ull))] =
            ((uint32_t)((w * PER + i) * 64 + lane) << 22) | ((uint32_t)(cl[i] + 1) << 11) | (uint32_t)(pl[i] + 1);
        slot[cl[i]] = 1;                                         // (benign same-value races)
        if (pl[i] >= 0) slot[pl[i]] = 1;
      }
      pos += __popcll(m);
    }
  }
  __syncthreads();
  // rank the touched clusters (ascending cluster id); thread per cluster
  {
    int my[(1024 + NW * 64 - 1) / (NW * 64)];
    int q = 0;
    for (int k = tid; k < K; k += NW * 64, ++q) {
      int rank = -1;
      if (slot[k] != 0xFFFF) {
        rank = 0;
        for (int o = 0; o < k; ++o) rank += slot[o] != 0xFFFF ? 1 : 0;
      }
      my[q] = rank;
    }
    int nt = 0;
    if (tid == 0)
      for (int o = 0; o < K; ++o) nt += slot[o] != 0xFFFF ? 1 : 0;
    __syncthreads();
    q = 0;
    for (int k = tid; k < K; k += NW * 64, ++q) slot[k] = my[q] < 0 ? 0xFFFF : (uint16_t)my[q];
    if (tid == 0) wcount[NW] = nt;
  }
  __syncthreads();
  const int ntouched = wcount[NW];
  typedef float gvec_t __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const float *xr = x + row0 * d;
  const int dws = min(d, 515);                                  // table row stride: one column window
  const int nmine = (total - w + NW - 1) / NW;
  auto entry = [&](int i) { return list[min(w + i * NW, total - 1)]; };
  unsigned long long *gq = sumq + (int64_t)chunk_img[c] * K * d;
  for (int base = 0; base < ntouched; base += S)                 // (one round unless > S clusters are touched)
    for (int cw = 0, dw = 0; cw < d; cw += dw) {                 // column windows of 512 (one for d <= 515)
      dw = (d - cw <= 515) ? d - cw : 512;                       // the last window takes the d mod 4 tail along
      const int nq = dw / 4, tail0 = nq * 4;                     // quads, then dw mod 4 scalar columns
      // ---- zero the slots of this round
      {
        const int tot2 = (min(S, ntouched - base) * dws + 1) / 2;
        u64x2 *t2 = reinterpret_cast<u64x2 *>(tab);
        for (int i = tid; i < tot2; i += NW * 64) t2[i] = u64x2{0ull, 0ull};
      }
      __syncthreads();
      // ---- every wave takes entries w, w + NW, ...: load the row's window once, add / subtract
      auto issue = [&](int i0, gvec_t (&v)[UNROLL][2], float (&t)[UNROLL]) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int r = (int)(entry(i0 + u) >> 22);
          const float *src = xr + (int64_t)r * d + cw;
          v[u][0] = v[u][1] = gvec_t{0.f, 0.f, 0.f, 0.f};
          if (nq > 0) {                                            // (uniform; rows shorter than one quad: scalar columns only)
            v[u][0] = *reinterpret_cast<const gvec_t *>(src + 4 * min(lane, nq - 1));
            v[u][1] = *reinterpret_cast<const gvec_t *>(src + 4 * min(lane + 64, nq - 1));
          }
          t[u] = src[min(tail0 + lane, dw - 1)];
        }
      };
      auto fold = [&](int i0, const gvec_t (&v)[UNROLL][2], const float (&t)[UNROLL]) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          if (i0 + u < nmine) {
            const uint32_t e = entry(i0 + u);
            const int labs[2] = {(int)((e >> 11) & 2047u) - 1, (int)(e & 2047u) - 1};
            int sl[2];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              const int s0 = labs[side] >= 0 ? (int)slot[labs[side]] - base : -1;      // (listed labels are touched)
              sl[side] = (s0 >= 0 && s0 < S) ? s0 : -1;
            }
            if (sl[0] < 0 && sl[1] < 0) continue;                  // (uniform per entry)
            long long q[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int j = 0; j < 4; ++j) q[h][j] = to_fixed(v[u][h][j]);
            const long long qt = to_fixed(t[u]);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              if (sl[side] < 0) continue;
              unsigned long long *rowp = tab + (size_t)sl[side] * dws;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int q4 = lane + 64 * h;
                if (q4 < nq) {
#pragma unroll
                  for (int j = 0; j < 4; ++j)
                    atomicAdd(rowp + 4 * q4 + j, (unsigned long long)(side ? -q[h][j] : q[h][j]));
                }
              }
              if (tail0 + lane < dw) atomicAdd(rowp + tail0 + lane, (unsigned long long)(side ? -qt : qt));
            }
          }
        }
      };
      gvec_t va[UNROLL][2], vb[UNROLL][2];
      float ta[UNROLL], tb[UNROLL];
      issue(0, va, ta);
      for (int i0 = 0; i0 < nmine; i0 += 2 * UNROLL) {
        issue(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        fold(i0, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        issue(i0 + 2 * UNROLL, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        fold(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      // ---- flush the slots of this round / window into the image's table
      for (int k = w; k < K; k += NW) {
        const int s0 = slot[k] == 0xFFFF ? -1 : (int)slot[k] - base;
        if (s0 >= 0 && s0 < S)
          for (int i = lane; i < dw; i += 64) {
            const unsigned long long v = tab[(size_t)s0 * dws + i];
            if (v) atomicAdd(gq + (size_t)k * d + cw + i, v);
          }
      }
      __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Persistent variant for K * d * 8 <= ~132 KiB (K = 64, d = 258): the whole [K][d] int64
// table of ONE image lives in LDS, one workgroup (8 waves) per CU owns a contiguous range
// of chunks (mostly one image), and every WAVE works through its own 256-row strips with
// no workgroup barrier: labels of the strip -> wave-private list of changed rows -> rows
// loaded (8 in flight, double buffered), converted, ds_add_u64 into the shared table.  The
// table is flushed with global atomics only when the workgroup's image changes -- about 5 M
// global atomics per launch instead of one flush per chunk (17 M), which is what bounded the
// late iterations where only a few per cent of the rows move.
// The LDS atomic rate bounds the kernel (tripling the atomics multiplies every launch by 2.4;
// a ds_add_u64 costs ~9 cycles whatever the number of active lanes), hence: the d mod 4 tail
// columns of a whole batch of rows share one atomic per side, and consecutive changed rows
// of one cluster add up in registers first (run-length carries; pixels are in raster order).
// NV: 16-byte vectors per lane and row (1: d <= 259, 2: d <= 515)
// elimination builds (tools/probes/ab_cfg.sh with ab_libs built by make EXTRA=-DHSGK_FX_ELIM=n): 1 no LDS atomics,
// 2 no row loads, 3 neither -- WRONG sums, timing only
#ifndef HSGK_FX_ELIM
#define HSGK_FX_ELIM 0
#endif
#if HSGK_FX_ELIM & 1
#define HSGK_FX_ATOMIC(p, v) do { if ((v) == 0x123456789abcull) atomicAdd(p, v); } while (0)
#else
#define HSGK_FX_ATOMIC(p, v) atomicAdd(p, v)
#endif
#if HSGK_FX_ELIM & 2
#define HSGK_FX_LOADV(p) (gvec_t{(float)((uintptr_t)(p) & 1023) * 1e-4f, 0.25f, -0.125f, 0.0625f})
#define HSGK_FX_LOADT(p) ((float)((uintptr_t)(p) & 255) * 1e-3f)
#else
#define HSGK_FX_LOADV(p) (*reinterpret_cast<const gvec_t *>(p))
#define HSGK_FX_LOADT(p) (*(p))
#endif
#ifdef HSGK_FX_TIMING
__device__ unsigned long long g_fx_ts[8];
#define HSGK_FXT(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_fx_ts[i], now_ - fx_ts); fx_ts = now_; } while (0)
#endif
// DC > 0: the row length as a compile-time constant (258: the headline shape) -- strides, quad counts and the
// lane tests fold away (the loop body shrinks by about a fifth); 0: from the argument
template <int NW, int UNROLL, int NV, int DC = 0>
__global__ __launch_bounds__(NW * 64) void update_sums_persistent_kernel(
    const float *__restrict__ x, int d_arg, const int32_t *__restrict__ prev,
    const int32_t *__restrict__ cur, const int64_t *__restrict__ chunk_row0,
    const int32_t *__restrict__ chunk_rows, const int32_t *__restrict__ chunk_img, int K, int P,
    unsigned long long *__restrict__ sumq, const hsgk_segkm_meta *__restrict__ meta) {
    const int d = DC > 0 ? DC : d_arg;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
#ifdef HSGK_FX_TIMING
  unsigned long long fx_ts = __builtin_readcyclecounter();
#endif
  // P workgroups share one chunk range, each owning the clusters [k0, k0 + kn) (P = 1 when the
  // whole table fits LDS): a workgroup reads the rows that leave or join ITS clusters
  const int KP = (K + P - 1) / P;
  const int part_k = (int)(blockIdx.x % (unsigned)P), range = (int)(blockIdx.x / (unsigned)P);
  const int nranges = (int)(gridDim.x / (unsigned)P);
  const int k0 = part_k * KP, kn = min(KP, K - k0);
  unsigned long long *tab = reinterpret_cast<unsigned long long *>(lds_raw);          // [KP][d]
  uint32_t *lists = reinterpret_cast<uint32_t *>(tab + (size_t)KP * d + 2);           // [NW][256]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t *list = lists + w * 256;
  const int nc = (int)meta->n_chunks;
  const int c_begin = (int)(((int64_t)range * nc) / nranges);
  const int c_end = (int)(((int64_t)(range + 1) * nc) / nranges);
  if (kn <= 0) return;
  typedef float gvec_t __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const int nq = d / 4, tail0 = nq * 4;
  // the d mod 4 tail columns of a whole batch of UNROLL rows go through ONE load and one
  // ds_add_u64 per side (lane = (row of the batch, tail column)): an LDS atomic costs the same
  // ~9 cycles with 2 active lanes as with 64, and the LDS atomic rate is what bounds this kernel
  const int tw = d - tail0;
  const int tu = lane / max(tw, 1), tc = lane - tu * max(tw, 1);
  const bool tact = tw > 0 && tu < UNROLL;
#ifndef HSGK_FX_STRIP
#define HSGK_FX_STRIP 256
#endif
  constexpr int STRIP = HSGK_FX_STRIP, SPC = HSGK_CHUNK / STRIP, LPL = STRIP / 64;          // strips per chunk, labels per lane
  int c = c_begin;
  while (c < c_end) {
    // ---- the run of chunks [c, ce) of one image
    const int b = chunk_img[c];
    int ce = c + 1;
    while (ce < c_end && chunk_img[ce] == b) ++ce;
    {
      const int tot2 = (kn * d + 1) / 2;
      u64x2 *t2 = reinterpret_cast<u64x2 *>(tab);
      for (int i = tid; i < tot2; i += NW * 64) t2[i] = u64x2{0ull, 0ull};
    }
    __syncthreads();
    // Run-length carries: consecutive changed rows that leave (side 1) or join (side 0) the
    // same cluster add up in registers and reach the LDS table with one set of atomics per
    // run -- pixels are in raster order, so on images (and for the seed-grid labels of the
    // first update) the runs are long; the LDS atomic rate is what bounds this kernel.
    int clab[2] = {-1, -1};
    long long cq[2][NV][4];
#pragma unroll
    for (int sd = 0; sd < 2; ++sd)
#pragma unroll
      for (int h = 0; h < NV; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) cq[sd][h][j] = 0;
    // LDS layout of a table row: the four columns 4 q + j of quad q = lane + 64 h sit at
    // 256 h + j * nh + lane (nh = quads of that 64-quad block), so that one ds_add_u64 -- fixed j,
    // all lanes -- touches CONSECUTIVE 8-byte words (two conflict-free LDS passes) instead of
    // words 32 bytes apart (eight lanes per bank pair: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    // was 0.69 with the natural layout).  The tail columns keep their place; the flush undoes it.
    auto flush_run = [&](int side) {
      if (clab[side] >= 0) {
        unsigned long long *rowp = tab + (size_t)clab[side] * d;
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          const int q4 = lane + 64 * h;
          const int nh = min(64, nq - 64 * h);
          if (q4 < nq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) HSGK_FX_ATOMIC(rowp + 256 * h + j * nh + lane, (unsigned long long)cq[side][h][j]);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < NV; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) cq[side][h][j] = 0;
    };
    // ---- strips of the run, dealt to the waves round robin; no barrier inside
    const int nstrips = (ce - c) * SPC;
    for (int st = w; st < nstrips; st += NW) {
#ifdef HSGK_FX_TIMING
      fx_ts = __builtin_readcyclecounter();
#endif
      const int cc = c + st / SPC, part = st - (st / SPC) * SPC;
      const int n = min(chunk_rows[cc] - part * STRIP, STRIP);
      if (n <= 0) continue;
      const int64_t row0 = chunk_row0[cc] + (int64_t)part * STRIP;
      int pl[LPL], cl[LPL];
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        const int r = 64 * i + lane;
        const int rr = min(r, n - 1);
        pl[i] = get_label(prev, row0 + rr);
        cl[i] = get_label(cur, row0 + rr);
        if (r >= n) pl[i] = cl[i];
      }
      int total = 0;
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        // labels local to this workgroup's clusters, + 1 (0: not one of them / not added yet)
        const uint32_t ln = (uint32_t)(cl[i] - k0) < (uint32_t)kn ? (uint32_t)(cl[i] - k0 + 1) : 0u;
        const uint32_t lo = (uint32_t)(pl[i] - k0) < (uint32_t)kn ? (uint32_t)(pl[i] - k0 + 1) : 0u;
        const bool ch = pl[i] != cl[i] && (ln | lo) != 0u;
        const unsigned long long m = __ballot(ch);
        if (ch)   // row (8 bits) | new label + 1 (11 bits) | old label + 1 (11 bits)
          list[total + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)(64 * i + lane) << 22) | (ln << 11) | lo;
        total += __popcll(m);
      }
      if (total == 0) continue;
      // (wave-private list: own LDS writes are visible to own reads in order)
      const float *xr = x + row0 * d;
      // (wave-uniform by construction; readfirstlane puts the row number and both labels on the scalar unit: the run
      //  tests become scalar branches instead of v_cmp + saveexec pairs and the row address a scalar multiply)
      auto entry = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)list[min(i, total - 1)]); };
      auto entry_v = [&](int i) { return list[min(i, total - 1)]; };      // (per-lane index: the tail columns' lanes)
      auto issue = [&](int i0, gvec_t (&v)[UNROLL][NV], float &t) {
#ifdef HSGK_FX_TIMING
        {   // (probe) the list entries alone: LDS reads that queue behind the other waves' atomics
          uint32_t es = 0;
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) es += entry(i0 + u);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(es)::"memory");
          HSGK_FXT(5);
        }
#endif
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          const int r = (int)(entry(i0 + u) >> 22);
          const float *src = xr + (int64_t)r * d;
#pragma unroll
          for (int h = 0; h < NV; ++h)    // unconditional (d >= 4 here): a branch around a load costs the counted waits
            v[u][h] = HSGK_FX_LOADV(src + 4 * min(lane + 64 * h, nq - 1));
        }
        t = HSGK_FX_LOADT(xr + (int64_t)(entry_v(i0 + min(tu, UNROLL - 1)) >> 22) * d + min(tail0 + tc, d - 1));
      };
      auto fold = [&](int i0, const gvec_t (&v)[UNROLL][NV], const float &t) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
          if (i0 + u < total) {
            const uint32_t e = entry(i0 + u);
            const int labs[2] = {(int)((e >> 11) & 2047u) - 1, (int)(e & 2047u) - 1};
            long long q[NV][4];
#pragma unroll
            for (int h = 0; h < NV; ++h)
#pragma unroll
              for (int j = 0; j < 4; ++j) q[h][j] = to_fixed(v[u][h][j]);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              const int lab = labs[side];              // (wave-uniform)
              if (lab < 0) continue;
              if (lab != clab[side]) { flush_run(side); clab[side] = lab; }
#pragma unroll
              for (int h = 0; h < NV; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) cq[side][h][j] += side ? -q[h][j] : q[h][j];
            }
          }
        }
        if (tact && i0 + tu < total) {
          const uint32_t e = entry_v(i0 + tu);
          const long long qt = to_fixed(t);
          const int ln = (int)((e >> 11) & 2047u) - 1, lo = (int)(e & 2047u) - 1;
          if (ln >= 0) HSGK_FX_ATOMIC(tab + (size_t)ln * d + tail0 + tc, (unsigned long long)qt);
          if (lo >= 0) HSGK_FX_ATOMIC(tab + (size_t)lo * d + tail0 + tc, (unsigned long long)(-qt));
        }
      };
      gvec_t va[UNROLL][NV], vb[UNROLL][NV];
      float ta, tb;
#ifdef HSGK_FX_TIMING
      // probe build (tools/probes/fx_timing.py): where a wave's time goes -- [0] labels + list, [1] waiting for a
      // batch of rows (explicit s_waitcnt vmcnt with the NEXT batch still in flight), [2] conversion + issuing the
      // LDS atomics, [3] waiting for those atomics to drain (s_waitcnt lgkmcnt(0)), [4] issuing loads
      HSGK_FXT(0);
      issue(0, va, ta);
      HSGK_FXT(4);
      for (int i0 = 0; i0 < total; i0 += 2 * UNROLL) {
        issue(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        HSGK_FXT(4);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL * NV + 1) : "memory");
        HSGK_FXT(1);
        fold(i0, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        HSGK_FXT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HSGK_FXT(3);
        issue(i0 + 2 * UNROLL, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        HSGK_FXT(4);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL * NV + 1) : "memory");
        HSGK_FXT(1);
        fold(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        HSGK_FXT(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HSGK_FXT(3);
      }
#else
      issue(0, va, ta);
      for (int i0 = 0; i0 < total; i0 += 2 * UNROLL) {
        issue(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
        fold(i0, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        issue(i0 + 2 * UNROLL, va, ta);
        __builtin_amdgcn_sched_barrier(0);
        fold(i0 + UNROLL, vb, tb);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    flush_run(0);
    flush_run(1);
    __syncthreads();
    // ---- flush the image's table
    unsigned long long *gq = sumq + ((int64_t)b * K + k0) * d;
    for (int i = tid; i < kn * d; i += NW * 64) {
      const int kk = i / d, col = i - kk * d;
      int src = col;                                  // tail columns: in place
      if (col < tail0) {
        const int h = col >> 8, cb = col & 255, nh = min(64, nq - 64 * h);
        src = 256 * h + (cb & 3) * nh + (cb >> 2);
      }
      const unsigned long long v = tab[kk * d + src];
      if (v) atomicAdd(gq + i, v);
    }
    __syncthreads();
    c = ce;
  }
}

// ---------------------------------------------------------------------------
// Matrix-core variant for K <= 64 and up to 8 x 32 bounded columns (|x| <= 1: the normalised
// embedding columns of segment_by_kmeans): the exact integer sums as a contraction
//     sums[k][c] += sum over changed rows r of  w[k][r] * q'(x[r][c]),     w = +1 (row joins k), -1 (row leaves k)
// on v_mfma_i32_32x32x32_i8.  q' = rint(x * 2^40) + 2^41 (non-negative, < 2^42; the bias rides in the rounding
// constant of to_fixed) is cut into three 14-bit digits, each fed as a 6-bit low piece (one-hot operand +-1) and an
// 8-bit high piece (one-hot operand +-64, piece biased by -128 into int8): int8 x int8 products summed in int32 are
// exact, three int32 accumulators per (cluster, column) hold the digit sums of every row the workgroup meets in an
// image (|sum| <= rows x 8255 < 2^31 up to 260 K rows; the kernel folds earlier), and the biases leave at the fold
// through the net row count of the cluster.  The result is the same integer as the LDS-atomic kernels above.
// Why: those run at the LDS atomic rate (~280 CU cycles per changed row in the first update of a call); here the
// table never touches LDS -- every wave owns 32 columns x 64 clusters x 3 digits in 96 accumulator registers, the
// signed one-hot operand of a batch of 32 rows is a 2 KiB byte matrix in LDS that all eight waves read, and a row
// costs ~12 vector instructions per element for the digit cut plus 12 matrix instructions per 32 rows and wave.
// The changed rows travel global -> LDS by LDS-DMA (global_load_lds, 1 KiB = one row per instruction, three batches
// of 32 rows in flight per workgroup: no register holds a row before its digits are cut), so the gather latency
// hides behind two batches of work.  Sparse launches scan up to eight chunks of labels per list.
// Columns beyond the 32-column blocks (d mod 32, at least the two location columns, whose range the caller owns)
// keep the ds_add_u64 route on a small [K][tail] table.
typedef int mx_v4i __attribute__((ext_vector_type(4)));
typedef int mx_v16i __attribute__((ext_vector_type(16)));
constexpr int kMxGroup = 8;                          // batches of 32 changed rows per one-hot matrix group
constexpr int kMxTailMax = 40;                       // tail columns (d - 32 * blocks) the small table holds
constexpr int kMxTailDma = 16;                       // ... of which the LDS-DMA path stages (4 rows x tail <= 64 lanes)
constexpr int kMxCap = HSGK_CHUNK;                   // list entries
constexpr int kMxRing = 3;                           // row batches in LDS
constexpr long long kMxFoldRows = 200000;            // entries after which the int32 digit sums are folded

__device__ __forceinline__ void mx_to_fixed_biased(float v, uint32_t &lo, uint32_t &hi) {
  // rint(v * 2^40) + 2^41 in the low 42 mantissa bits of the sum (see to_fixed)
  const double t = __builtin_fma((double)v, 1099511627776.0, 6755399441055744.0 + 2199023255552.0);
  const unsigned long long b = __builtin_bit_cast(unsigned long long, t);
  lo = (uint32_t)b; hi = (uint32_t)(b >> 32);
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. the row
// batches still in flight by LDS-DMA
__device__ __forceinline__ void mx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS atomics the compiler does not see: it drains vmcnt before every LDS atomic it knows of once LDS-DMA is in
// flight (the atomic might read what the DMA writes) -- these tables are never a DMA target
__device__ __forceinline__ void mx_lds_add(int *p, int v) {
  asm volatile("ds_add_u32 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void mx_lds_add(unsigned long long *p, unsigned long long v) {
  asm volatile("ds_add_u64 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}

static size_t mx_lds_bytes(int K, int tw) {
  return (size_t)kMxRing * 32 * 256 * 4 + (size_t)kMxGroup * 64 * 32 + 2 * (size_t)(kMxCap + 32) * 4 +
         (size_t)kMxRing * 32 * kMxTailDma * 4 + (size_t)64 * (tw > 0 ? tw : 1) * 8 + 64 * 4 + 16;
}

#ifndef HSGK_MX_DEBUG
#define HSGK_MX_DEBUG 0        // elimination builds (tools/probes/ab_mx.sh): 1 no digit cut / MFMA, 2 no row DMA, 3 neither, 4 no MFMA
#endif
template <int NMB>
__global__ __launch_bounds__(512) void update_sums_mfma_kernel(
    const float *__restrict__ x, int d, int ncb, const int32_t *__restrict__ prev,
    const int32_t *__restrict__ cur, const int64_t *__restrict__ chunk_row0,
    const int32_t *__restrict__ chunk_rows, const int32_t *__restrict__ chunk_img, int K,
    unsigned long long *__restrict__ sumq, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tail0 = 32 * ncb, tw = d - tail0;
  float *rows = reinterpret_cast<float *>(lds_raw);                                        // [ring][32][256]
  uint8_t *W = reinterpret_cast<uint8_t *>(rows + kMxRing * 32 * 256);                      // [group][64][32] signed one-hot bytes
  uint32_t *ent = reinterpret_cast<uint32_t *>(W + kMxGroup * 64 * 32);                     // (new label + 1) << 8 | old label + 1
  uint32_t *off = ent + kMxCap + 32;                                                        // row * d (elements from the list's first row)
  float *tails = reinterpret_cast<float *>(off + kMxCap + 32);                              // [ring][32 * tw] (tw <= kMxTailDma)
  unsigned long long *tail = reinterpret_cast<unsigned long long *>(tails + kMxRing * 32 * kMxTailDma);   // [64][tw]
  int *cnt = reinterpret_cast<int *>(tail + 64 * max(tw, 1));                               // [64] net rows per cluster
  int *ctl = cnt + 64;                                                                      // [0] entries, [1] overflow
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 31, g = lane >> 5;
  const int nc = (int)meta->n_chunks;
  const int c_begin = (int)(((int64_t)blockIdx.x * nc) / gridDim.x);
  const int c_end = (int)(((int64_t)(blockIdx.x + 1) * nc) / gridDim.x);
  const bool tdma = tw > 0 && tw <= kMxTailDma;
  mx_v16i acc[NMB][3];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][j][i] = 0;
  for (int i = tid; i < 64 * tw; i += 512) tail[i] = 0ull;
  if (tid < 64) cnt[tid] = 0;
  if (tid < 2) ctl[tid] = 0;
  __syncthreads();
  long long since = 0;                                   // entries since the last fold (workgroup-uniform)
  int c = c_begin, kc = 1;
  while (c < c_end) {
    const int b = chunk_img[c];
    int ce = c + 1;
    while (ce < c_end && ce - c < kc && chunk_img[ce] == b) ++ce;
    const int64_t row0 = chunk_row0[c];
    const float *xr = x + row0 * d;
    // ---- changed rows of the chunks [c, ce) -> ent / off (any order: the sums do not depend on it)
    for (int cc = c; cc < ce; ++cc) {
      const int nrows = chunk_rows[cc];
      const int64_t crow0 = chunk_row0[cc];
      const uint32_t rbase = (uint32_t)(crow0 - row0);
#pragma unroll
      for (int j = 0; j < HSGK_CHUNK / 512; ++j) {
        const int r = j * 512 + tid;
        const bool in = r < nrows;
        const int pl = in ? get_label(prev, crow0 + r) : -1;
        const int cl = in ? get_label(cur, crow0 + r) : -1;
        const uint32_t ln = (uint32_t)cl < (uint32_t)K ? (uint32_t)cl + 1u : 0u;
        const uint32_t lo = (uint32_t)pl < (uint32_t)K ? (uint32_t)pl + 1u : 0u;
        const bool ch = ln != lo;
        const unsigned long long m = __ballot(ch);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&ctl[0], __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          if (ch) {
            const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < kMxCap) {
              ent[idx] = (ln << 8) | lo;
              off[idx] = (rbase + (uint32_t)r) * (uint32_t)d;
            } else {
              ctl[1] = 1;
            }
          }
        }
      }
    }
    __syncthreads();
    const int total = ctl[0];
    if (ctl[1]) {                                        // more changed rows than the list holds: one chunk at a time
      __syncthreads();
      if (tid < 2) ctl[tid] = 0;
      __syncthreads();
      kc = 1;
      continue;
    }
    if (tid < 32 && total + tid < ((total + 31) & ~31)) { ent[total + tid] = 0u; off[total + tid] = 0u; }   // padding: row 0, no label
    const int nb = (total + 31) >> 5;
    // this wave's four rows of batch bi (and their tail columns) -> ring slot bi % kMxRing, by LDS-DMA
    auto dma = [&](int bi) {
      const int buf = bi % kMxRing;
      const uint32_t *op = off + 32 * bi + 4 * w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readfirstlane((int)op[i]);
        if (4 * lane < tail0 && !(HSGK_MX_DEBUG & 2))
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xr + o + 4 * lane),
                                           (void __attribute__((address_space(3))) *)(rows + (buf * 32 + 4 * w + i) * 256), 16, 0, 0);
      }
      if (tdma && lane < 4 * tw && !(HSGK_MX_DEBUG & 2)) {
        const int i = lane / tw, cc = lane - i * tw;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xr + op[i] + tail0 + cc),
                                         (void __attribute__((address_space(3))) *)(tails + (buf * 32 + 4 * w) * tw), 4, 0, 0);
      }
    };
    if (nb > 0) {
      __syncthreads();                                   // padding written, counters read by everyone
      if (tid < 2) ctl[tid] = 0;
      dma(0);
      if (nb > 1) dma(1);
    }
    for (int bi = 0; bi < nb; ++bi) {
      const int gb = bi & (kMxGroup - 1);
      if (gb == 0) {
        const int nbg = min(kMxGroup, nb - bi);
        if (bi) mx_barrier();                            // the previous group's one-hot bytes are no longer read
        for (int i = tid; i < nbg * 512; i += 512) reinterpret_cast<uint32_t *>(W)[i] = 0u;
        mx_barrier();
        const int e = 32 * bi + tid;
        if (tid < 32 * nbg && e < total) {
          const uint32_t en = ent[e];
          const uint32_t ln = en >> 8, lo = en & 255u;
          if (ln) { W[((tid >> 5) * 64 + ln - 1) * 32 + (tid & 31)] = (uint8_t)1; mx_lds_add(&cnt[ln - 1], 1); }
          if (lo) { W[((tid >> 5) * 64 + lo - 1) * 32 + (tid & 31)] = (uint8_t)0xFF; mx_lds_add(&cnt[lo - 1], -1); }
        }
      }
      // this wave's share of batch bi has landed (the loads of batch bi + 1 may still fly), then everybody's
      if (bi + 1 < nb) {
        if (tdma) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      mx_barrier();
      if (bi + 2 < nb) dma(bi + 2);                      // (its ring slot was read in batch bi - 1: before the barrier)
      const int buf = bi % kMxRing;
      if (w < ncb && !(HSGK_MX_DEBUG & 1)) {
        const float *src = rows + (buf * 32 + 16 * g) * 256 + 32 * w + n;
        mx_v4i bn[3], bb[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint32_t pn[3] = {0u, 0u, 0u}, pb[3] = {0u, 0u, 0u};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            uint32_t lo, hi;
            mx_to_fixed_biased(src[(4 * r + s4) * 256], lo, hi);
            const uint32_t mid = __builtin_amdgcn_alignbit(hi, lo, 28);          // bits 28 .. 59
            pn[0] |= (lo & 63u) << (8 * s4);
            pb[0] |= __builtin_amdgcn_ubfe(lo, 6, 8) << (8 * s4);
            pn[1] |= __builtin_amdgcn_ubfe(lo, 14, 6) << (8 * s4);
            pb[1] |= __builtin_amdgcn_ubfe(lo, 20, 8) << (8 * s4);
            pn[2] |= (mid & 63u) << (8 * s4);
            pb[2] |= __builtin_amdgcn_ubfe(hi, 2, 8) << (8 * s4);
          }
#pragma unroll
          for (int j = 0; j < 3; ++j) { bn[j][r] = (int)pn[j]; bb[j][r] = (int)(pb[j] ^ 0x80808080u); }
        }
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) {
          const mx_v4i aw = *reinterpret_cast<const mx_v4i *>(W + ((gb * 64 + 32 * mb + n) * 32 + 16 * g));
          mx_v4i a64;
#pragma unroll
          for (int r = 0; r < 4; ++r) a64[r] = (int)(((uint32_t)aw[r] << 6) & 0xC0C0C0C0u);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
#if HSGK_MX_DEBUG & 4
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][j][r] += (aw[r] & bn[j][r]) + (a64[r] & bb[j][r]);
#else
            acc[mb][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(aw, bn[j], acc[mb][j], 0, 0, 0);
            acc[mb][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a64, bb[j], acc[mb][j], 0, 0, 0);
#endif
          }
        }
      }
      if (w == gb && tw > 0) {                           // the tail columns of the batch: one wave, lane = (row slot, column)
        auto add_tail = [&](int e, int cc, float tvv) {
          const uint32_t en = ent[e];
          const long long qt = to_fixed(tvv);
          if (en >> 8) mx_lds_add(tail + ((en >> 8) - 1) * tw + cc, (unsigned long long)qt);
          if (en & 255u) mx_lds_add(tail + ((en & 255u) - 1) * tw + cc, (unsigned long long)(-qt));
        };
        if (tdma) {
          for (int it = lane; it < 32 * tw; it += 64) {
            const int e = 32 * bi + it / tw;
            if (e < total) add_tail(e, it - (it / tw) * tw, tails[buf * 32 * tw + it]);
          }
        } else {                                         // (wide tails: plain loads; the compiler's waits also drain the ring)
          for (int it = lane; it < 32 * tw; it += 64) {
            const int e = 32 * bi + it / tw, cc = it - (it / tw) * tw;
            if (e < total) add_tail(e, cc, xr[off[e] + tail0 + cc]);
          }
        }
      }
    }
    __syncthreads();                                     // ent / off / the ring are free; cnt and tail are final
    since += total;
    // the next list: up to eight chunks when few rows change
    kc = total * 16 < (ce - c) * HSGK_CHUNK ? 8 : (total * 8 < (ce - c) * HSGK_CHUNK ? 4 : 1);
    c = ce;
    const bool last = c >= c_end || chunk_img[c] != b || since > kMxFoldRows;
    if (last) {
      // ---- fold the digit sums of image b into its table
      unsigned long long *gq = sumq + (int64_t)b * K * d;
      if (w < ncb && since > 0) {
        // sum of q = sum_j 2^(14 j) (acc_j + 8192 count) - 2^41 count
        constexpr long long kBias = 8192ll * (1ll + (1ll << 14) + (1ll << 28)) - (1ll << 41);
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int m = 32 * mb + (i & 3) + 8 * (i >> 2) + 4 * g;
            if (m < K) {
              const long long sq = (long long)acc[mb][0][i] + ((long long)acc[mb][1][i] << 14) +
                                   ((long long)acc[mb][2][i] << 28) + (long long)cnt[m] * kBias;
              if (sq) atomicAdd(gq + (size_t)m * d + 32 * w + n, (unsigned long long)sq);
            }
          }
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][j][i] = 0;
      }
      if (since > 0)
        for (int i = tid; i < K * tw; i += 512) {
          const unsigned long long tvv = tail[i];
          if (tvv) { atomicAdd(gq + (size_t)(i / tw) * d + tail0 + (i - (i / tw) * tw), tvv); tail[i] = 0ull; }
        }
      __syncthreads();
      if (tid < 64) cnt[tid] = 0;
      since = 0;
      __syncthreads();
    }
  }
}

// centroid row = normalise((float)sum * 2^-40): one rounding per element, then the C1 norm chain
__global__ __launch_bounds__(256) void finalize_fx_kernel(const long long *__restrict__ sumq, int d,
                                                          int K, float eps, float *__restrict__ cent,
                                                          int32_t *__restrict__ zero_a, int na,
                                                          int32_t *__restrict__ zero_b, float *__restrict__ errc,
                                                          int32_t *__restrict__ keep_a, int keep_valid) {
  extern __shared__ float row[];    // [d] + 1
  __shared__ float esum[4];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (k == 0 && b == 0) {           // queue counters of the E-step that follows
    for (int i = tid; i < na; i += 256) {
      if (keep_a) keep_a[i] = keep_valid ? zero_a[i] : 0;   // (the all-K row lists: last iteration's lengths steer this one's routing)
      zero_a[i] = 0;
    }
    if (zero_b && tid == 0) zero_b[0] = 0;
  }
  const long long *src = sumq + ((int64_t)b * K + k) * d;
  for (int i = tid; i < d; i += 256) row[i] = (float)src[i] * 9.094947017729282e-13f;   // 2^-40
  __syncthreads();
  if (tid == 0) {
    // the canonical chain; its LDS reads are issued 16 at a time (one dependent read per fmaf cost
    // ~8 us per launch: most of this kernel at training resolutions)
    float ss = 0.0f;
    int i = 0;
    for (; i + 16 <= d; i += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = row[i + u];
#pragma unroll
      for (int u = 0; u < 16; ++u) ss = fmaf(v[u], v[u], ss);
    }
    for (; i < d; ++i) ss = fmaf(row[i], row[i], ss);
    float nrm = sqrtf(ss);
    if (!(nrm >= eps)) nrm = eps;
    row[d] = nrm;
  }
  __syncthreads();
  const float nrm = row[d];
  float *out = cent + ((int64_t)b * K + k) * d;
  float e2 = 0.0f;
  for (int i = tid; i < d; i += 256) {
    const float v = row[i] / nrm;
    out[i] = v;
    const float e = v - (float)(_Float16)v;                  // exact residual
    e2 = fmaf(e, e, e2);
  }
  // errc(k) = |c_k - fp16(c_k)|_2 (x 1.0001): the hi-plane filters' table error bound, measured here instead of by a
  // launch of its own per iteration (kmeans.hip, centroid_half_err_kernel: a bound, so the summation order is free)
  if (errc) {
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if ((tid & 63) == 0) esum[tid >> 6] = e2;
    __syncthreads();
    if (tid == 0) errc[(int64_t)b * K + k] = sqrtf((esum[0] + esum[1]) + (esum[2] + esum[3])) * 1.0001f;
  }
}

// sumq[b][k][:] += the prep workgroups' partial sums whose label is k.  Workgroup per (k, image):
// the image's label list is scanned 2048 entries at a time into an LDS list of the matching
// entries (order is free: integer sums), then thread = column adds the listed partial rows,
// sixteen independent loads in flight per column pass.
__global__ __launch_bounds__(256) void m0_reduce_kernel(const long long *__restrict__ part,
                                                        const int32_t *__restrict__ lab, int entries,
                                                        int d, int K, long long *__restrict__ sumq) {
  __shared__ int list[2048];
  __shared__ int cnt;
  const int k = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int32_t *lb = lab + (int64_t)b * entries;
  const long long *pb = part + (int64_t)b * entries * d;
  for (int c0 = 0; c0 < d; c0 += 1024) {
    long long acc[4] = {0, 0, 0, 0};
    int col[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) col[p] = min(c0 + 256 * p + tid, d - 1);
    for (int e0 = 0; e0 < entries; e0 += 2048) {
      if (tid == 0) cnt = 0;
      int l[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) l[u] = lb[min(e0 + 256 * u + tid, entries - 1)];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u + tid;
        const bool hit = e < entries && l[u] == k;
        const unsigned long long m = __ballot(hit);
        if (m) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&cnt, __popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
        }
      }
      __syncthreads();
      const int n = cnt;
      const int npass = min(4, (d - c0 + 255) / 256);       // column passes of this window that exist
      if (npass == 2) {
        // d = 258 and its kin: the second column pass holds a few columns only -- its loads ride in the first pass's
        // round trips (threads whose second column exists) instead of walking the list a second time
        const bool has2 = c0 + 256 + tid < d;
        for (int i = 0; i < n; i += 16) {
          long long v[16], v2[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const long long *rowp = pb + (int64_t)list[min(i + u, n - 1)] * d;
            v[u] = rowp[col[0]];
            v2[u] = has2 ? rowp[col[1]] : 0ll;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            acc[0] += (i + u < n) ? v[u] : 0ll;
            acc[1] += (i + u < n) ? v2[u] : 0ll;
          }
        }
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          for (int i = 0; p < npass && i < n; i += 16) {       // 16 partial rows in flight per thread
            long long v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = pb[(int64_t)list[min(i + u, n - 1)] * d + col[p]];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[p] += (i + u < n) ? v[u] : 0ll;
          }
      }
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (c0 + 256 * p + tid < d) sumq[((int64_t)b * K + k) * d + c0 + 256 * p + tid] += acc[p];
  }
}

int launch_m0_reduce(const PrepM0 &m0, int B, int WT, int d, hipStream_t s) {
  if (B <= 0 || m0.K <= 0) return 0;
  hipLaunchKernelGGL(m0_reduce_kernel, dim3(m0.K, B), dim3(256), 0, s,
                     reinterpret_cast<const long long *>(m0.part), m0.lab, 2 * WT, d, m0.K,
                     reinterpret_cast<long long *>(m0.sumq));
  HSGK_LAUNCH_CHECK();
  return 0;
}

// any row length (the strip kernel walks the columns in windows of 512)
bool sums_fx_eligible(int d) { return d >= 1; }

// HSGK_MSTEP=mfma: the matrix-core update where it applies (opt-in: exact, measured SLOWER than the LDS-atomic
// kernel -- first update launch of a 48 x 448 x 448 call 1.85 ms against 1.24 ms, the sparse launches 0.16-0.31
// against 0.15-0.31 ms; elimination table in profiles/r04_mstep_mfma.txt: the digit cut costs as many vector
// instructions per element as the 64-bit adds it replaces).  Read per call.
static bool fx_const_d() {                  // HSGK_FX_CONST_D=0: the generic instance whatever the row length (A/B)
  const char *e = getenv("HSGK_FX_CONST_D");
  return !(e && e[0] == '0');
}
static bool mstep_mfma_enabled() {
  const char *e = getenv("HSGK_MSTEP");
  return e && e[0] == 'm';
}

int launch_update_sums(const float *x, int d, const int32_t *prev, const int32_t *cur,
                       const ChunkTable &t, int max_chunks, int K, long long *sumq,
                       const hsgk_segkm_meta *meta, hipStream_t s, int unit_cols) {
  if (max_chunks <= 0) return 0;
  HSGK_REQUIRE(K <= 1023, "too many clusters for the exact-sum update (11-bit label fields)");
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  {
    // matrix-core variant: K <= 64, the leading `unit_cols` columns bounded by 1 in 32-column blocks
    const int ncb = unit_cols / 32 < 8 ? unit_cols / 32 : 8;
    if (K <= 64 && ncb >= 1 && d - 32 * ncb <= kMxTailMax && d - 32 * ncb >= 0 && mstep_mfma_enabled()) {
      const int nranges = max_chunks < n_cu ? max_chunks : n_cu;
      auto km = K <= 32 ? update_sums_mfma_kernel<1> : update_sums_mfma_kernel<2>;
      const size_t ldsm = mx_lds_bytes(K, d - 32 * ncb);
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(km),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsm));
      hipLaunchKernelGGL(km, dim3(nranges), dim3(512), ldsm, s, x, d, ncb, prev, cur, t.chunk_row0, t.chunk_rows,
                         t.chunk_img, K, reinterpret_cast<unsigned long long *>(sumq), meta);
      HSGK_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    // persistent variant: the image's table in LDS, split by clusters over P <= 8 workgroups
    // when it does not fit one (every part re-scans the labels and reads the rows of its clusters)
    // Rows of up to 259 columns (one 16-byte vector per lane): SIXTEEN waves of four rows in flight each (102
    // registers, four waves per SIMD) instead of eight waves of eight -- a wave's strips are a chain of dependent
    // round trips (labels -> list -> rows -> LDS atomics), and twice the waves halve the chain: cfg2 3.55 -> 3.18 ms
    // per call, K = 256 at 768^2 1.75 -> 1.49.  Longer rows (two vectors per lane, K split over workgroups) keep
    // eight waves of eight (16 waves measured slower there: 1.28 -> 1.47 ms at K = 128, D = 386).
    const bool narrow = d / 4 <= 64;
    const int NWP = narrow ? 16 : 8;
    int P = 1;
    while (P < 8 && (size_t)((K + P - 1) / P) * d * 8 + 16 + (size_t)NWP * 256 * 4 + 32 > 150 * 1024) ++P;
    const size_t ldsp = (size_t)((K + P - 1) / P) * d * 8 + 16 + (size_t)NWP * 256 * 4 + 32;
    if (ldsp <= 150 * 1024 && d <= 515 && d >= 4) {
      auto launch = [&](auto kp) -> int {
        HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kp),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
        const int nranges = max_chunks < n_cu / P ? max_chunks : (n_cu / P > 0 ? n_cu / P : 1);
        hipLaunchKernelGGL(kp, dim3(nranges * P), dim3(NWP * 64), ldsp, s, x, d, prev, cur, t.chunk_row0,
                           t.chunk_rows, t.chunk_img, K, P, reinterpret_cast<unsigned long long *>(sumq), meta);
        HSGK_LAUNCH_CHECK();
        return 0;
      };
      if (d == 258 && fx_const_d()) return launch(update_sums_persistent_kernel<16, 4, 1, 258>);
      if (d == 130 && fx_const_d()) return launch(update_sums_persistent_kernel<16, 4, 1, 130>);
      return narrow ? launch(update_sums_persistent_kernel<16, 4, 1>) : launch(update_sums_persistent_kernel<8, 8, 2>);
    }
  }
  constexpr int NW = 4, S = 12;
  auto kern = update_sums_kernel<NW, 4, S>;
  const size_t lds = (size_t)S * (d < 515 ? d : 515) * 8 + 8 + (size_t)(kFxRows + 2) * 4 + (NW + 1) * 4 + (size_t)K * 2 + 32;
  HSGK_REQUIRE(lds <= 150 * 1024, "row too long for the exact-sum table");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(max_chunks * (HSGK_CHUNK / kFxRows)), dim3(NW * 64), lds, s, x, d, prev,
                     cur, t.chunk_row0, t.chunk_rows, t.chunk_img, K,
                     reinterpret_cast<unsigned long long *>(sumq), meta);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int launch_finalize_fx(const long long *sumq, int d, int K, int B, float eps, float *cent, hipStream_t s,
                       int32_t *zero_a, int na, int32_t *zero_b, float *errc, int32_t *keep_a, bool keep_valid) {
  if (B <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(finalize_fx_kernel, dim3(K, B), dim3(256), (size_t)(d + 1) * 4, s, sumq, d, K, eps,
                     cent, zero_a, zero_a ? na : 0, zero_b, errc, zero_a ? keep_a : nullptr, keep_valid ? 1 : 0);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // namespace hsgk

#ifdef HSGK_FX_TIMING
extern "C" __attribute__((visibility("default"))) int hsgk_debug_fx_timing(unsigned long long *out) {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(hsgk::g_fx_ts), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out[i] = h[i];
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hsgk::g_fx_ts), z, sizeof(z));
  return 0;
}
#endif
