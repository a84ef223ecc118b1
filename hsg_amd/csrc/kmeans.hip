// kmeans.hip -- Lloyd iteration of the spherical k-means (reference
// hsg/utils/segsort/common.py:67-97): M-step = calculate_prototypes_from_labels
// (:11-41), E-step = find_nearest_prototypes (:44-64).
//
// Canonical arithmetic:
//   M  order C2 (this file: the batch M-step of the C ABI, kmeans_with_initial_labels):
//      per chunk (<= 2048 consecutive rows of one image) every (cluster, column) is
//      summed sequentially in row order from +0.0f; chunk partials are summed
//      sequentially in chunk order from +0.0f.  The Lloyd loop of segment_by_kmeans
//      uses the exact sums of order C2x instead (sums_fx.hip).  Either way the
//      centroid is the sum divided by its norm (fmaf chain over columns, sqrtf, '/').
//   E  <x, c_k> is one fmaf chain over ascending column index from +0.0f --
//      exactly what v_mfma_f32_32x32x2_f32 computes (k-ordered fmaf chain) --
//      argmax takes the first maximal index.
#include "common.h"
#include <atomic>
#include <mutex>

#include "accumulate.h"
#include "score_tiles.h"
#include "score_tiles_bf16.h"
#include "score_tiles_f16.h"
#include "score_tiles_f16t.h"

static_assert(hsgk::kHalfSlackRows == hsgk::kHalfSlackRowsHost, "fp16 copy slack");

namespace hsgk {

// max(a, b) in ONE instruction for a bit-tagged score: fmaxf makes the compiler canonicalise an operand whose bits
// came out of integer arithmetic first (v_max_f32 x, x, x -- a fourth instruction per score in the tagged top-2
// loops, a quarter of their work); v_med3_f32(a, b, +inf) takes raw inputs, and the +inf sits in a scalar register
// the compiler cannot see through (a visible constant is folded back to fmax).  NaN operands are ignored like fmaxf's.
__device__ __forceinline__ float max_raw(float a, float b) {
  float pinf;
  asm("s_mov_b32 %0, 0x7f800000" : "=s"(pinf));
  return __builtin_amdgcn_fmed3f(a, b, pinf);
}



// ===========================================================================
// M-step, stage 1: chunk partial sums.
// One workgroup (4 waves) per chunk.  Every cluster is OWNED by one wave
// (hash of the label) and a wave walks the chunk's rows in order, so each LDS
// word is only ever updated by one wave, in row order: plain LDS
// read-add-write (the LDS executes one wave's DS instructions in issue order)
// gives the sequential sum of order C2 independent of scheduling.  LDS float
// atomics were measured ~160 cycles per wave-instruction on gfx950 and are not
// used.  A wave reads a whole row with VEC floats per lane (1 KiB per load
// instruction at VEC=4) and keeps UNROLL rows in flight.
//
// LDS row stride DS = d rounded up to VEC floats so the per-lane VEC-wide
// LDS accesses stay naturally aligned.
template <int VEC, int UNROLL, int NW, int TAILV>
__global__ __launch_bounds__(NW * 64) void accumulate_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ klab,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    int K, int kb0, int kbn, float *__restrict__ partial, unsigned char *__restrict__ pmask,
    const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ float sums[];   // [kbn][DS], the row list, the presence flags
  __shared__ int wcount[8];
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int DS = (d + VEC - 1) / VEC * VEC;
  const int tot = kbn * DS;
  uint32_t *rlist = reinterpret_cast<uint32_t *>(sums + tot);   // [HSGK_CHUNK] (row << 10 | label)
  unsigned char *present = reinterpret_cast<unsigned char *>(rlist + HSGK_CHUNK);   // [kbn rounded up to 4]
  for (int i = tid; i < tot; i += NW * 64) sums[i] = 0.0f;
  for (int i = tid; i < (kbn + 3) / 4; i += NW * 64) reinterpret_cast<uint32_t *>(present)[i] = 0u;
  // (chunk_accumulate's barrier orders the zeroing before the first fold / flag)
  const int64_t row0 = chunk_row0[c];
  chunk_accumulate<VEC, UNROLL, int32_t, NW, TAILV>(x + row0 * d, d, DS, klab + row0, chunk_rows[c],
                                                    kb0, kbn, sums, rlist, wcount, present);
  __syncthreads();
  // only the clusters that own rows of this chunk are written; pmask tells finalize
  // which partial rows exist (a skipped row is an exact +0.0 sum: adding it is the identity)
  float *out = partial + ((int64_t)c * K + kb0) * d;
  for (int k = w; k < kbn; k += NW) {
    const bool has = present[k] != 0;
    if (lane == 0) pmask[(int64_t)c * K + kb0 + k] = has ? 1 : 0;
    if (has)
      for (int i = lane; i < d; i += 64) out[k * d + i] = sums[k * DS + i];
  }
}

int launch_accumulate(const float *x, int d, const int32_t *klab, const ChunkTable &t,
                      int max_chunks, int K, float *partial, unsigned char *pmask,
                      const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  const bool wide = d >= 256;
  const int DS = wide ? (d + 3) / 4 * 4 : d;
  // cluster rows per pass so that the table fits LDS (two workgroups per CU
  // when possible: <= 76 KiB each).
  const size_t list_bytes = (size_t)HSGK_CHUNK * 4 + 1024;      // row list + presence flags
  const size_t budget2 = 78 * 1024 - list_bytes, budget1 = 156 * 1024 - list_bytes;
  int kbn = K;
  if ((size_t)kbn * DS * 4 > budget2) {
    kbn = (int)(budget1 / ((size_t)DS * 4));
    if (kbn > K) kbn = K;
  }
  if (kbn > 1024) kbn = 1024;               // label field of the row list is 10 bits
  HSGK_REQUIRE(kbn >= 1, "row too long for the LDS segment table");
  // 16-byte global loads only need dword alignment on gfx950.
  constexpr int NWA = 4;   // measured: 4 waves x 16 rows in flight 2.17 ms, 8 waves 2.32 ms, 24 rows 3.23 ms (cfg2)
  // rows of 256 + (65..255) columns (C = 384: d = 386) prefetch their second vector too
  const bool tailv = wide && d < 512 && d - 256 > 64;
  auto kern = tailv ? accumulate_kernel<4, 8, NWA, 1>
              : wide ? accumulate_kernel<4, 16, NWA, 0> : accumulate_kernel<1, 16, NWA, 0>;
  size_t lds = (size_t)kbn * DS * 4 + list_bytes;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(158 * 1024)));
  for (int kb0 = 0; kb0 < K; kb0 += kbn) {
    int cur = K - kb0 < kbn ? K - kb0 : kbn;
    hipLaunchKernelGGL(kern, dim3(max_chunks), dim3(NWA * 64), lds, s, x, d, klab,
                       t.chunk_row0, t.chunk_rows, K, kb0, cur, partial, pmask, meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

// ===========================================================================
// M-step, stage 2: sum the chunk partials of one (image, cluster) in chunk
// order, normalise, write the centroid row.  grid = (K, B).
__global__ __launch_bounds__(256) void finalize_kernel(
    const float *__restrict__ partial, const unsigned char *__restrict__ pmask, int d, int K,
    const int32_t *__restrict__ img_chunk0, float eps, float *__restrict__ cent) {
  extern __shared__ float row[];    // [d] + 1, then the list of chunks that hold rows of this cluster
  __shared__ int npresent;
  const int k = blockIdx.x, b = blockIdx.y;
  const int c0 = img_chunk0[b], c1 = img_chunk0[b + 1];
  const int tid = threadIdx.x;
  int *plist = reinterpret_cast<int *>(row + d + 1);          // [c1 - c0], ascending
  if (tid < 64) {                                              // wave 0: ordered compaction of the mask column
    int n = 0;
    for (int base = c0; base < c1; base += 64) {
      const int c = base + tid;
      const bool has = c < c1 && pmask[(int64_t)c * K + k] != 0;
      const unsigned long long m = __ballot(has);
      if (has) plist[n + __popcll(m & ((1ull << tid) - 1ull))] = c;
      n += __popcll(m);
    }
    if (tid == 0) npresent = n;
  }
  __syncthreads();
  const int np = npresent;
  for (int i = tid; i < d; i += 256) {
    float tsum = 0.0f;
    int q = 0;
    for (; q + 4 <= np; q += 4) {                              // loads of four partial rows in flight
      const float p0 = partial[((int64_t)plist[q] * K + k) * d + i];
      const float p1 = partial[((int64_t)plist[q + 1] * K + k) * d + i];
      const float p2 = partial[((int64_t)plist[q + 2] * K + k) * d + i];
      const float p3 = partial[((int64_t)plist[q + 3] * K + k) * d + i];
      tsum = tsum + p0;
      tsum = tsum + p1;
      tsum = tsum + p2;
      tsum = tsum + p3;
    }
    for (; q < np; ++q) tsum = tsum + partial[((int64_t)plist[q] * K + k) * d + i];
    row[i] = tsum;
  }
  __syncthreads();
  if (tid == 0) {
    float ss = 0.0f;
    for (int i = 0; i < d; ++i) ss = fmaf(row[i], row[i], ss);
    float nrm = sqrtf(ss);
    if (!(nrm >= eps)) nrm = eps;
    row[d] = nrm;
  }
  __syncthreads();
  const float nrm = row[d];
  float *out = cent + ((int64_t)b * K + k) * d;
  for (int i = tid; i < d; i += 256) out[i] = row[i] / nrm;
}

int launch_finalize(const float *partial, const unsigned char *pmask, int d, int K, int B,
                    const ChunkTable &t, int max_chunks_per_image, float eps, float *cent,
                    hipStream_t s) {
  if (B <= 0 || K <= 0) return 0;
  // LDS: the row + the present-chunk list (at most the image with the most chunks)
  const size_t lds = ((size_t)(d + 1) + (size_t)(max_chunks_per_image > 0 ? max_chunks_per_image : 1)) * 4;
  HSGK_REQUIRE(lds <= 160 * 1024, "too many chunks for the finalize list");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(finalize_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(finalize_kernel, dim3(K, B), dim3(256), lds, s,
                     partial, pmask, d, K, t.img_chunk0, eps, cent);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// E-step: assign kernel = score_tiles engine (score_tiles.h) + argmax epilogue.
// One workgroup per chunk (or per 1/split of a chunk); lane (j, h) of a wave
// ends a tile with half of the KB scores of row j, so the argmax is in-register
// plus one cross-half exchange; ties keep the first (lowest) index.
struct ArgmaxEpi {
  int kb0, K, nrows, first_block;
  int64_t crow0;
  int32_t *klab;
  float *best;
  template <int MB>
  __device__ inline void operator()(int tile, const f32x16 (&acc)[MB]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[m][r];
        if (k < K && v > bv) { bv = v; bi = k; }      // ascending k, strict >: first maximum
      }
    const float ov = __shfl_xor(bv, 32);
    const int oi = __shfl_xor(bi, 32);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    const int px = tile * TPX + w * 32 + j;
    if (h == 0 && px < nrows) {
      const int64_t row = crow0 + px;
      if (bi == 0x7fffffff) bi = kb0;           // every score NaN: keep first index
      if (first_block) {
        put_label(klab, row, bi);
        if (best) best[row] = bv;
      } else if (bv > best[row]) {              // later blocks win only strictly
        put_label(klab, row, bi);
        best[row] = bv;
      }
    }
  }
};

template <int KB, int NW, int KC, bool EVEN_D>
__global__ __launch_bounds__(NW * 64) void assign_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K, int kb0,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int32_t *__restrict__ klab,
    float *__restrict__ best, int first_block, int split,
    const hsgk_segkm_meta *__restrict__ meta) {
  constexpr int TPX = NW * 32;
  extern __shared__ float lds[];
  // `split` workgroups share one chunk (split = 1 when there are plenty of
  // chunks; > 1 keeps all CUs busy on small batches); each takes a contiguous
  // range of the chunk's tiles and stages the centroid block once.
  const int c = blockIdx.x / split;
  if (c >= meta->n_chunks) return;
  const int part = blockIdx.x - c * split;
  const int tps = (HSGK_CHUNK / TPX + split - 1) / split;      // tiles per workgroup
  const int nrows = min(chunk_rows[c] - part * tps * TPX, tps * TPX);
  if (nrows <= 0) return;
  const int64_t crow0 = chunk_row0[c] + (int64_t)part * tps * TPX;
  const int b = chunk_img[c];
  ArgmaxEpi epi{kb0, K, nrows, first_block, crow0, klab, best};
  score_tiles<KB, NW, KC, EVEN_D>(x, d, cent + ((int64_t)b * K + kb0) * d, min(KB, K - kb0),
                                  crow0, nrows, lds, epi);
}

// ===========================================================================
// E-step, fast path: bf16x3 split filter (score_tiles_bf16.h) + exact re-score
// of the rows whose two best approximate scores are closer than kSplitGap.
// Requires unit-norm rows and centroids (true inside the Lloyd loop), K <= 64,
// even d.  Both kernels use the same (chunk, part) -> row-range mapping.
// Queue entry of the exact pass: the row, its image and the CANDIDATE centroids
// (every k whose approximate score is within kSplitGap of the best one -- a
// superset of all exact maximisers, see score_tiles_bf16.h).  cand = up to three
// indices in bytes 0..2 and their count in byte 3; count 255 = "more than three,
// re-score all K".
// Exact-queue entry: row id + up to seven candidate clusters.  cand = c0 | c1 << 8 | c2 << 16 |
// n << 24 (n = 1..7 candidates; 255: re-score all K), cand_hi = c3 | c4 << 8 | c5 << 16 | c6 << 24.
// The image of a row is found from the row offsets (binary search) by the exact pass.
struct SplitEntry { int32_t row; uint32_t cand; uint32_t cand_hi; };
constexpr int kSplitLdsList = 1024;      // per-workgroup staging of entries (6 B each)

struct SplitEpi {
  int K, nrows, img;
  int64_t crow0;
  int32_t *klab;
  uint16_t *qpx;         // LDS [kSplitLdsList]
  uint32_t *qcand;       // LDS [kSplitLdsList]
  int *qn;               // LDS counter
  SplitEntry *gqueue;    // global queue (overflow path)
  int32_t *gcount;
  const int32_t *rlw;    // gathered pass (second level): this wave's LDS row-id slots [2][32], else null
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[2]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    // branch-free running top-2: b2 = max(b2, min(b1, v)); b1 = max(b1, v)
    float sc[2][16];
    float b1 = -INFINITY, b2 = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = k < K ? sacc[m][r] : -INFINITY;
        sc[m][r] = v;
        b2 = fmaxf(b2, fminf(b1, v));
        bi = v > b1 ? k : bi;
        b1 = max_raw(b1, v);
      }
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int oi = __shfl_xor(bi, 32);
    // merged top-2 of the two halves
    float t1, t2;
    int ti;
    if (o1 > b1) { t1 = o1; ti = oi; t2 = fmaxf(b1, o2); }
    else { t1 = b1; ti = bi; t2 = fmaxf(o1, b2); }
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const bool amb = valid && !(t1 - t2 > kSplitGap);       // ambiguous (or NaN)
    const int64_t grow = rlw ? (int64_t)rlw[(tile & 1) * 32 + j] : crow0 + px;
    if (h == 0 && valid) put_label(klab, grow, ti);
    if (!__any(amb)) return;
    // candidate list of this lane's half, then merged with the partner half
    const float thr = t1 - kSplitGap;
    uint32_t list = 0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool hit = sc[m][r] >= thr;                    // NaN scores never hit
        list = hit ? ((list << 8) | k) : list;
        cnt += hit ? 1 : 0;
      }
    const uint32_t olist = __shfl_xor(list, 32);
    const int ocnt = __shfl_xor(cnt, 32);
    const int tot = cnt + ocnt;
    uint32_t cand = 255u << 24;
    if (tot <= 3 && tot >= 1 && t1 == t1)
      cand = (list & ((1u << (8 * cnt)) - 1u)) | (olist << (8 * cnt)) | ((uint32_t)tot << 24);
    if (h == 0 && amb) {
      const int pos = atomicAdd(qn, 1);
      if (pos < kSplitLdsList) {
        qpx[pos] = (uint16_t)px;
        qcand[pos] = cand;
      } else {                                               // staging full: straight to global
        const int g = atomicAdd(gcount, 1);
        gqueue[g] = SplitEntry{(int32_t)grow, cand, 0u};
      }
    }
  }
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void assign_split_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, int32_t *__restrict__ gcount, int split,
    const hsgk_segkm_meta *__restrict__ meta) {
  constexpr int TPX = NW * 32;
  // All LDS comes from ONE dynamic array: a static __shared__ object in front
  // of it would leave the base only 4-byte aligned and every ds_read_b128 of the
  // engine would be split (measured: LDS 87 % busy, 2x slower kernel).
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char *tail = lds_raw + split_lds_bytes<NW>(d);
  int *qnp = reinterpret_cast<int *>(tail - 16);            // [0] count, [1] global base
  uint32_t *qcand = reinterpret_cast<uint32_t *>(tail);     // [kSplitLdsList]
  uint16_t *qpx = reinterpret_cast<uint16_t *>(qcand + kSplitLdsList);
  // PERSISTENT when split == 0: gridDim.x workgroups share the chunks in
  // contiguous ranges (so the table is re-staged only when the image changes and
  // there is no per-chunk launch / ramp-up cost); otherwise one workgroup per
  // 1/split of a chunk (small batches).
  int c_begin, c_end, part = 0, tps = HSGK_CHUNK / TPX;
  if (split == 0) {
    const int nc = (int)meta->n_chunks;
    c_begin = (int)(((int64_t)blockIdx.x * nc) / gridDim.x);
    c_end = (int)(((int64_t)(blockIdx.x + 1) * nc) / gridDim.x);
  } else {
    c_begin = blockIdx.x / split;
    c_end = c_begin < meta->n_chunks ? c_begin + 1 : c_begin;
    part = blockIdx.x - c_begin * split;
    tps = (HSGK_CHUNK / TPX + split - 1) / split;
  }
  int staged_img = -1;
  for (int c = c_begin; c < c_end; ++c) {
    int nrows = min(chunk_rows[c] - part * tps * TPX, tps * TPX);
    if (nrows <= 0) continue;
    const int64_t crow0 = chunk_row0[c] + (int64_t)part * tps * TPX;
    const int b = chunk_img[c];
    if (split == 0)          // merge the following chunks of the same image (adjacent rows) into
      while (c + 1 < c_end && chunk_img[c + 1] == b && chunk_row0[c + 1] == crow0 + nrows &&
             nrows + chunk_rows[c + 1] <= 0xFFFF) {            // one pass: queue offsets are u16
        nrows += chunk_rows[c + 1];
        ++c;
      }
    if (threadIdx.x == 0) qnp[0] = 0;       // ordered before any epilogue by the engine's barrier
    SplitEpi epi{K, nrows, b, crow0, klab, qpx, qcand, qnp, gqueue, gcount, nullptr};
    score_tiles_split<NW, 4>(x, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi,
                             b != staged_img);
    staged_img = b;
    __syncthreads();
    // publish the staged entries with ONE global atomic per chunk; the exact pass
    // then balances the queue over the whole chip
    const int qn = min(qnp[0], kSplitLdsList);
    if (qn > 0) {
      if (threadIdx.x == 0) qnp[1] = atomicAdd(gcount, qn);
      __syncthreads();
      const int base = qnp[1];
      for (int i = threadIdx.x; i < qn; i += NW * 64)
        gqueue[base + i] = SplitEntry{(int32_t)(crow0 + qpx[i]), qcand[i], 0u};
    }
    __syncthreads();                        // queue drained before the next chunk resets it
  }
}

// Exact pass over the queue.  Lane group of 4 = one entry, lane = one candidate:
// each lane runs the canonical C1 chain acc = fmaf(c_k[dd], x[dd], acc) over
// ascending dd for ITS (row, centroid) pair -- bit-identical to the fp32 MFMA
// engine -- streaming both rows with 16-byte loads (8 in flight).  The group
// keeps the largest exact score, lowest index on ties; entries with five to seven
// candidates make each lane run a second chain.  Entries with more (count 255:
// duplicate / empty clusters, wide near-ties) take the whole wave:
// lane k = centroid k, row elements broadcast with v_readlane.
__device__ __forceinline__ float exact_chain(const float *__restrict__ ck, const float *__restrict__ xr, int d) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));    // rows are 8-byte aligned
  float acc = 0.0f;
  const int d4 = d & ~3;
  int t0 = 0;
  for (; t0 + 32 <= d4; t0 += 32) {
    f4u cv[8], xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      cv[u] = *reinterpret_cast<const f4u *>(ck + t0 + 4 * u);
      xv[u] = *reinterpret_cast<const f4u *>(xr + t0 + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = fmaf(cv[u].x, xv[u].x, acc);
      acc = fmaf(cv[u].y, xv[u].y, acc);
      acc = fmaf(cv[u].z, xv[u].z, acc);
      acc = fmaf(cv[u].w, xv[u].w, acc);
    }
  }
  for (int dd = t0; dd < d; ++dd) acc = fmaf(ck[dd], xr[dd], acc);
  return acc;
}

#ifdef HSGK_Q_STATS
__device__ unsigned long long g_qstats[8];   // debug build: exact-queue entries by candidate count
#endif
// The same chain for all 64 lanes of a wave at once, lanes 4 g .. 4 g + 3 = the (up to four) candidates of entry g,
// with the loads shared inside the group of four: exact_chain above makes every lane fetch its two rows in 16-byte
// pieces -- 64 different cache lines per load instruction, and the texture addresser, not the bytes, bounds the
// exact pass (K = 256 at 768^2: 0.32 GB in 0.18 ms, vector pipe 8 % busy).  Here the four lanes of a group fetch
// 64 CONTIGUOUS bytes of one row per instruction (the pixel row once instead of four times, each candidate's
// centroid row in turn), park them in a wave-private LDS window and every lane reads back ITS pair of rows: a
// quarter of the line requests, the arithmetic and its order unchanged.  `kmine`: this lane's centroid (any valid
// index on idle lanes); window: kExactStageFloats floats per wave.
constexpr int kExactStride = 36;                                  // floats per staged 32-float piece (bank spread)
constexpr int kExactStageFloats = (16 + 64) * kExactStride;
__device__ __forceinline__ float exact_chain_coop(const float *__restrict__ ct, int kmine, const float *__restrict__ xr,
                                                  int d, float *__restrict__ win) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));    // rows are 8-byte aligned
  typedef float f4a __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, ci = lane & 3, g = lane >> 2;
  const float *crow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) crow[j] = ct + (int64_t)__shfl(kmine, (lane & ~3) + j) * d;
  float *xs = win + g * kExactStride;
  float *cs = win + (16 + 4 * g) * kExactStride;
  float acc = 0.0f;
  const int d4 = d & ~3;
  int t0 = 0;
  f4u xq[2], cq[4][2];
  auto fetch = [&](int t) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int o = t + 4 * (ci + 4 * s2);
      xq[s2] = *reinterpret_cast<const f4u *>(xr + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) cq[j][s2] = *reinterpret_cast<const f4u *>(crow[j] + o);
    }
  };
  if (32 <= d4) fetch(0);
  for (; t0 + 32 <= d4; t0 += 32) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      *reinterpret_cast<f4a *>(xs + 4 * (ci + 4 * s2)) = xq[s2];
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f4a *>(cs + j * kExactStride + 4 * (ci + 4 * s2)) = cq[j][s2];
    }
    if (t0 + 64 <= d4) fetch(t0 + 32);                   // (the next piece is on its way while this one is summed)
    // (wave-private window: own writes are visible to own reads in order)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f4a xv = *reinterpret_cast<const f4a *>(xs + 4 * u);
      const f4a cv = *reinterpret_cast<const f4a *>(cs + ci * kExactStride + 4 * u);
      acc = fmaf(cv.x, xv.x, acc);
      acc = fmaf(cv.y, xv.y, acc);
      acc = fmaf(cv.z, xv.z, acc);
      acc = fmaf(cv.w, xv.w, acc);
    }
  }
  const float *ck = ct + (int64_t)kmine * d;
  for (int dd = t0; dd < d; ++dd) acc = fmaf(ck[dd], xr[dd], acc);
  return acc;
}

// One batch of 16 exact-queue entries on a wave: lanes 4 g .. 4 g + 3 serve entry g (`ent`, `have`
// and `img` are per-group values, identical on the four lanes); writes the exact label of the row.
// win != nullptr: the shared-load chains (exact_chain_coop) through that wave-private LDS window.
// hl.rows != nullptr: an entry that asks for ALL K centroids is not scored here (one entry at a time on a whole
// wave, K chains from K different table rows: 2 - 2.7 ms per iteration at K = 256 when a few per cent of the rows
// sit between many near-duplicate centroids -- the 'mixture' input at 768^2, profiles/r05_cfg4_mixture_before.txt);
// its row goes on the image's list for assign_hard_rows_kernel, the same C1 chains on the fp32 matrix pipe.
// The first `skip` such entries of an image stay on the whole-wave path below (a few hundred per iteration on i.i.d.
// rows: cheaper than the dense pass's start-up), list positions [skip, count) go to the dense pass.
// prev (nullable): the images' counts of the previous Lloyd iteration -- an image that had more than `skip` such
// rows then will have them again and the dense pass will run anyway, so all of its entries go to the list.
// Counters sit kHardStride ints apart (one 128-byte line per image, `prev` in lines of its own): a line that takes
// atomics serialises everything else that touches it (profiles/r05_hard_routing_ab.txt: a plain load of `prev`
// from the counters' line made the exact pass 2.5 x slower).
constexpr int kHardStride = 32;
struct HardList { int32_t *rows; int32_t *count; int64_t cap; int skip; const int32_t *prev; };   // [B][cap] ids
// A wave collects its all-K rows in a wave-private LDS buffer and publishes them 48 - 64 at a time: one global
// atomic per flush and image instead of one per batch of 16 entries (31 k -> 2 k per iteration at 4 x 768^2).
constexpr int kHardBuf = 64;
struct HardBuf { int32_t *row; int32_t *img; int n; };

// one row against all K centroids on the whole wave: lane = centroid k0 + lane, first maximum wins
__device__ __forceinline__ void exact_all_k(const int row, const int himg, const float *__restrict__ x, int d,
                                            const float *__restrict__ cent, int K, int32_t *__restrict__ klab) {
  const int lane = threadIdx.x & 63;
  float hv = -INFINITY;
  int hi = 0x7fffffff;
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + lane;
    float a = -INFINITY;
    if (k < K) a = exact_chain(cent + ((int64_t)himg * K + k) * d, x + (int64_t)row * d, d);
    if (k < K && a == a && a > hv) { hv = a; hi = k; }
  }
  if (hi == 0x7fffffff) hi = lane;                         // (all NaN: lowest lane index, as before)
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(hv, off);
    const int oi = __shfl_xor(hi, off);
    if (ov > hv || (ov == hv && oi < hi)) { hv = ov; hi = oi; }
  }
  if (lane == 0) put_label(klab, row, hi < K ? hi : 0);
}

__device__ __forceinline__ void hard_flush(HardBuf &hb, const HardList hl, const float *__restrict__ x, int d,
                                           const float *__restrict__ cent, int K, int32_t *__restrict__ klab) {
  const int lane = threadIdx.x & 63;
  const bool have = lane < hb.n;
  const int row = have ? hb.row[lane] : 0, img = have ? hb.img[lane] : 0;
  unsigned long long todo = __ballot(have), inkernel = 0ull;
  while (todo) {
    const int lead = __builtin_ctzll(todo);
    const int limg = __builtin_amdgcn_readlane(img, lead);
    const unsigned long long grp = __ballot(have && img == limg) & todo;
    int base = 0;
    if (lane == lead) base = atomicAdd(&hl.count[limg * kHardStride], __popcll(grp));
    base = __shfl(base, lead);
    const bool in = (grp >> lane) & 1ull;
    const int pos = base + __popcll(grp & ((1ull << lane) - 1ull));
    const int skip = (hl.prev && hl.prev[limg * kHardStride] > hl.skip) ? 0 : hl.skip;
    if (in && pos >= skip) hl.rows[(int64_t)limg * hl.cap + pos] = row;
    inkernel |= __ballot(in && pos < skip);
    todo &= ~grp;
  }
  hb.n = 0;
  while (inkernel) {
    const int src = __builtin_ctzll(inkernel);
    inkernel &= inkernel - 1;
    exact_all_k(__builtin_amdgcn_readlane(row, src), __builtin_amdgcn_readlane(img, src), x, d, cent, K, klab);
  }
}

__device__ __forceinline__ void exact_rescore16(const SplitEntry ent, const bool have, const int img,
                                                const float *__restrict__ x, int d,
                                                const float *__restrict__ cent, int K,
                                                int32_t *__restrict__ klab, float *__restrict__ win = nullptr,
                                                const HardList hl = HardList{nullptr, nullptr, 0, 0, nullptr},
                                                HardBuf *hb = nullptr) {
  const int lane = threadIdx.x & 63;
  const int ci = lane & 3;
  const int n = have ? (int)(ent.cand >> 24) : 0;
  // candidate ci, and candidate ci + 4 for the entries that carry more than four
  const int ka = (int)(ci < 3 ? (ent.cand >> (8 * ci)) & 255u : ent.cand_hi & 255u);
  const int kb = (int)((ent.cand_hi >> (8 * (ci + 1))) & 255u);           // (ci + 4 <= 6: ci <= 2)
  const bool act = n != 255 && ci < n;
  const bool act2 = n != 255 && ci < 3 && ci + 4 < n;
#ifdef HSGK_Q_STATS
  if (ci == 0 && have) atomicAdd(&g_qstats[n == 255 ? 7 : min(n, 6)], 1ull);   // (tools/probes/qstats.py)
#endif
  const float *xr = x + (int64_t)ent.row * d;
  const float *ct = cent + (int64_t)img * K * d;
  float acc = -INFINITY;
  if (win) {
    const float a = exact_chain_coop(ct, act ? ka : 0, xr, d, win);          // (all lanes take part in the loads)
    if (act) acc = a;
  } else if (act) {
    acc = exact_chain(ct + (int64_t)ka * d, xr, d);
  }
  float bv = (act && acc == acc) ? acc : -INFINITY;     // NaN never wins
  int bi = act ? ka : 0x7fffffff;
  if (__any(act2)) {
    float acc2 = -INFINITY;
    if (win) {
      const float a = exact_chain_coop(ct, act2 ? kb : 0, xr, d, win);
      if (act2) acc2 = a;
    } else if (act2) acc2 = exact_chain(ct + (int64_t)kb * d, xr, d);
    const float v2 = (act2 && acc2 == acc2) ? acc2 : -INFINITY;
    const int i2 = act2 ? kb : 0x7fffffff;
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
  }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (ci == 0 && n >= 1 && n <= 7) put_label(klab, ent.row, bi == 0x7fffffff ? 0 : bi);
  // entries that need all K centroids
  unsigned long long hard = __ballot(ci == 0 && n == 255);
  if (hl.rows && hb) {
    if (hard) {
      const bool mine = ci == 0 && n == 255;
      if (mine) {
        const int p = hb->n + __popcll(hard & ((1ull << lane) - 1ull));
        hb->row[p] = ent.row;
        hb->img[p] = img;
      }
      hb->n += __popcll(hard);
      if (hb->n > kHardBuf - 16) hard_flush(*hb, hl, x, d, cent, K, klab);
    }
    return;
  }
  while (hard) {
    const int src = __builtin_ctzll(hard);
    hard &= hard - 1;
    exact_all_k(__builtin_amdgcn_readlane(ent.row, src), __builtin_amdgcn_readlane(img, src), x, d, cent, K, klab);
  }
}

__global__ __launch_bounds__(256) void assign_requeue_rows_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    int32_t *__restrict__ klab, const SplitEntry *__restrict__ gqueue,
    const int32_t *__restrict__ gcount, const int64_t *__restrict__ img_row0, int B,
    const HardList hl = HardList{nullptr, nullptr, 0, 0, nullptr}) {
  __shared__ __attribute__((aligned(16))) float exact_win[4 * kExactStageFloats];
  __shared__ int32_t hard_buf[4][2][kHardBuf];
  HardBuf hb{hard_buf[threadIdx.x >> 6][0], hard_buf[threadIdx.x >> 6][1], 0};
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const int total = *gcount;
  const int grp = lane >> 2;
  for (int e0 = wave * 16; e0 < total; e0 += nwaves * 16) {
    const int e = e0 + grp;
    SplitEntry ent{0, 0u, 0u};
    if (e < total) ent = gqueue[e];
    // image of the row: last b with img_row0[b] <= row
    int img = 0;
    {
      int hi = B;
      while (hi - img > 1) {
        const int mid = (img + hi) >> 1;
        if (img_row0[mid] <= (int64_t)ent.row) img = mid; else hi = mid;
      }
    }
    exact_rescore16(ent, e < total, img, x, d, cent, K, klab, exact_win + (threadIdx.x >> 6) * kExactStageFloats, hl,
                    &hb);
  }
  if (hl.rows && hb.n) hard_flush(hb, hl, x, d, cent, K, klab);
}

// ---------------------------------------------------------------------------
// Dense exact pass over the rows that asked for all K centroids (HardList, filled by the exact pass above): units
// of NW * 32 listed rows of one image against the image's whole table in blocks of 64 centroids on the fp32 matrix
// pipe -- v_mfma_f32_32x32x2_f32 IS the canonical C1 chain (score_tiles.h), so the label is the exact engine's
// (first maximum) by construction, no bound involved.  A wave keeps its 32 rows' running best across the table
// blocks in two registers; the rows are gathered once per block (1 KB each, from L2 after the first).  Work: the
// grid strides over the units of all images; a workgroup without a unit reads the B counters and leaves.
struct HardEpi {
  int kb0, K, first;
  float *sv;             // LDS [rows of the unit] running best score ...
  int *si;               // ... and its centroid
  template <int MB>
  __device__ inline void operator()(int tile, const f32x16 (&acc)[MB]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = acc[m][r];
        if (k < K && v > bv) { bv = v; bi = k; }           // ascending k, strict >: first maximum; NaN never wins
      }
    const float ov = __shfl_xor(bv, 32);
    const int oi = __shfl_xor(bi, 32);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    if (h == 0) {                                          // (a row's state is touched by its own wave only)
      const int p = tile * TPX + w * 32 + j;
      if (first || bv > sv[p]) { sv[p] = bv; si[p] = bi; }  // later blocks win only strictly
    }
  }
};

constexpr int kHardMaxTiles = 8;           // tiles of NW * 32 rows per unit at most

template <int NW, int KC, bool EVEN_D>
__global__ __launch_bounds__(NW * 64) void assign_hard_rows_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K, int B, const HardList hl,
    int32_t *__restrict__ klab) {
  constexpr int TPX = NW * 32;
  extern __shared__ float lds[];
  float *sv = lds + score_tiles_lds_bytes<64, NW, KC>(d) / 4;
  int *si = reinterpret_cast<int *>(sv + kHardMaxTiles * TPX);
  // rows on the lists; the unit grows with them so that a table block is staged once for several tiles
  auto skip_of = [&](int b) { return (hl.prev && hl.prev[b * kHardStride] > hl.skip) ? 0 : hl.skip; };
  auto count_of = [&](int b) { return hl.count[b * kHardStride]; };
  // (one image per lane: a loop over the images is one dependent round trip per image -- 13 us before an EMPTY
  //  launch returned at B = 24)
  int64_t total = 0;
  {
    const int ln = threadIdx.x & 63;
    for (int b0 = 0; b0 < B; b0 += 64) total += (b0 + ln < B) ? max(count_of(b0 + ln) - skip_of(b0 + ln), 0) : 0;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
  }
  if (total == 0) return;
  const int64_t tiles_all = (total + TPX - 1) / TPX;
  const int ut = (int)min((int64_t)kHardMaxTiles, max((int64_t)1, (tiles_all + gridDim.x - 1) / gridDim.x));
  const int U = ut * TPX;
  int b = 0, ub = 0;                       // image under the cursor, units before it
  for (int u = blockIdx.x;; u += gridDim.x) {
    int nu = 0;
    while (b < B && u >= ub + (nu = (max(count_of(b) - skip_of(b), 0) + U - 1) / U)) { ub += nu; ++b; }
    if (b >= B) return;
    const int lu = u - ub, skip = skip_of(b);
    const int nrows = min(count_of(b) - skip - lu * U, U);
    const int32_t *list = hl.rows + (int64_t)b * hl.cap + skip + (int64_t)lu * U;
    for (int kb0 = 0; kb0 < K; kb0 += 64) {
      __syncthreads();                     // nobody still reads the previous table block (or the previous unit's)
      HardEpi epi{kb0, K, kb0 == 0, sv, si};
      score_tiles<64, NW, KC, EVEN_D, HardEpi, int32_t>(x, d, cent + ((int64_t)b * K + kb0) * d, min(64, K - kb0),
                                                        0, nrows, lds, epi, list);
    }
    // (own rows only: the wave that wrote a row's state reads it)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < 32)
      for (int t = 0; t * TPX < nrows; ++t) {
        const int p = t * TPX + w * 32 + lane;
        if (p < nrows) put_label(klab, list[p], si[p] == 0x7fffffff ? 0 : si[p]);
      }
  }
}

// The same pass with the rows gathered ONCE: a wave pulls its 32 listed rows through a wave-private LDS window in
// 32-column chunks (128 contiguous bytes of a row per 16 lanes) straight into the registers the matrix instruction
// takes its B operand from -- NFULL * 16 + 1 of them per lane (d = 32 NFULL + 2: 66, 130, 258) -- and keeps them
// there while the image's table streams through LDS in blocks of 64 centroids; the next block is on its way into
// registers while this one is scored.  The pass above re-gathers every row once per table block, and the gather
// (1 KB rows from all over the image, 128 bytes at a time) is what bounds it: 100 k rows of 4 x 768^2 took 0.35 ms
// per iteration, the matrix work alone is 0.09.  Running best per lane in two registers, first maximum wins.
template <int NFULL>
__global__ __launch_bounds__(512) void assign_hard_regs_kernel(
    const float *__restrict__ x, const float *__restrict__ cent, int K, int B, const HardList hl,
    int32_t *__restrict__ klab) {
  constexpr int NW = 8, TPX = NW * 32, D = 32 * NFULL + 2, NS = 16 * NFULL + 1, DP = D | 1, XS = 33;
  constexpr int F2 = 64 * (D / 2), TB = (F2 + 511) / 512;        // float2 elements of a table block, per thread
  extern __shared__ float lds[];
  float *cent_s = lds;                                            // [64][DP]
  float *xw = lds + 64 * DP + (threadIdx.x >> 6) * (32 * XS);     // this wave's window [32][XS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, h = lane >> 5;
  auto skip_of = [&](int b) { return (hl.prev && hl.prev[b * kHardStride] > hl.skip) ? 0 : hl.skip; };
  auto count_of = [&](int b) { return hl.count[b * kHardStride]; };
  // (one image per lane: a loop over the images is one dependent round trip per image -- 13 us before an EMPTY
  //  launch returned at B = 24)
  int64_t total = 0;
  {
    const int ln = threadIdx.x & 63;
    for (int b0 = 0; b0 < B; b0 += 64) total += (b0 + ln < B) ? max(count_of(b0 + ln) - skip_of(b0 + ln), 0) : 0;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
  }
  if (total == 0) return;
  // rows per unit: as many as keep every workgroup of the grid busy, 32 (one wave) to 256 (all eight) -- a short
  // list (the first iteration of an i.i.d. batch: ~4 k rows) then runs one wave per SIMD on many CUs instead of
  // two on a few (the matrix instructions of a unit take the same time for one wave as for four)
  const int UR = (int)min((int64_t)TPX, max((int64_t)32, ((total + gridDim.x - 1) / gridDim.x + 31) / 32 * 32));
  float2 tb[TB];
  auto prefetch = [&](const float *src, int kvalid) {             // table block -> registers (rows >= kvalid: zero)
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int f = tid + 512 * u;
      const int k = f / (D / 2);
      tb[u] = (f < F2 && k < kvalid) ? *reinterpret_cast<const float2 *>(src + 2 * (int64_t)f) : make_float2(0.0f, 0.0f);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      const int f = tid + 512 * u;
      if (f < F2) {
        const int k = f / (D / 2), col = 2 * (f - k * (D / 2));
        cent_s[k * DP + col] = tb[u].x;
        cent_s[k * DP + col + 1] = tb[u].y;
      }
    }
  };
  int b = 0, ub = 0;
  for (int u = blockIdx.x;; u += gridDim.x) {
    int nu = 0;
    while (b < B && u >= ub + (nu = (max(count_of(b) - skip_of(b), 0) + UR - 1) / UR)) { ub += nu; ++b; }
    if (b >= B) return;
    const int lu = u - ub, skip = skip_of(b);
    const int nrows = min(count_of(b) - skip - lu * UR, UR);
    const int32_t *list = hl.rows + (int64_t)b * hl.cap + skip + (int64_t)lu * UR;
    const float *ct = cent + (int64_t)b * K * D;
    prefetch(ct, min(64, K));
    // ---- gather: rows of this wave -> xr (B operand of k-step st: column 2 st + h of row j)
    float xr[NS];
    {
      const int n = nrows - w * 32;                               // rows this wave owns (<= 0: it idles along)
      const int lpx = lane >> 4, lf2 = lane & 15;
      const float *rp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int px = max(min(lpx + 4 * i, n - 1), -(w * 32));   // rows past the end re-read a valid one
        rp[i] = x + (int64_t)list[w * 32 + px] * D + 2 * lf2;
      }
      float2 pre[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pre[i] = *reinterpret_cast<const float2 *>(rp[i]);
#pragma unroll
      for (int q = 0; q < NFULL; ++q) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xw[(lpx + 4 * i) * XS + 2 * lf2] = pre[i].x;
          xw[(lpx + 4 * i) * XS + 2 * lf2 + 1] = pre[i].y;
        }
        if (q + 1 < NFULL) {
#pragma unroll
          for (int i = 0; i < 8; ++i) pre[i] = *reinterpret_cast<const float2 *>(rp[i] + 32 * (q + 1));
        }
        // (wave-private window: LDS operations of a wave execute in order)
#pragma unroll
        for (int st = 0; st < 16; ++st) xr[16 * q + st] = xw[j * XS + 2 * st + h];
      }
      const int jc = max(min(j, n - 1), -(w * 32));
      xr[NS - 1] = x[(int64_t)list[w * 32 + jc] * D + 32 * NFULL + h];
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    __syncthreads();                       // the previous unit's last table block is not read any more
    commit();
    __syncthreads();
    for (int kb0 = 0; kb0 < K; kb0 += 64) {
      if (kb0 + 64 < K) prefetch(ct + (int64_t)(kb0 + 64) * D, min(64, K - kb0 - 64));
      if (w * 32 < nrows) {                 // (wave-uniform: a wave without rows only helps with the table blocks)
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
      const float *cb = cent_s + j * DP + h;
      float a0 = cb[0], a1 = cb[32 * DP];
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        float n0 = 0.0f, n1 = 0.0f;
        if (st + 1 < NS) { n0 = cb[2 * st + 2]; n1 = cb[32 * DP + 2 * st + 2]; }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, xr[st], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, xr[st], acc[1], 0, 0, 0);
        a0 = n0;
        a1 = n1;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float v = acc[m][r];
          if (k < K && v > bv) { bv = v; bi = k; }           // ascending k, strict >: first maximum; NaN never wins
        }
      }
      if (kb0 + 64 < K) {
        __syncthreads();                   // every wave is done with this block
        commit();
        __syncthreads();
      }
    }
    const float ov = __shfl_xor(bv, 32);
    const int oi = __shfl_xor(bi, 32);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    const int px = w * 32 + j;
    if (h == 0 && px < nrows) put_label(klab, list[px], bi == 0x7fffffff ? 0 : bi);
  }
}

static int hard_skip(int B) {               // HSGK_HARD_SKIP=n: entries per image that stay on the whole-wave path
  const char *e = getenv("HSGK_HARD_SKIP");
  if (e) return atoi(e) > 0 ? atoi(e) : 0;
  return 2048 / B > 64 ? 2048 / B : 64;
}
static bool hard_prev_enabled() {          // HSGK_HARD_PREV=0: the skip budget regardless of the last iteration's counts (A/B)
  const char *e = getenv("HSGK_HARD_PREV");
  return !(e && e[0] == '0');
}
static bool hard_rows_enabled() {          // HSGK_HARD=0: the whole-wave chains inside the exact pass (A/B)
  const char *e = getenv("HSGK_HARD");
  return !(e && e[0] == '0');
}

// the launch that follows an exact pass which was handed `hl`; returns 1 when no configuration fits the row length
static int launch_assign_hard_rows(const float *x, int d, const float *cent, int K, int B, const HardList hl,
                                   int32_t *klab, hipStream_t s) {
  if (!hl.rows || B <= 0) return 0;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  {
    // rows gathered once into registers (d = 66 / 130 / 258); HSGK_HARD=lds: the pass that streams them per block
    const char *e = getenv("HSGK_HARD");
    auto regs = [&](auto kern, int nfull) -> int {
      const size_t lds = ((size_t)64 * ((32 * nfull + 2) | 1) + (size_t)8 * 32 * 33) * 4;
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), lds, s, x, cent, K, B, hl, klab);
      HSGK_LAUNCH_CHECK();
      return 0;
    };
    if (!(e && e[0] == 'l')) {
      if (d == 258) return regs(assign_hard_regs_kernel<8>, 8);
      if (d == 130) return regs(assign_hard_regs_kernel<4>, 4);
      if (d == 66) return regs(assign_hard_regs_kernel<2>, 2);
    }
  }
  const bool even = (d & 1) == 0;
  auto go = [&](auto kern, size_t lds) -> int {
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), lds, s, x, d, cent, K, B, hl, klab);
    HSGK_LAUNCH_CHECK();
    return 0;
  };
  const size_t st = (size_t)kHardMaxTiles * 256 * 8;
  const size_t l32 = score_tiles_lds_bytes<64, 8, 32>(d) + st, l16 = score_tiles_lds_bytes<64, 8, 16>(d) + st;
  if (l32 <= 160 * 1024)
    return even ? go(assign_hard_rows_kernel<8, 32, true>, l32) : go(assign_hard_rows_kernel<8, 32, false>, l32);
  if (l16 <= 160 * 1024)
    return even ? go(assign_hard_rows_kernel<8, 16, true>, l16) : go(assign_hard_rows_kernel<8, 16, false>, l16);
  return 1;
}
static bool hard_rows_fit(int d) { return score_tiles_lds_bytes<64, 8, 16>(d) + (size_t)kHardMaxTiles * 256 * 8 <= 160 * 1024; }

static int launch_assign_split(const float *x, int d, const float *cent, int K, int B,
                               const ChunkTable &t, int max_chunks, int32_t *klab,
                               SplitEntry *gqueue, int32_t *gcount, const hsgk_segkm_meta *meta,
                               hipStream_t s) {
  constexpr int NW = 8, TPX = NW * 32, kTiles = HSGK_CHUNK / TPX;
  int split = 1;
  while (split < kTiles && (int64_t)max_chunks * split < 2048) split *= 2;
  int grid = max_chunks * split;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  if (split == 1 && max_chunks >= 4 * n_cu) {       // enough chunks: one persistent WG per CU
    split = 0;
    grid = n_cu;
  }
  HSGK_CHECK_HIP(hipMemsetAsync(gcount, 0, sizeof(int32_t), s));
  {
    auto kern = assign_split_kernel<NW>;
    const size_t lds = split_lds_bytes<NW>(d) + (size_t)kSplitLdsList * 6;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, x, d, cent, K, t.chunk_row0,
                       t.chunk_rows, t.chunk_img, klab, gqueue, gcount, split, meta);
    HSGK_LAUNCH_CHECK();
  }
  // exact pass over the (chip-wide balanced) queue; its length is only known on
  // the device, so a fixed grid strides over it
  hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                     gqueue, gcount, t.img_row0, B);
  HSGK_LAUNCH_CHECK();
  return 0;
}

bool assign_split_eligible(int d, int K) {
  return K <= 64 && split_shape_ok(d) &&
         split_lds_bytes<8>(d) + (size_t)kSplitLdsList * 6 <= 160 * 1024;
}

// image of row r: last b in [0, B) with img_row0[b] <= r (binary search: a linear walk is one
// dependent global load per image -- up to 30 us before a workgroup's first tile at B = 48)
__device__ inline int image_of_row(const int64_t *__restrict__ img_row0, int B, int64_t r) {
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (img_row0[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// ===========================================================================
// Three-level E-step (unit-norm rows, K <= 64):
//   1. assign_half_kernel       fp16 filter over the fp16 copy of the rows (half the
//                               HBM bytes); ambiguous rows -> per-image row queue
//   2. assign_split_rows_kernel bf16x3 filter over the queued fp32 rows (gathered);
//                               ambiguous rows + candidate sets -> exact queue
//   3. assign_requeue_rows_kernel  exact fp32 chains of the candidates
// Labels are identical to assign_kernel (each level only decides rows whose exact
// argmax is strictly unique within its proven error bound).
constexpr int kHalfLdsList = 4096;

struct HalfEpi {
  int K, nrows;
  int64_t crow0;
  int32_t *klab;
  uint16_t *qpx;         // LDS [kHalfLdsList]
  int *qn;               // LDS counter
  int32_t *gq;           // this image's row queue (overflow path)
  int32_t *gcnt;         // its length
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[2], float err) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    // Tagged top-2: a score carries its position among the lane's 32 scores (table block m, register r) in its
    // low 5 mantissa bits -- one v_and_or per score, a perturbation below 32 ulp <= 3.8e-6 that the gap accounts
    // for -- so the running top-2 is max + med3 with no index bookkeeping: three vector instructions per score
    // instead of the ~9 of compare / select / min / max with an explicit index (the epilogue was 11 % of the
    // kernel, tools/probes/ab_tkernel.sh: 863 -> 766 us without it; the matrix work hides completely).  Blocks
    // entirely below K are unmasked, entirely above skipped (uniform).
    float t1 = -INFINITY, t2 = -INFINITY;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (m * 32 >= K) continue;
      if ((m + 1) * 32 <= K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~31u) | (uint32_t)(m * 16 + r));
          t2 = __builtin_amdgcn_fmed3f(t1, t2, v);
          t1 = max_raw(t1, v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~31u) | (uint32_t)(m * 16 + r));
          v = k < K ? v : -INFINITY;
          t2 = __builtin_amdgcn_fmed3f(t1, t2, v);
          t1 = max_raw(t1, v);
        }
      }
    }
    const uint32_t tg = __float_as_uint(t1) & 31u;
    int ti = (int)(tg >> 4) * 32 + (int)(tg & 3u) + 8 * (int)((tg >> 2) & 3u) + 4 * h;
    {
      const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
      const int oi = __shfl_xor(ti, 32);
      if (o1 > t1) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
      else { t2 = fmaxf(o1, t2); }
    }
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const bool amb = h == 0 && valid && !(t1 - t2 > half_gap(err) + 8.0e-6f);  // ambiguous (or NaN); + 2 x the tag perturbation
    if (h == 0 && valid) put_label(klab, crow0 + px, ti);     // provisional for ambiguous rows
    const unsigned long long m = __ballot(amb);
    if (!m) return;
    // one LDS atomic per wave-tile
    int base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(qn, __popcll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (amb) {
      const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < kHalfLdsList) qpx[pos] = (uint16_t)px;
      else gq[atomicAdd(gcnt, 1)] = (int32_t)(crow0 + px);           // staging full: straight to global
    }
  }
};

// Persistent: gridDim.x workgroups share the rows [0, N) of the batch in equal contiguous
// ranges (a multiple of the 256-row tile; no dependence on the chunk table, so small batches
// balance as well as large ones), cut at image boundaries into passes; the table is re-staged
// only when the image changes.
template <int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void assign_half_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, int K, const int64_t *__restrict__ img_row0, int B,
    int32_t *__restrict__ klab, int32_t *__restrict__ q1, int32_t *__restrict__ q1count,
    int64_t q1cap, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32;
  uint16_t *qpx = reinterpret_cast<uint16_t *>(lds_raw + half_lds_bytes<NW>(d));
  int *qnp = reinterpret_cast<int *>(qpx + kHalfLdsList);
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (r >= r_end) return;
  int b = image_of_row(img_row0, B, r);                          // image of the first row
  int staged_img = -1;
  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;                            // (empty images are skipped)
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)(0xFFFF / TPX) * TPX);   // queue offsets are u16
    const int64_t crow0 = r;
    if (threadIdx.x == 0) qnp[0] = 0;       // ordered before any epilogue by the engine's barrier
    HalfEpi epi{K, nrows, crow0, klab, qpx, qnp, q1 + (int64_t)b * q1cap, q1count + b};
    if constexpr (DEPTH == 6)
      score_tiles_half<NW, 6, HalfEpi, 2, 2, 2, false, false, 4>(xm, xt, d, cent + (int64_t)b * K * d, K, crow0, nrows,
                                                                lds_raw, epi, b != staged_img);
    else
      score_tiles_half<NW, DEPTH>(xm, xt, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi,
                                  b != staged_img);
    staged_img = b;
    __syncthreads();
    const int qn = min(qnp[0], kHalfLdsList);
    if (qn > 0) {
      if (threadIdx.x == 0) qnp[1] = atomicAdd(q1count + b, qn);
      __syncthreads();
      int32_t *dst = q1 + (int64_t)b * q1cap + qnp[1];
      for (int i = threadIdx.x; i < qn; i += NW * 64) dst[i] = (int32_t)(crow0 + qpx[i]);
    }
    __syncthreads();                        // queue drained before the next pass resets it
    r += nrows;
  }
}

// The same first level fed from the TILE-ORDERED copy (score_tiles_f16t.h): rows go from global memory
// straight into the MFMA B operand registers, the LDS holds only the table planes and the queue staging.
// Needs every image to start on a multiple of 32 rows (no compaction, H W % 32 == 0).
template <int NW, int NFULL>
__global__ __launch_bounds__(NW * 64) void assign_half_t_kernel(
    const _Float16 *__restrict__ xmT, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, int K, const int64_t *__restrict__ img_row0, int B,
    int32_t *__restrict__ klab, int32_t *__restrict__ q1, int32_t *__restrict__ q1count,
    int64_t q1cap, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32;
  uint16_t *qpx = reinterpret_cast<uint16_t *>(lds_raw + half_t_lds_bytes<2, 2>(d));
  int *qnp = reinterpret_cast<int *>(qpx + kHalfLdsList);
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (r >= r_end) return;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)(0xFFFF / TPX) * TPX);
    const int64_t crow0 = r;
    if (threadIdx.x == 0) qnp[0] = 0;       // ordered before any epilogue by the engine's barrier
    HalfEpi epi{K, nrows, crow0, klab, qpx, qnp, q1 + (int64_t)b * q1cap, q1count + b};
    score_tiles_half_t<NW, NFULL>(xmT, xt, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi,
                                  b != staged_img);
    staged_img = b;
    __syncthreads();
    const int qn = min(qnp[0], kHalfLdsList);
    if (qn > 0) {
      if (threadIdx.x == 0) qnp[1] = atomicAdd(q1count + b, qn);
      __syncthreads();
      int32_t *dst = q1 + (int64_t)b * q1cap + qnp[1];
      for (int i = threadIdx.x; i < qn; i += NW * 64) dst[i] = (int32_t)(crow0 + qpx[i]);
    }
    __syncthreads();
    r += nrows;
  }
}

// row-major fp16 copy xm[n][DM] -> tile order (score_tiles_f16t.h); a wave per 32-row block
__global__ __launch_bounds__(256) void rows_to_tiles_kernel(const _Float16 *__restrict__ xm, int DM, int64_t nblk,
                                                            _Float16 *__restrict__ xmT) {
  const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= nblk) return;
  const char *src = reinterpret_cast<const char *>(xm + (blk * 32 + j) * DM) + 16 * g;
  char *dst = reinterpret_cast<char *>(xmT + blk * 32 * DM) + lane * 16;
  for (int kb = 0; kb < DM / 16; ++kb)
    *reinterpret_cast<uint4 *>(dst + kb * 1024) = *reinterpret_cast<const uint4 *>(src + kb * 32);
}
int launch_rows_to_tiles(const _Float16 *xm, int d, int64_t rows, _Float16 *xmT, hipStream_t s) {
  const int64_t nblk = (rows + 31) / 32;
  if (nblk <= 0) return 0;
  hipLaunchKernelGGL(rows_to_tiles_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, s, xm, half_main_cols(d), nblk, xmT);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// 64 < K <= 128: ONE pass with the hi plane of the table only (the hi + lo planes of 128
// centroids do not fit LDS), four accumulator sets.  The table's own fp16 rounding error
// errc(k) = |c_k - fp16(c_k)|_2 is measured (centroid_half_err_kernel) and enters the bound
// next to the row's: |s~ - s| <= |c| err(row) + |xh| errc(k) + 4.2e-5 (d <= 450: 29 MFMA
// column blocks, gamma_450), so a row is decided when its two best approximate scores differ
// by more than 2.0002 err(row) + 2.003 max_k errc(k) + 8.5e-5; every other row goes straight
// to the exact chains with its candidate set (no bf16x3 level: its planes do not fit either).
__global__ __launch_bounds__(256) void centroid_half_err_kernel(const float *__restrict__ cent, int d,
                                                                int64_t rows, float *__restrict__ errc) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float e2 = 0.0f;
  for (int i = lane; i < d; i += 64) {
    const float v = cent[r * d + i];
    const float e = v - (float)(_Float16)v;                  // exact residual
    e2 = fmaf(e, e, e2);
  }
  for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
  if (lane == 0) errc[r] = sqrtf(e2) * 1.0001f;
}

__device__ inline float half_wide_gap(float err, float errc_max) {
  return fmaf(2.0002f, err, fmaf(2.003f, errc_max, 8.5e-5f));
}

template <int MB, bool HILO = false>
struct HalfWideEpi {
  int K, nrows, img;
  int64_t crow0;
  float errc_max;
  int32_t *klab;
  uint16_t *qpx;         // LDS [kSplitLdsList]
  uint32_t *qcand;       // LDS [kSplitLdsList]
  int *qn;               // LDS counter
  SplitEntry *gqueue;    // global exact queue (overflow path)
  int32_t *gcount;
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[MB], float err) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    // Tagged top-2 (see HalfWide1Epi): a score carries its row within the 32-row table block in its low 4
    // mantissa bits (<= 1.8e-6, added to the gap), so a block's running top-2 is max + med3 without index
    // bookkeeping; blocks entirely below K are unmasked, above skipped; the candidate scan only visits blocks
    // whose maximum reaches an ambiguous row's threshold.
    float bm1[MB], bm2[MB];
    float t1 = -INFINITY, t2 = -INFINITY;
    int tm = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float b1 = -INFINITY, b2 = -INFINITY;
      if (m * 32 < K) {
        if ((m + 1) * 32 <= K) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~15u) | (uint32_t)r);
            b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
            b1 = max_raw(b1, v);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~15u) | (uint32_t)r);
            v = k < K ? v : -INFINITY;
            b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
            b1 = max_raw(b1, v);
          }
        }
      }
      bm1[m] = b1;
      bm2[m] = b2;
      const bool up = b1 > t1;
      t2 = up ? fmaxf(t1, b2) : fmaxf(t2, b1);
      tm = up ? m : tm;
      t1 = up ? b1 : t1;
    }
    const uint32_t tg = __float_as_uint(t1) & 15u;
    int ti = tm * 32 + (int)(tg & 3u) + 8 * (int)(tg >> 2) + 4 * h;
    {
      const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
      const int oi = __shfl_xor(ti, 32);
      if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
      else { t2 = fmaxf(o1, t2); }
    }
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const float gap = (HILO ? half_gap(err) : half_wide_gap(err, errc_max)) + 4.0e-6f;   // (hi + lo table planes: no table error term)
    const bool amb = valid && !(t1 - t2 > gap);               // ambiguous (or NaN)
    if (h == 0 && valid) put_label(klab, crow0 + px, ti);     // provisional for ambiguous rows
    if (!__any(amb)) return;
    // candidates of this lane's half, merged with the partner half (<= 7, else "all")
    const float thr = amb ? t1 - gap - 2.0e-6f : INFINITY;
    // The scan used to read all 16 scores of every block some ambiguous row reaches, with a 64-bit shift / or per
    // score: ~10 vector instructions per score on most tiles -- more than the top-2 pass itself.  A lane's block
    // maximum and runner-up are already known: below thr2 (thr less the tag of the tested value) the block holds no
    // candidate of the lane, with only the maximum above it exactly one, whose index is its tag.
    const float thr2 = thr - 2.0e-6f;
    unsigned long long list = 0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (!__any(bm1[m] >= thr2)) continue;
      if (!__any(bm2[m] >= thr2)) {                     // at most the block maximum: its index is its tag
        const uint32_t tgm = __float_as_uint(bm1[m]) & 15u;
        const uint32_t k = m * 32 + (tgm & 3u) + 8 * (tgm >> 2) + 4 * h;
        const bool hit = bm1[m] >= thr2;
        list = hit ? ((list << 8) | k) : list;
        cnt += hit ? 1 : 0;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                    // some lane holds two candidates in this block: scan it
        const uint32_t k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool hit = k < (uint32_t)K && sacc[m][r] >= thr;  // NaN scores never hit
        list = hit ? ((list << 8) | k) : list;
        cnt += hit ? 1 : 0;
      }
    }
    const unsigned long long olist = __shfl_xor(list, 32);
    const int ocnt = __shfl_xor(cnt, 32);
    const int tot = cnt + ocnt;
    uint32_t cand = 255u << 24, cand_hi = 0u;
    if (tot <= 7 && tot >= 1 && t1 == t1) {
      const unsigned long long all = (list & ((1ull << (8 * cnt)) - 1ull)) | (olist << (8 * cnt));
      cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
      cand_hi = (uint32_t)(all >> 24);
    }
    if (h == 0 && amb) {
      const int pos = tot <= 3 || tot > 7 ? atomicAdd(qn, 1) : kSplitLdsList;   // (the staging holds one word per entry)
      if (pos < kSplitLdsList) {
        qpx[pos] = (uint16_t)px;
        qcand[pos] = cand;
      } else {
        const int g = atomicAdd(gcount, 1);
        gqueue[g] = SplitEntry{(int32_t)(crow0 + px), cand, cand_hi};
      }
    }
  }
};

template <int NW, int DEPTH, int MB, int PLANES = 1, int NBUF = 1>
__global__ __launch_bounds__(NW * 64) void assign_half_wide_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, int32_t *__restrict__ gcount,
    const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32;
  unsigned char *tail = lds_raw + half_lds_bytes<NW, MB, PLANES, NBUF>(d);
  int *qnp = reinterpret_cast<int *>(tail - 16);            // [0] count, [1] global base, [2] errc max bits
  uint32_t *qcand = reinterpret_cast<uint32_t *>(tail);     // [kSplitLdsList]
  uint16_t *qpx = reinterpret_cast<uint16_t *>(qcand + kSplitLdsList);
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (r >= r_end) return;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  float errc_max = 0.0f;
  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)(0xFFFF / TPX) * TPX);
    const int64_t crow0 = r;
    if (PLANES == 1 && b != staged_img) {                      // largest table rounding error of this image
      float m = 0.0f;
      for (int k = threadIdx.x & 63; k < K; k += 64) m = fmaxf(m, errc[(int64_t)b * K + k]);
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      errc_max = m;
    }
    if (threadIdx.x == 0) qnp[0] = 0;
    HalfWideEpi<MB, PLANES == 2> epi{K, nrows, b, crow0, errc_max, klab, qpx, qcand, qnp, gqueue, gcount};
    score_tiles_half<NW, DEPTH, HalfWideEpi<MB, PLANES == 2>, MB, PLANES, NBUF>(
        xm, xt, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi, b != staged_img);
    staged_img = b;
    __syncthreads();
    const int qn = min(qnp[0], kSplitLdsList);
    if (qn > 0) {
      if (threadIdx.x == 0) qnp[1] = atomicAdd(gcount, qn);
      __syncthreads();
      const int base = qnp[1];
      for (int i = threadIdx.x; i < qn; i += NW * 64)
        gqueue[base + i] = SplitEntry{(int32_t)(crow0 + qpx[i]), qcand[i], 0u};
    }
    __syncthreads();
    r += nrows;
  }
}

bool assign_half_wide_eligible(int d, int K) {
  return K > 64 && K <= 128 && half_wide_shape_ok(d) &&
         half_lds_bytes<8, 4, 1, 1>(d) + (size_t)kSplitLdsList * 6 <= 160 * 1024;
}

// errc: [B][K] scratch for the measured table rounding errors
int launch_assign_half_wide(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent,
                            float *errc, int K, int B, const ChunkTable &t, int max_chunks, int32_t *klab,
                            void *qrows, int32_t *qcount, const hsgk_segkm_meta *meta, hipStream_t s,
                            bool table_ready, int32_t *hard_rows, int32_t *hard_count, int64_t hard_cap) {
  if (max_chunks <= 0 || B <= 0) return 0;
  HardList hl{nullptr, nullptr, 0, 0, nullptr};
  if (hard_rows && hard_count && hard_rows_enabled() && hard_rows_fit(d)) {
    hl = HardList{hard_rows, hard_count, hard_cap, hard_skip(B), table_ready && hard_prev_enabled() ? hard_count + (size_t)(B + 1) * kHardStride : nullptr};
  }
  // (also with the lists switched off: hsgk_lloyd_requeued_rows sums these counters)
  if (!table_ready && hard_count) HSGK_CHECK_HIP(hipMemsetAsync(hard_count, 0, sizeof(int32_t) * (size_t)B * kHardStride, s));
  constexpr int NW = 8, TPX = NW * 32, MB = 4;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  const int64_t max_tiles = ((int64_t)max_chunks * HSGK_CHUNK + TPX - 1) / TPX;
  const int grid = (int)(max_tiles < n_cu ? max_tiles : n_cu);
  if (!table_ready) {                      // (inside the Lloyd loop launch_finalize_fx has done both)
    HSGK_CHECK_HIP(hipMemsetAsync(qcount, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(centroid_half_err_kernel, dim3((unsigned)(((int64_t)B * K + 3) / 4)), dim3(256), 0, s,
                       cent, d, (int64_t)B * K, errc);
    HSGK_LAUNCH_CHECK();
  }
  {
    const bool deep = ((d / 64) & 3) == 0;
    auto kern = deep ? assign_half_wide_kernel<NW, 4, MB> : assign_half_wide_kernel<NW, 2, MB>;
    const size_t lds = half_lds_bytes<NW, MB, 1, 1>(d) + (size_t)kSplitLdsList * 6;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, xm, xt, d, cent, errc, K, t.img_row0, B,
                       klab, reinterpret_cast<SplitEntry *>(qrows), qcount, meta);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                     reinterpret_cast<const SplitEntry *>(qrows), qcount, t.img_row0, B, hl);
  HSGK_LAUNCH_CHECK();
  return launch_assign_hard_rows(x, d, cent, K, B, hl, klab, s);
}

// ---------------------------------------------------------------------------
// 128 < K <= 256: the same hi-plane filter in TWO table halves per pass.  The first half
// leaves (best, second best, index) of every row in a 16-byte state record; the second
// half's epilogue gets that record prefetched with the row tail, merges, and decides.
// The first half leaves its best THREE values (and the indices of two): the candidate set of
// an undecided row is then known exactly unless three first-half scores lie within the gap
// (such rows are re-scored against all K centroids by the whole-wave path).
// running top-3 of a lane's scores (values) with the indices of the two best; branch-free
// First half of a two-half pass: per row the best and second-best score (exact), the index of
// the best, and the indices of ALL first-half centroids within the row's gap of that best (at
// most four; more: flag) -- a superset of the first-half members of the final candidate set
// {k : score >= overall best - gap}, whose threshold can only be higher.
// Record: {best, second, idx0 | idx1 << 8 | idx2 << 16 | idx3 << 24 (idx0 = best), count (1..4; 255: more)}.
template <int MB>
struct HalfStateEpi {
  int K, nrows;
  int64_t crow0;
  float errc_max;
  uint4 *state;
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[MB], float err) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    float b1 = -INFINITY, b2 = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = k < K ? sacc[m][r] : -INFINITY;
        b2 = fmaxf(b2, fminf(b1, v));
        bi = v > b1 ? k : bi;
        b1 = max_raw(b1, v);
      }
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int oi = __shfl_xor(bi, 32);
    float t1, t2;
    int ti;
    if (o1 > b1 || (o1 == b1 && oi < bi)) { t1 = o1; ti = oi; t2 = fmaxf(b1, o2); }
    else { t1 = b1; ti = bi; t2 = fmaxf(o1, b2); }
    const float gap = half_wide_gap(err, errc_max);
    uint32_t idx = (uint32_t)ti, count = 1u;
    if (__any(!(t1 - t2 > gap))) {                             // some row of the wave has close seconds
      const float thr = t1 - gap;
      uint32_t list = 0;
      int cnt = 0;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool hit = k < (uint32_t)K && k != (uint32_t)ti && sacc[m][r] >= thr;   // (NaN never hits)
          list = hit ? ((list << 8) | k) : list;
          cnt += hit ? 1 : 0;
        }
      const uint32_t olist = __shfl_xor(list, 32);
      const int ocnt = __shfl_xor(cnt, 32);
      const int tot = cnt + ocnt;                              // besides the best itself
      if (tot <= 3) {
        const uint32_t others = (cnt ? (list & ((1u << (8 * cnt)) - 1u)) : 0u) | (cnt < 4 ? olist << (8 * cnt) : 0u);
        idx |= others << 8;
        count = 1u + (uint32_t)tot;
      } else {
        count = 255u;
      }
      if (!(t1 == t1)) count = 255u;                           // NaN scores: leave it to the exact pass
    }
    const int px = tile * TPX + w * 32 + j;
    if (h == 0 && px < nrows)
      state[crow0 + px] = make_uint4(__float_as_uint(t1), __float_as_uint(t2), idx, count);
  }
};

template <int MB>
struct HalfMergeEpi {
  int K2, nrows, img;          // K2 = centroids in the second half
  int64_t crow0;
  float errc_max;
  int32_t *klab;
  uint16_t *qpx;               // LDS [kSplitLdsList]
  uint32_t *qcand;             // LDS [kSplitLdsList]
  int *qn;
  SplitEntry *gqueue;
  int32_t *gcount;
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[MB], float err, const u32x4 &st) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    constexpr int KH = 32 * MB;
    float b1 = -INFINITY, b2 = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = k < K2 ? sacc[m][r] : -INFINITY;
        b2 = fmaxf(b2, fminf(b1, v));
        bi = v > b1 ? k : bi;
        b1 = max_raw(b1, v);
      }
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int oi = __shfl_xor(bi, 32);
    float u1, u2;
    int ui;
    if (o1 > b1 || (o1 == b1 && oi < bi)) { u1 = o1; ui = oi; u2 = fmaxf(b1, o2); }
    else { u1 = b1; ui = bi; u2 = fmaxf(o1, b2); }
    // merge with the first half (lower indices win ties)
    const float a1 = __uint_as_float(st[0]), a2 = __uint_as_float(st[1]);
    const uint32_t aidx = st[2], acount = st[3];
    float t1, t2;
    int ti;
    if (u1 > a1) { t1 = u1; ti = ui + KH; t2 = fmaxf(a1, u2); }
    else { t1 = a1; ti = (int)(aidx & 255u); t2 = fmaxf(u1, a2); }
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const float gap = half_wide_gap(err, errc_max);
    const bool amb = valid && !(t1 - t2 > gap);                    // ambiguous (or NaN)
    if (h == 0 && valid) put_label(klab, crow0 + px, ti);          // provisional for ambiguous rows
    if (!__any(amb)) return;
    // candidate set {k : score >= best - gap}: second-half members from the registers; first-half
    // members: the stored list (a superset) whenever the first half's best is itself within the gap
    const float thr = t1 - gap;
    unsigned long long list = 0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool hit = k < (uint32_t)K2 && sacc[m][r] >= thr;     // NaN scores never hit
        list = hit ? ((list << 8) | (unsigned long long)(k + KH)) : list;
        cnt += hit ? 1 : 0;
      }
    const unsigned long long olist = __shfl_xor(list, 32);
    const int ocnt = __shfl_xor(cnt, 32);
    int tot = cnt + ocnt;
    unsigned long long all = 0;
    if (tot <= 7) all = (cnt ? (list & ((1ull << (8 * cnt)) - 1ull)) : 0ull) | (cnt < 8 ? olist << (8 * cnt) : 0ull);
    const bool first_in = a1 >= thr;                               // (false for NaN)
    bool over = tot > 7 || !(t1 == t1);
    if (first_in && !over) {
      if (acount == 255u || tot + (int)acount > 7) {
        over = true;
      } else {
        all |= (unsigned long long)(aidx & (acount >= 4u ? 0xFFFFFFFFu : ((1u << (8 * acount)) - 1u))) << (8 * tot);
        tot += (int)acount;
      }
    }
    uint32_t cand = 255u << 24, cand_hi = 0u;
    if (!over && tot >= 1) {
      cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
      cand_hi = (uint32_t)(all >> 24);
    }
    if (h == 0 && amb) {
      const int pos = over || tot <= 3 ? atomicAdd(qn, 1) : kSplitLdsList;   // (the staging holds one word per entry)
      if (pos < kSplitLdsList) {
        qpx[pos] = (uint16_t)px;
        qcand[pos] = cand;
      } else {
        gqueue[atomicAdd(gcount, 1)] = SplitEntry{(int32_t)(crow0 + px), cand, cand_hi};
      }
    }
  }
};

template <int NW, int DEPTH, int MB>
__global__ __launch_bounds__(NW * 64) void assign_half_wide2_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab, uint4 *__restrict__ state,
    SplitEntry *__restrict__ gqueue, int32_t *__restrict__ gcount,
    const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32, KH = 32 * MB;
  unsigned char *tail = lds_raw + half_lds_bytes<NW, MB, 1, 1>(d);
  int *qnp = reinterpret_cast<int *>(tail - 16);
  uint32_t *qcand = reinterpret_cast<uint32_t *>(tail);     // [kSplitLdsList]
  uint16_t *qpx = reinterpret_cast<uint16_t *>(qcand + kSplitLdsList);
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (r >= r_end) return;
  int b = image_of_row(img_row0, B, r);
  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)(0xFFFF / TPX) * TPX);
    const int64_t crow0 = r;
    float em = 0.0f;
    for (int k = threadIdx.x & 63; k < K; k += 64) em = fmaxf(em, errc[(int64_t)b * K + k]);
    for (int off = 32; off > 0; off >>= 1) em = fmaxf(em, __shfl_xor(em, off));
    const float *tb = cent + (int64_t)b * K * d;
    HalfStateEpi<MB> e0{KH, nrows, crow0, em, state};
    score_tiles_half<NW, DEPTH, HalfStateEpi<MB>, MB, 1, 1>(xm, xt, d, tb, KH, crow0, nrows, lds_raw, e0, true);
    __syncthreads();
    if (threadIdx.x == 0) qnp[0] = 0;
    HalfMergeEpi<MB> e1{K - KH, nrows, b, crow0, em, klab, qpx, qcand, qnp, gqueue, gcount};
    score_tiles_half<NW, DEPTH, HalfMergeEpi<MB>, MB, 1, 1, true>(xm, xt, d, tb + (int64_t)KH * d, K - KH, crow0,
                                                                   nrows, lds_raw, e1, true, state);
    __syncthreads();
    const int qn = min(qnp[0], kSplitLdsList);
    if (qn > 0) {
      if (threadIdx.x == 0) qnp[1] = atomicAdd(gcount, qn);
      __syncthreads();
      const int base = qnp[1];
      for (int i = threadIdx.x; i < qn; i += NW * 64)
        gqueue[base + i] = SplitEntry{(int32_t)(crow0 + qpx[i]), qcand[i], 0u};
    }
    __syncthreads();
    r += nrows;
  }
}

// ---------------------------------------------------------------------------
// 128 < K <= 256 in ONE pass: the hi plane of all 256 centroids (143 KB at d = 258) stays in LDS with four
// single-buffered row windows (four waves, one per SIMD, eight accumulator sets each: 161.8 KB), so the fp16
// rows are streamed ONCE per E-step instead of once per table half, no 16-byte state record per row is
// written and read back, and there is one epilogue per row tile.  (profiles/r03_cfg4_*: the two-half kernel
// ran at 0.725 ms per launch with the matrix pipe 21 % busy, the LDS 19 %, the vector ALU 41 % -- phases in
// series, twice.)  Undecided rows go to the workgroup's OWN slice of the exact queue (it starts at the
// workgroup's first row: a workgroup never has more entries than rows) through an LDS counter -- no global
// atomic per entry, no staging list in LDS (there is no room for one); the exact pass walks the slices.
struct SegQueue { int32_t *count; int64_t *row0; };     // per filter workgroup: entries, first row (= slice start)

template <int MB>
struct HalfWide1Epi {
  int K, nrows;
  int64_t crow0;
  float errc_max;
  int32_t *klab;
  int *qn;               // LDS counter of the workgroup (over all its passes)
  SplitEntry *slice;     // the workgroup's slice of the exact queue
  __device__ inline void operator()(int tile, const f32x16 (&sacc)[MB], float err) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    // Scores carry their row WITHIN the 32-row table block in the low 4 mantissa bits (register index r: one
    // v_and_or with an inline constant; a perturbation of at most 15 ulp <= 1.8e-6 that the gap accounts for),
    // so the running top-2 of a block needs no index bookkeeping: b1 = max, b2 = med3(b1, b2, v) -- three
    // vector instructions per score.  Blocks entirely below K are unmasked, entirely above skipped (uniform).
    float bm1[MB], bm2[MB];
    float t1 = -INFINITY, t2 = -INFINITY;
    int tm = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float b1 = -INFINITY, b2 = -INFINITY;
      if (m * 32 < K) {
        if ((m + 1) * 32 <= K) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~15u) | (uint32_t)r);
            b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
            b1 = max_raw(b1, v);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = __uint_as_float((__float_as_uint(sacc[m][r]) & ~15u) | (uint32_t)r);
            v = k < K ? v : -INFINITY;
            b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
            b1 = max_raw(b1, v);
          }
        }
      }
      bm1[m] = b1;
      bm2[m] = b2;
      // merge into the lane's running top-2 (strict >: the lower block wins a tie)
      const bool up = b1 > t1;
      t2 = up ? fmaxf(t1, b2) : fmaxf(t2, b1);
      tm = up ? m : tm;
      t1 = up ? b1 : t1;
    }
    const uint32_t tg = __float_as_uint(t1) & 15u;
    int ti = tm * 32 + (int)(tg & 3u) + 8 * (int)(tg >> 2) + 4 * h;
    {
      const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
      const int oi = __shfl_xor(ti, 32);
      if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
      else { t2 = fmaxf(o1, t2); }
    }
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const float gap = half_wide_gap(err, errc_max) + 4.0e-6f;      // + 2 x the tag perturbation
    const bool amb = valid && !(t1 - t2 > gap);                    // ambiguous (or NaN)
    if (h == 0 && valid) put_label(klab, crow0 + px, ti);          // provisional for ambiguous rows
    if (!__any(amb)) return;
    // candidates of this lane's half, merged with the partner half (<= 7, else "all"); only table blocks
    // whose maximum reaches some ambiguous row's threshold are scanned
    const float thr = amb ? t1 - gap - 2.0e-6f : INFINITY;
    // The scan used to read all 16 scores of every block some ambiguous row reaches, with a 64-bit shift / or per
    // score: ~10 vector instructions per score on most tiles -- more than the top-2 pass itself.  A lane's block
    // maximum and runner-up are already known: below thr2 (thr less the tag of the tested value) the block holds no
    // candidate of the lane, with only the maximum above it exactly one, whose index is its tag.
    const float thr2 = thr - 2.0e-6f;
    unsigned long long list = 0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (!__any(bm1[m] >= thr2)) continue;
      if (!__any(bm2[m] >= thr2)) {                     // at most the block maximum: its index is its tag
        const uint32_t tgm = __float_as_uint(bm1[m]) & 15u;
        const uint32_t k = m * 32 + (tgm & 3u) + 8 * (tgm >> 2) + 4 * h;
        const bool hit = bm1[m] >= thr2;
        list = hit ? ((list << 8) | k) : list;
        cnt += hit ? 1 : 0;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                    // some lane holds two candidates in this block: scan it
        const uint32_t k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool hit = k < (uint32_t)K && sacc[m][r] >= thr;      // NaN scores never hit
        list = hit ? ((list << 8) | k) : list;
        cnt += hit ? 1 : 0;
      }
    }
    const unsigned long long olist = __shfl_xor(list, 32);
    const int ocnt = __shfl_xor(cnt, 32);
    const int tot = cnt + ocnt;
    uint32_t cand = 255u << 24, cand_hi = 0u;
    if (tot <= 7 && tot >= 1 && t1 == t1) {
      const unsigned long long all = (list & ((1ull << (8 * cnt)) - 1ull)) | (olist << (8 * cnt));
      cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
      cand_hi = (uint32_t)(all >> 24);
    }
    if (h == 0 && amb) slice[atomicAdd(qn, 1)] = SplitEntry{(int32_t)(crow0 + px), cand, cand_hi};
  }
};

template <int NW, int DEPTH, int MB, int NFULL>
__global__ __launch_bounds__(NW * 64) void assign_half_wide1_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, SegQueue seg, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32;
  int *qnp = reinterpret_cast<int *>(lds_raw + half_lds_bytes<NW, MB, 1, 1>(d));
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (threadIdx.x == 0) { seg.count[blockIdx.x] = 0; seg.row0[blockIdx.x] = r < r_end ? r : 0; qnp[0] = 0; }
  if (r >= r_end) return;
  const int64_t slice0 = r;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  float errc_max = 0.0f;
  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)1 << 24);
    const int64_t crow0 = r;
    if (b != staged_img) {                                     // largest table rounding error of this image
      float m = 0.0f;
      for (int k = threadIdx.x & 63; k < K; k += 64) m = fmaxf(m, errc[(int64_t)b * K + k]);
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      errc_max = m;
    }
    HalfWide1Epi<MB> epi{K, nrows, crow0, errc_max, klab, qnp, gqueue + slice0};
    score_tiles_half<NW, DEPTH, HalfWide1Epi<MB>, MB, 1, 1>(
        xm, xt, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi, b != staged_img);
    staged_img = b;
    r += nrows;
  }
  __syncthreads();
  if (threadIdx.x == 0) seg.count[blockIdx.x] = qnp[0];
}

// ---------------------------------------------------------------------------
// The same one-pass K <= 256 filter with TWO waves per SIMD: eight waves in four PAIRS.  A pair shares one row
// window (32 rows x 64 columns; each wave loads and stores 16 of its rows) and splits the table: wave `hf` of a
// pair scores the tile against table blocks 4 hf .. 4 hf + 3 (four accumulator sets, 64 registers instead of
// 128), so the LDS budget is that of the four-wave kernel (table 143 KB + four windows) while every SIMD has a
// second wave to issue from when one waits for its LDS operands, its accumulators or the row stream.  The two
// table halves of a row meet in the epilogue through the pair's (then idle) window: top-2 of the partner's half,
// the merged decision and threshold, the partner's candidate list.  Workgroup barriers order the shared window
// (two per 64-column chunk) and the three exchanges of the epilogue.
template <int DEPTH>
__global__ __launch_bounds__(512) void assign_half_pair_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, SegQueue seg, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NP = 4, NW = 8, TPX = NP * 32, KC = 64, XSB = 72, MBW = 4, TR = 256;
  const int DM = half_main_cols(d);
  const int RS = DM + 16 + 8;
  uint16_t *chs = reinterpret_cast<uint16_t *>(lds_raw);                 // [256][RS] hi plane of the table
  uint16_t *xs = chs + TR * RS;                                          // [NP][32][XSB] one window per pair
  int *qnp = reinterpret_cast<int *>(xs + NP * 32 * XSB);               // [0] queue length of the workgroup
  int *pcnt = qnp + 4;                                                   // [NP][2] arrival counters of the pairs' waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int p = wu >> 1, hf = wu & 1;
  uint16_t *xw = xs + p * (32 * XSB);
  // epilogue exchange area of the pair = its window (idle between a tile's last chunk and the next store)
  float *ex_f = reinterpret_cast<float *>(xw);                           // [0..31] t1, [32..63] t2 of the partner half
  int *ex_i = reinterpret_cast<int *>(xw) + 64;                          // [0..31] ti
  float *ex_thr = reinterpret_cast<float *>(xw) + 96;                    // [0..31] threshold of the row (inf: decided)
  unsigned long long *ex_list = reinterpret_cast<unsigned long long *>(reinterpret_cast<int *>(xw) + 128);   // [32]
  int *ex_cnt = reinterpret_cast<int *>(xw) + 192;                       // [0..31]
  int *ex_any = reinterpret_cast<int *>(xw) + 224;                       // [0] any ambiguous row in this pair's tile

  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (tid == 0) { seg.count[blockIdx.x] = 0; seg.row0[blockIdx.x] = r < r_end ? r : 0; qnp[0] = 0; }
  if (tid < 2 * NP) pcnt[tid] = 0;
  if (r >= r_end) return;
  // Pair-local synchronisation instead of workgroup barriers: a wave publishes its arrival count and polls its
  // partner's (LDS executes a wave's operations in order, so everything the wave wrote or read before the count
  // is done when the partner sees it).  The four pairs then run out of phase -- pairs 2 and 3 start half a tile
  // late -- and a SIMD, which hosts one wave of two different pairs, overlaps the matrix phase of one with the
  // vector-heavy epilogue of the other (with workgroup barriers all eight waves moved in lockstep: 0.60 ms per
  // launch against 0.63 for the four-wave kernel).
  int epoch = 0;
  int *cnt_mine = pcnt + 2 * p + hf, *cnt_other = pcnt + 2 * p + (1 - hf);
  auto psync = [&]() {
    ++epoch;
    if (lane == 0) __hip_atomic_store(cnt_mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt_other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch)
      asm volatile("s_nop 3");
    asm volatile("" ::: "memory");
  };
  SplitEntry *slice = gqueue + r;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  float errc_max = 0.0f;
  const int nfull = DM / KC;
  const bool has_tail = d > DM;
  auto bar = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };

  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)1 << 24);
    const int64_t crow0 = r;
    if (b != staged_img) {
      float m = 0.0f;
      for (int k = lane; k < K; k += 64) m = fmaxf(m, errc[(int64_t)b * K + k]);
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      errc_max = m;
      // ---- table -> fp16 hi plane (rows >= K and the padding zeroed); every wave converts whole rows
      bar();                                           // nobody still reads the previous table / windows
      const float *table = cent + (int64_t)b * K * d;
      uint32_t *ch32 = reinterpret_cast<uint32_t *>(chs);
      const int RS2 = RS >> 1, dp = d >> 1;
      constexpr int PPL = 4;
      for (int k0 = w; k0 < TR; k0 += 4 * NW) {
        float2 v[4][PPL];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = min(k0 + u * NW, K - 1);
#pragma unroll
          for (int i = 0; i < PPL; ++i)
            v[u][i] = *reinterpret_cast<const float2 *>(table + (int64_t)k * d + 2 * min(lane + 64 * i, dp - 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + u * NW;
          if (k < TR) {
            const bool live = k < K;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
              const int pr = lane + 64 * i;
              if (pr < RS2) {
                uint32_t hi = 0u, lo = 0u;
                if (live && pr < dp) f16_split2(v[u][i].x, v[u][i].y, hi, lo);
                ch32[k * RS2 + pr] = hi;
              }
            }
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      staged_img = b;
    }

    // ---- the row stream of THIS wave: rows 16 hf .. 16 hf + 15 of the pair's 32-row tile, asm loads with
    //      counted waits (see score_tiles_f16.h); the fp16 copy has kHalfSlackRows readable rows past the end
    const int ntile = (nrows + TPX - 1) / TPX;         // (all pairs run all tiles: they share the barriers)
    const int lpx = lane >> 4, lf = lane & 15;
    const uint32_t voff = (uint32_t)(lpx * DM + 4 * lf) * 2u;
    const char *wbase = reinterpret_cast<const char *>(xm + (crow0 + p * 32 + 16 * hf) * DM);
    const int64_t tile_bytes = (int64_t)TPX * DM * 2, grp_bytes = (int64_t)4 * DM * 2;
    int ld_tile = 0, ld_q = 0;
    constexpr int LOADS = 4;
    auto load_next = [&](uint2 (&pre)[LOADS]) {
      const char *tb = wbase + ld_tile * tile_bytes + ld_q * (KC * 2);
#pragma unroll
      for (int i = 0; i < LOADS; ++i)
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(pre[i]) : "v"(voff), "s"(tb + i * grp_bytes));
      const bool wrap = ld_q + 1 == nfull;
      ld_q = wrap ? 0 : ld_q + 1;
      ld_tile += wrap ? 1 : 0;
    };
#define HSGK_VMWAIT4(N, P) \
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N))
    auto store_chunk = [&](const uint2 (&pre)[LOADS]) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i)
        *reinterpret_cast<uint2 *>(xw + (16 * hf + lpx + 4 * i) * XSB + 4 * lf) = pre[i];
    };
    f32x16 acc[MBW];
    auto zero_acc = [&]() {
#pragma unroll
      for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) acc[m][rr] = 0.0f;
    };
    struct Ops { f16x8 b; f16x8 ah[MBW]; };
    const uint16_t *hp0 = chs + (4 * hf * 32 + j) * RS + 8 * g;          // this wave's table blocks
    auto load_ops = [&](const uint16_t *bp, int col0, Ops &o) {
      o.b = *reinterpret_cast<const f16x8 *>(bp);
#pragma unroll
      for (int m = 0; m < MBW; ++m) o.ah[m] = *reinterpret_cast<const f16x8 *>(hp0 + m * 32 * RS + col0);
    };
    auto mfma_ops = [&](const Ops &o) {
#if defined(HSGK_PAIR_DEBUG) && HSGK_PAIR_DEBUG >= 2          // probes: no matrix work
#pragma unroll
      for (int m = 0; m < MBW; ++m) acc[m][0] += (float)o.ah[m][0] * (float)o.b[0];
#else
#pragma unroll
      for (int m = 0; m < MBW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.ah[m], o.b, acc[m], 0, 0, 0);
#endif
    };
    auto compute_chunk = [&](int q) {
      const uint16_t *bp = xw + j * XSB + 8 * g;
      Ops o0, o1;
      load_ops(bp, q * KC, o0);
      __builtin_amdgcn_sched_barrier(0);
      load_ops(bp + 16, q * KC + 16, o1);
      mfma_ops(o0);
      __builtin_amdgcn_sched_barrier(0);
      load_ops(bp + 32, q * KC + 32, o0);
      mfma_ops(o1);
      __builtin_amdgcn_sched_barrier(0);
      load_ops(bp + 48, q * KC + 48, o1);
      mfma_ops(o0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_ops(o1);
    };
    const uint32_t toff = (uint32_t)(p * 32 + j) * 8u;
    auto load_tail = [&](int tile, uint2 &v) {
      const char *tb = reinterpret_cast<const char *>(xt + crow0 + (int64_t)tile * TPX);
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(toff), "s"(tb));
    };
    uint2 preA[LOADS], preB[LOADS], preC[DEPTH == 4 ? LOADS : 1], preD[DEPTH == 4 ? LOADS : 1];
    if (p >= 2) __builtin_amdgcn_s_sleep(127);          // half a tile (~8 K cycles) behind pairs 0 and 1
    load_next(preA);
    load_next(preB);
    if constexpr (DEPTH == 4) { load_next(preC); load_next(preD); }
    zero_acc();
    uint2 tailv = {0u, 0u};
#define HSGK_PAIR_STEP(PRE, QQ)                                               \
  HSGK_VMWAIT4(LOADS * (DEPTH - 1), PRE);                                     \
  psync();                                 /* the window is free: the partner has read the previous chunk */ \
  store_chunk(PRE);                                                           \
  __builtin_amdgcn_sched_barrier(0);                                          \
  load_next(PRE);                                                             \
  __builtin_amdgcn_sched_barrier(0);                                          \
  psync();                                 /* the window is complete */       \
  compute_chunk(QQ);                                                          \
  __builtin_amdgcn_sched_barrier(0);
    for (int tile = 0; tile < ntile; ++tile) {
      load_tail(tile, tailv);
      for (int q = 0; q < nfull; q += DEPTH) {
        if constexpr (DEPTH == 4) {
          HSGK_PAIR_STEP(preA, q)
          HSGK_PAIR_STEP(preB, q + 1)
          HSGK_PAIR_STEP(preC, q + 2)
          HSGK_PAIR_STEP(preD, q + 3)
        } else {
          HSGK_PAIR_STEP(preA, q)
          HSGK_PAIR_STEP(preB, q + 1)
        }
      }
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tailv) : "n"(LOADS * DEPTH));
      if (has_tail) {                                  // the two location columns: k = 0, 1 of one more k-block
        const u32x4 tv = {g == 0 ? tailv.x : 0u, 0u, 0u, 0u};
        Ops o;
        o.b = __builtin_bit_cast(f16x8, tv);
#pragma unroll
        for (int m = 0; m < MBW; ++m) o.ah[m] = *reinterpret_cast<const f16x8 *>(hp0 + m * 32 * RS + DM);
        mfma_ops(o);
      }
      const float err = __uint_as_float(tailv.y);
#if defined(HSGK_PAIR_DEBUG) && HSGK_PAIR_DEBUG >= 1          // probes: no epilogue
      if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) klab[crow0] = 1;
      psync();
      zero_acc();
      continue;
#endif
      // ---------------- epilogue: this wave's half of the table, then the pair
      const int h = g;
      float bm1[MBW], bm2[MBW];
      float t1 = -INFINITY, t2 = -INFINITY;
      int tm = 0;
#pragma unroll
      for (int m = 0; m < MBW; ++m) {
        const int mg = 4 * hf + m;                     // global table block
        float b1 = -INFINITY, b2 = -INFINITY;
        if (mg * 32 < K) {
          if ((mg + 1) * 32 <= K) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const float v = __uint_as_float((__float_as_uint(acc[m][rr]) & ~15u) | (uint32_t)rr);
              b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
              b1 = max_raw(b1, v);
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const int k = mg * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
              float v = __uint_as_float((__float_as_uint(acc[m][rr]) & ~15u) | (uint32_t)rr);
              v = k < K ? v : -INFINITY;
              b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
              b1 = max_raw(b1, v);
            }
          }
        }
        bm1[m] = b1;
        bm2[m] = b2;
        const bool up = b1 > t1;
        t2 = up ? fmaxf(t1, b2) : fmaxf(t2, b1);
        tm = up ? mg : tm;
        t1 = up ? b1 : t1;
      }
      const uint32_t tg = __float_as_uint(t1) & 15u;
      int ti = tm * 32 + (int)(tg & 3u) + 8 * (int)(tg >> 2) + 4 * h;
      {
        const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
        const int oi = __shfl_xor(ti, 32);
        if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
        else { t2 = fmaxf(o1, t2); }
      }
      psync();                                         // the partner's window reads of the last chunk are done
      if (hf == 1 && h == 0) { ex_f[j] = t1; ex_f[32 + j] = t2; ex_i[j] = ti; }
      psync();
      const int px = tile * TPX + p * 32 + j;
      const bool valid = px < nrows;
      const float gap = half_wide_gap(err, errc_max) + 4.0e-6f;
      bool amb = false;
      float thr = INFINITY;
      if (hf == 0) {
        const float o1 = ex_f[j], o2 = ex_f[32 + j];
        const int oi = ex_i[j];
        if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
        else { t2 = fmaxf(o1, t2); }
        amb = valid && !(t1 - t2 > gap);
        if (h == 0 && valid) put_label(klab, crow0 + px, ti);
        thr = amb ? t1 - gap - 2.0e-6f : INFINITY;
        const bool any = __any(amb);
        if (h == 0) ex_thr[j] = thr;
        if (lane == 0) ex_any[0] = any ? 1 : 0;
      }
      psync();
      const bool any = ex_any[0] != 0;                 // (pair-uniform; the barriers below are unconditional)
      unsigned long long list = 0;
      int cnt = 0;
      if (any) {
        thr = ex_thr[j];
        const float thr2 = thr - 2.0e-6f;              // (see HalfWideEpi: block maximum / runner-up instead of a scan)
#pragma unroll
        for (int m = 0; m < MBW; ++m) {
          if (!__any(bm1[m] >= thr2)) continue;
          const int mg = 4 * hf + m;
          if (!__any(bm2[m] >= thr2)) {
            const uint32_t tgm = __float_as_uint(bm1[m]) & 15u;
            const uint32_t k = mg * 32 + (tgm & 3u) + 8 * (tgm >> 2) + 4 * h;
            const bool hit = bm1[m] >= thr2;
            list = hit ? ((list << 8) | k) : list;
            cnt += hit ? 1 : 0;
            continue;
          }
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const uint32_t k = mg * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            const bool hit = k < (uint32_t)K && acc[m][rr] >= thr;
            list = hit ? ((list << 8) | k) : list;
            cnt += hit ? 1 : 0;
          }
        }
        const unsigned long long olist = __shfl_xor(list, 32);
        const int ocnt = __shfl_xor(cnt, 32);
        // this wave's half of the row: own lane half first, then the partner lane half (<= 7 kept)
        const int tot = cnt + ocnt;
        if (tot <= 7) list = (list & ((1ull << (8 * cnt)) - 1ull)) | (cnt < 8 ? (olist << (8 * cnt)) : 0ull);
        cnt = tot;
      }
      if (any && hf == 1 && h == 0) { ex_list[j] = list; ex_cnt[j] = cnt; }
      psync();
      if (any && hf == 0 && h == 0 && amb) {
        const unsigned long long plist = ex_list[j];
        const int pcnt = ex_cnt[j];
        const int tot = cnt + pcnt;
        uint32_t cand = 255u << 24, cand_hi = 0u;
        if (tot <= 7 && tot >= 1 && t1 == t1) {
          const unsigned long long all = (list & ((1ull << (8 * cnt)) - 1ull)) | (plist << (8 * cnt));
          cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
          cand_hi = (uint32_t)(all >> 24);
        }
        slice[atomicAdd(qnp, 1)] = SplitEntry{(int32_t)(crow0 + px), cand, cand_hi};
      }
      zero_acc();
      // (the next tile's first HSGK_PAIR_STEP starts with a barrier: the exchange area is read before it is overwritten)
    }
    HSGK_VMWAIT4(0, preA);
    HSGK_VMWAIT4(0, preB);
    if constexpr (DEPTH == 4) { HSGK_VMWAIT4(0, preC); HSGK_VMWAIT4(0, preD); }
#undef HSGK_PAIR_STEP
#undef HSGK_VMWAIT4
    r += nrows;
  }
  __syncthreads();
  if (tid == 0) seg.count[blockIdx.x] = qnp[0];
}

// ---------------------------------------------------------------------------
// 128 < K <= 256, rows of four 64-column chunks, from the TILE-ORDERED copy (score_tiles_f16t.h): NO row traffic
// through LDS and NO synchronisation between waves.  A wave loads its 32-row tile (16 KiB contiguous, sixteen 1-KiB
// loads) straight into the registers the MFMA takes its B operand from, keeps it there, and scores it against the
// resident hi plane of the table (143 KB of LDS) in FOUR passes of 64 centroids -- two accumulator sets instead of
// the eight a one-pass kernel needs (assign_half_wide1_kernel: one wave per SIMD) or the pair synchronisations the
// shared-window kernel pays (assign_half_pair_kernel: twelve per tile, matrix pipe 28 % busy): 2 x 64 row registers
// (this tile + the next one in flight), 32 accumulators, two waves per SIMD, every wave on its own.
// After a pass the 32 scores of a lane collapse to the pass's tagged top-3 (position among the lane's 32 scores
// in the low five mantissa bits: <= 31 ulp <= 3.7e-6, in the gap), the tag and three v_med3 per score, and the accumulators are
// free for the next pass.  At the end of the tile the row's best / second best come out of the 4 x 2 kept values of
// both lane halves; an undecided row's candidate set {k : score >= best - gap} is exactly the kept values above
// the threshold, unless some pass's THIRD value reaches it too -- then the entry asks for all K centroids (the
// whole-wave path of the exact pass).  Same approximation, gap and queue format as the kernels above: the labels
// stay those of the exact argmax.
__global__ __launch_bounds__(512) void assign_half_t256_kernel(
    const _Float16 *__restrict__ xmT, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, SegQueue seg, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NW = 8, TPX = NW * 32, DM = 256, RS = DM + 16 + 8, NKB = DM / 16, NG = 4;
  const uint16_t *chs = reinterpret_cast<const uint16_t *>(lds_raw);            // [256][RS] hi plane of the table
  int *qnp = reinterpret_cast<int *>(lds_raw + (size_t)256 * RS * 2);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const bool has_tail = d > DM;
  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TPX - 1) / ((int64_t)gridDim.x * TPX) * TPX;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (tid == 0) { seg.count[blockIdx.x] = 0; seg.row0[blockIdx.x] = r < r_end ? r : 0; qnp[0] = 0; }
#ifdef HSGK_T256_TURNS
  // EXPERIMENT (build with -DHSGK_T256_TURNS; measured neutral, profiles/r05_turns_ab.txt: 9.16 / 9.36 ms against
  // 9.13 / 9.11 at 4 x 768^2): the two waves of a SIMD take TURNS on the matrix pipe -- a wave holds its SIMD's turn
  // for the matrix instructions of a pass and gives it up before the pass's epilogue, so that one wave's ~1 100
  // cycles of MFMA would always run beside the other's ~700 cycles of vector work.  The counters had suggested the
  // two pipes idle in turn (matrix 43 %, vector 45 % busy: their SUM fills the kernel, profiles/r05_cfg4_pmc.txt);
  // enforcing the alternation changes nothing, so the waves already interleave and the rest of the time is the
  // LDS operand stream (one ds_read_b128 per MFMA: 1 MB per CU and round of tiles) and the row loads.
  int *turn = qnp + 8;                                       // [4] one word per SIMD (0: free)
  if (tid < 4) turn[tid] = 0;
  // HW_ID bits 5:4 = SIMD the wave runs on (s_getreg_b32 hwreg(HW_REG_HW_ID, 4, 2))
  const int simd = __builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)) & 3);
#endif
  if (r >= r_end) return;
  SplitEntry *slice = gqueue + r;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  float errc_max = 0.0f;
  // +/- infinity the compiler cannot see through: v_med3_f32(a, b, +inf) IS max(a, b) in one instruction, while a
  // visible constant is folded to fmax / fmin, which add a canonicalising v_max per operand of unknown origin
  float PINF, NINF;
  asm volatile("s_mov_b32 %0, 0x7f800000" : "=s"(PINF));
  asm volatile("s_mov_b32 %0, 0xff800000" : "=s"(NINF));
  constexpr int64_t blk_bytes = (int64_t)32 * DM * 2;
  const uint32_t voff = (uint32_t)lane * 16u;
  const uint32_t toff = (uint32_t)(wu * 32 + j) * 8u;

  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)1 << 24);
    const int64_t crow0 = r;
    if (b != staged_img) {
      float m = 0.0f;
      for (int k = lane; k < K; k += 64) m = fmaxf(m, errc[(int64_t)b * K + k]);
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      errc_max = m;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                  // nobody still reads the previous table
      stage_half_planes<NW, 8, 1>(lds_raw, cent + (int64_t)b * K * d, d, K);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      staged_img = b;
    }
    const int ntile = max(0, (nrows - wu * 32 + TPX - 1) / TPX);
    const char *wbase = reinterpret_cast<const char *>(xmT) + ((crow0 >> 5) + wu) * blk_bytes;
    const char *tbase = reinterpret_cast<const char *>(xt + crow0);
    // the next tile's rows (and tail words) in flight while this one is scored; "+v": the set keeps its registers
    // from load to load (an "=v" output may be given fresh registers and copied into place before it has landed)
    u32x4 Bn[NKB];
    uint2 tailn = {0u, 0u};
#pragma unroll
    for (int i = 0; i < NKB; ++i) Bn[i] = u32x4{0u, 0u, 0u, 0u};
    auto issue = [&](int tile) {              // unclamped: the copy has kHalfSlackRows readable rows past its end
      const char *tb = wbase + (int64_t)tile * (NW * blk_bytes);
#pragma unroll
      for (int i = 0; i < NKB; ++i)
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(Bn[i]) : "v"(voff), "s"(tb + i * 1024));
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(tailn) : "v"(toff), "s"(tbase + (int64_t)tile * TPX * 8));
    };
    if (ntile > 0) issue(0);
    for (int tile = 0; tile < ntile; ++tile) {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(Bn[0]), "+v"(Bn[1]), "+v"(Bn[2]), "+v"(Bn[3]), "+v"(Bn[4]), "+v"(Bn[5]), "+v"(Bn[6]), "+v"(Bn[7]),
                     "+v"(Bn[8]), "+v"(Bn[9]), "+v"(Bn[10]), "+v"(Bn[11]), "+v"(Bn[12]), "+v"(Bn[13]), "+v"(Bn[14]),
                     "+v"(Bn[15]), "+v"(tailn));
      u32x4 Bc[NKB];
#pragma unroll
      for (int i = 0; i < NKB; ++i) { Bc[i] = Bn[i]; asm volatile("" : "+v"(Bc[i])); }
      uint2 tailc = tailn;
      asm volatile("" : "+v"(tailc));
      __builtin_amdgcn_sched_barrier(0);
      issue(tile + 1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 tv = {h == 0 ? tailc.x : 0u, 0u, 0u, 0u};
      const float err = __uint_as_float(tailc.y);

      const float gap = half_wide_gap(err, errc_max) + 8.0e-6f;      // + 2 x the tag perturbation
      float g1[NG], g2[NG], g3[NG];
      // A pass whose THIRD value comes within the gap of the lane's best so far cannot be described by its top 2
      // (the seed-grid centroids of the first iteration: a pixel near a cell corner has four near-equal neighbours,
      // all in one pass and one lane half).  Such a pass is scanned again while its scores still sit in the
      // accumulators and the lane keeps the LIST of its scores above (best so far - gap) -- a superset of what the
      // final threshold admits; one listed pass per lane, a second one falls back to "all K centroids".
      float rbest = NINF;                       // the lane's best over the passes so far
      unsigned long long xlist = 0;
      int xcnt = 0, xpass = -1;
      bool xover = false;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        g1[g] = NINF; g2[g] = NINF; g3[g] = NINF;
        if (g * 64 >= K) continue;                                   // (uniform)
        // ---- scores of table blocks 2 g, 2 g + 1: 16 k-blocks + the tail k-block, operands one k-block ahead
        f32x16 acc[2];
        struct Ops { f16x8 a[2]; };
        const uint16_t *hp = chs + (g * 64 + j) * RS + 8 * h;
        auto ld = [&](int col0, Ops &o) {
          o.a[0] = *reinterpret_cast<const f16x8 *>(hp + col0);
          o.a[1] = *reinterpret_cast<const f16x8 *>(hp + 32 * RS + col0);
        };
        auto mm = [&](const Ops &o, const u32x4 &bv, bool first) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const f16x8 bb = __builtin_bit_cast(f16x8, bv);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.a[0], bb, first ? zero : acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.a[1], bb, first ? zero : acc[1], 0, 0, 0);
        };
        // operands TWO k-blocks ahead (three sets): with one set ahead every k-block's first MFMA waited for the
        // LDS reads issued just before it (two MFMAs = 64 cycles of matrix work do not cover an LDS round trip)
        Ops o[3];
#ifdef HSGK_T256_TURNS
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0)
          while (atomicCAS(turn + simd, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        ld(0, o[0]);
        ld(16, o[1]);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          __builtin_amdgcn_sched_barrier(0);
          if (kb + 2 <= NKB) ld((kb + 2) * 16, o[(kb + 2) % 3]);      // (kb + 2 == NKB: the tail k-block's columns)
          mm(o[kb % 3], Bc[kb], kb == 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (has_tail) mm(o[NKB % 3], tv, false);
#ifdef HSGK_T256_TURNS
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) __hip_atomic_store(turn + simd, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- tagged top-3 of the lane's 32 scores of this pass
        float a1 = NINF, a2 = NINF, a3 = NINF;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int mg = 2 * g + m;
          if (mg * 32 >= K) continue;                                // (uniform)
          const bool full = (mg + 1) * 32 <= K;
          // (a3 <= a2 <= a1 throughout: the median of {a2, v, a3} IS the new third -- a2 when v displaces it, v when
          //  v lands between a3 and a2, a3 otherwise; one instruction instead of min(a2, v) + max(a3, .))
          // Two loops under a REAL branch: written as one loop with `if (!full) v = k < K ? v : -inf` the compiler
          // if-converted the mask into every block -- two v_cndmask per score, their 128 lane masks spilled to
          // VGPR lanes and read back with two v_readlane each: eight vector instructions per score instead of four.
          if (full) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const float v = __uint_as_float((__float_as_uint(acc[m][rr]) & ~31u) | (uint32_t)(m * 16 + rr));
              a3 = __builtin_amdgcn_fmed3f(a2, v, a3);
              a2 = __builtin_amdgcn_fmed3f(a1, a2, v);
              a1 = __builtin_amdgcn_fmed3f(a1, v, PINF);                          // max
            }
          } else {
            asm volatile("" ::: "memory");                            // (keeps the ragged last block a branch of its own)
            const int rem = K - mg * 32 - 4 * h;                       // score rr exists iff (rr & 3) + 8 (rr >> 2) < rem
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              float v = __uint_as_float((__float_as_uint(acc[m][rr]) & ~31u) | (uint32_t)(m * 16 + rr));
              v = (rr & 3) + 8 * (rr >> 2) < rem ? v : NINF;
              a3 = __builtin_amdgcn_fmed3f(a2, v, a3);
              a2 = __builtin_amdgcn_fmed3f(a1, a2, v);
              a1 = __builtin_amdgcn_fmed3f(a1, v, PINF);
            }
          }
        }
        g1[g] = a1; g2[g] = a2; g3[g] = a3;
        rbest = __builtin_amdgcn_fmed3f(rbest, a1, PINF);
        const bool third = a3 >= rbest - gap;
        if (__any(third)) {
          const float lthr = third ? rbest - gap : PINF;
          unsigned long long sl = 0;
          int sc = 0;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int mg = 2 * g + m;
            if (mg * 32 >= K) continue;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const uint32_t k = mg * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
              const float v = __uint_as_float((__float_as_uint(acc[m][rr]) & ~31u) | (uint32_t)(m * 16 + rr));
              const bool hit = k < (uint32_t)K && v >= lthr;
              sl = hit ? ((sl << 8) | k) : sl;
              sc += hit ? 1 : 0;
            }
          }
          if (third) {
            xover = xover || xpass >= 0 || sc > 8;
            xlist = sl; xcnt = sc; xpass = g;
          }
        }
      }
      // ---- the row: best / second best over the passes and both lane halves
      auto index_of = [&](float v, int g) {
        const uint32_t tg = __float_as_uint(v) & 31u;
        return (2 * g + (int)(tg >> 4)) * 32 + (int)(tg & 3u) + 8 * (int)((tg >> 2) & 3u) + 4 * h;
      };
      float t1 = NINF, t2 = NINF;
      int tg = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        tg = g1[g] > t1 ? g : tg;
        t2 = __builtin_amdgcn_fmed3f(t1, t2, g1[g]);
        t1 = __builtin_amdgcn_fmed3f(t1, g1[g], PINF);
        t2 = __builtin_amdgcn_fmed3f(t1, t2, g2[g]);
      }
      int ti = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) ti = tg == g ? index_of(t1, g) : ti;
      {
        const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
        const int oi = __shfl_xor(ti, 32);
        if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
        else { t2 = fmaxf(o1, t2); }
      }
      const int px = tile * TPX + wu * 32 + j;
      const bool valid = px < nrows;
#if defined(HSGK_T256_DEBUG) && (HSGK_T256_DEBUG & 2)          // probe: every row goes to the exact pass
      const bool amb = valid;
#else
      const bool amb = valid && !(t1 - t2 > gap);                    // ambiguous (or NaN)
#endif
      if (h == 0 && valid) put_label(klab, crow0 + px, ti);          // provisional for ambiguous rows
      if (!__any(amb)) continue;
      // candidates: the kept values of both lane halves that reach the threshold (tagged values against a
      // threshold from a tagged best: the 2 x 3.7e-6 of the tags are in the gap); a pass whose third value
      // reaches it as well may hold more -> all K centroids
      const float thr = amb ? t1 - gap : INFINITY;
      unsigned long long list = 0;
      int cnt = 0;
#if defined(HSGK_T256_DEBUG) && (HSGK_T256_DEBUG & 1)          // probe: every queued row asks for all K centroids
      bool over = true;
#else
      bool over = xover;
#endif
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g * 64 >= K) continue;
        if (xpass == g) {                          // the listed pass: all of its list, if the pass matters at all
          const bool use = g1[g] >= thr;
          const bool take = use && cnt + xcnt <= 7;                  // (more than seven in all: all K anyway)
          over = over || (use && !take);
          const int sh = take ? 8 * xcnt : 0;                        // <= 56
          list = take ? ((list << sh) | (xlist & ((1ull << sh) - 1ull))) : list;
          cnt += take ? xcnt : 0;
        } else {
          const bool h1 = g1[g] >= thr, h2 = g2[g] >= thr;
          over = over || g3[g] >= thr;
          list = h1 ? ((list << 8) | (unsigned long long)index_of(g1[g], g)) : list;
          cnt += h1 ? 1 : 0;
          list = h2 ? ((list << 8) | (unsigned long long)index_of(g2[g], g)) : list;
          cnt += h2 ? 1 : 0;
        }
      }
      over = over || cnt > 8;
      const unsigned long long olist = __shfl_xor(list, 32);
      const int ocnt = __shfl_xor(cnt, 32);
      const int oover = __shfl_xor((int)over, 32);       // (unconditional: a short-circuited shuffle would read inactive lanes)
      over = over || oover != 0;
      const int tot = cnt + ocnt;
      uint32_t cand = 255u << 24, cand_hi = 0u;
      if (!over && tot <= 7 && tot >= 1 && t1 == t1) {
        const unsigned long long all = (list & ((1ull << (8 * cnt)) - 1ull)) | (olist << (8 * cnt));
        cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
        cand_hi = (uint32_t)(all >> 24);
      }
      if (h == 0 && amb) slice[atomicAdd(qnp, 1)] = SplitEntry{(int32_t)(crow0 + px), cand, cand_hi};
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(Bn[0]), "+v"(Bn[1]), "+v"(Bn[2]), "+v"(Bn[3]), "+v"(Bn[4]), "+v"(Bn[5]), "+v"(Bn[6]), "+v"(Bn[7]),
                   "+v"(Bn[8]), "+v"(Bn[9]), "+v"(Bn[10]), "+v"(Bn[11]), "+v"(Bn[12]), "+v"(Bn[13]), "+v"(Bn[14]),
                   "+v"(Bn[15]), "+v"(tailn));
    r += nrows;
  }
  __syncthreads();
  if (tid == 0) seg.count[blockIdx.x] = qnp[0];
}

// ---------------------------------------------------------------------------
// K <= 256, d / 64 == 4: the TABLE IN REGISTERS.  The kernels above keep the 256 x d fp16 table in LDS (143 KB)
// and move a 32-row window per pair of waves through what is left of it -- two pair synchronisations per
// 64-column chunk, twelve per 32-row tile, the matrix pipe 28 % busy.  Here a wave OWNS 32 centroids: their fp16
// rows are the MFMA A operand, resident in 68 registers for a whole image, so the LDS is free for the ROWS --
// 128-row tiles, double buffered, staged by contiguous 16-byte copies of the fp16 copy, read by all eight waves
// as B operands.  A wave scores the tile's four 32-row blocks against its 32 centroids (68 MFMAs), leaves each
// row's tagged top-2 of its block in LDS, and after one barrier every wave merges the eight blocks of a row in
// block order (the lower block wins a tie, as the single-pass kernels do), so that every wave knows the row's
// decision and threshold and appends the candidates of ITS block for the undecided rows (block maximum /
// runner-up rule, HalfWideEpi).  Three workgroup barriers per 128 rows instead of twelve pair syncs per 32.
// Same approximation (fp16 hi rows x fp16 hi table, fp32 accumulation over 16-column blocks), same gap, same
// queue format as assign_half_pair_kernel: the labels stay those of the exact argmax.
// MEASURED SLOWER than the pair kernel and therefore opt-in (HSGK_WIDE2=regs): 0.65-0.75 ms per launch against
// 0.527 at cfg4.  rocprofv3: the matrix pipe is busy the same 320 M cycles, but a 128-row interval takes ~20 K
// cycles for 4.4 K of MFMA work per SIMD -- 59 % of the wave cycles in s_waitcnt, LDS 41 % busy (42 % of that in
// bank conflicts the row layout should not have), three barriers with all eight waves in the same phase, and every
// wave merging every row.  A variant that let the accumulators die per block (tagged top-3, a third candidate in
// one lane = all centroids) sent 65 K rows per iteration to the all-K exact path (the pair kernel: 574).
__global__ __launch_bounds__(512, 2) void assign_half_regs_kernel(
    const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d,
    const float *__restrict__ cent, const float *__restrict__ errc, int K,
    const int64_t *__restrict__ img_row0, int B, int32_t *__restrict__ klab,
    SplitEntry *__restrict__ gqueue, SegQueue seg, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int DM = 256, RS = DM + 8, TR = 128, NSUB = 4, KS = DM / 16;
  constexpr int kTileBytes = TR * RS * 2;
  uint16_t *tiles = reinterpret_cast<uint16_t *>(lds_raw);                       // [2][TR][RS]
  uint2 *tails = reinterpret_cast<uint2 *>(lds_raw + 2 * kTileBytes);            // [2][TR]
  float4 *part = reinterpret_cast<float4 *>(tails + 2 * TR);                     // [NSUB][8][32] (p1, p2, index)
  int *ccnt = reinterpret_cast<int *>(part + NSUB * 8 * 32);                     // [TR] candidates of a row
  unsigned char *clist = reinterpret_cast<unsigned char *>(ccnt + TR);           // [TR][8]
  int *qnp = reinterpret_cast<int *>(clist + TR * 8);                            // [0] queue length of the workgroup
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, g = lane >> 5;
  const int wu = __builtin_amdgcn_readfirstlane(w);

  const int64_t N = meta->n_rows;
  const int64_t per = (N + (int64_t)gridDim.x * TR - 1) / ((int64_t)gridDim.x * TR) * TR;
  int64_t r = (int64_t)blockIdx.x * per;
  const int64_t r_end = min(N, r + per);
  if (tid == 0) { seg.count[blockIdx.x] = 0; seg.row0[blockIdx.x] = r < r_end ? r : 0; qnp[0] = 0; }
  if (tid < TR) ccnt[tid] = 0;
  if (r >= r_end) return;
  SplitEntry *slice = gqueue + r;
  int b = image_of_row(img_row0, B, r);
  int staged_img = -1;
  float errc_max = 0.0f;
  const bool has_tail = d > DM;
  const bool mine = 32 * wu < K;                       // (waves whose block lies past K only take part in the barriers)
  f16x8 ta[KS + 1];                                    // this wave's 32 centroids: lane (i, g) holds C[32 w + i][16 s + 8 g ..]
#pragma unroll
  for (int s2 = 0; s2 <= KS; ++s2) ta[s2] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};

  while (r < r_end) {
    while (img_row0[b + 1] <= r) ++b;
    const int64_t seg_end = min(r_end, img_row0[b + 1]);
    const int nrows = (int)min(seg_end - r, (int64_t)1 << 24);
    const int64_t crow0 = r;
    if (b != staged_img) {
      float m = 0.0f;
      for (int k = lane; k < K; k += 64) m = fmaxf(m, errc[(int64_t)b * K + k]);
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      errc_max = m;
      const int kc = 32 * wu + j;
      const float *crow = cent + ((int64_t)b * K + min(kc, K - 1)) * d;
      const bool live = kc < K;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        const float4 v0 = *reinterpret_cast<const float4 *>(crow + 16 * s2 + 8 * g);
        const float4 v1 = *reinterpret_cast<const float4 *>(crow + 16 * s2 + 8 * g + 4);
        uint32_t h[4], lo;
        f16_split2(v0.x, v0.y, h[0], lo);
        f16_split2(v0.z, v0.w, h[1], lo);
        f16_split2(v1.x, v1.y, h[2], lo);
        f16_split2(v1.z, v1.w, h[3], lo);
        const u32x4 hv = {live ? h[0] : 0u, live ? h[1] : 0u, live ? h[2] : 0u, live ? h[3] : 0u};
        ta[s2] = __builtin_bit_cast(f16x8, hv);
      }
      {                                                // the two location columns: k = 0, 1 of one more k-block
        uint32_t h0 = 0u, lo;
        if (has_tail && live && g == 0) f16_split2(crow[DM], crow[DM + 1], h0, lo);
        const u32x4 hv = {h0, 0u, 0u, 0u};
        ta[KS] = __builtin_bit_cast(f16x8, hv);
      }
      staged_img = b;
    }
    const int ntile = (nrows + TR - 1) / TR;
    // ---- staging: a tile is TR contiguous rows of the fp16 copy (kHalfSlackRows readable rows past the end).
    //      (Macros, not lambdas: captured by a lambda the eight staging registers went to scratch memory -- stored
    //       right behind their loads, i.e. every tile waited for its own prefetch.)
#define HSGK_REGS_LOAD(TILE)                                                                              \
  {                                                                                                       \
    const uint4 *src_ = reinterpret_cast<const uint4 *>(xm + (crow0 + (int64_t)(TILE) * TR) * DM);        \
    pre0 = src_[tid]; pre1 = src_[tid + 512]; pre2 = src_[tid + 1024]; pre3 = src_[tid + 1536];           \
    pre4 = src_[tid + 2048]; pre5 = src_[tid + 2560]; pre6 = src_[tid + 3072]; pre7 = src_[tid + 3584];   \
    if (tid < TR) pret = xt[crow0 + (int64_t)(TILE) * TR + tid];                                          \
  }
#define HSGK_REGS_PUT(P, U) *reinterpret_cast<uint4 *>(dst_ + ((tid + 512 * (U)) >> 5) * RS + 8 * ((tid + 512 * (U)) & 31)) = P
#define HSGK_REGS_STORE(BUF)                                                                              \
  {                                                                                                       \
    uint16_t *dst_ = tiles + (BUF) * (TR * RS);                                                           \
    HSGK_REGS_PUT(pre0, 0); HSGK_REGS_PUT(pre1, 1); HSGK_REGS_PUT(pre2, 2); HSGK_REGS_PUT(pre3, 3);       \
    HSGK_REGS_PUT(pre4, 4); HSGK_REGS_PUT(pre5, 5); HSGK_REGS_PUT(pre6, 6); HSGK_REGS_PUT(pre7, 7);       \
    if (tid < TR) tails[(BUF) * TR + tid] = pret;                                                         \
  }
    uint4 pre0, pre1, pre2, pre3, pre4, pre5, pre6, pre7;
    uint2 pret = {0u, 0u};
    HSGK_REGS_LOAD(0)
    __syncthreads();                                   // nobody still reads the buffers of the previous pass
    HSGK_REGS_STORE(0)
    if (ntile > 1) HSGK_REGS_LOAD(1)
    for (int tile = 0; tile < ntile; ++tile) {
      const int buf = tile & 1;
      __syncthreads();                                 // B0: tile staged; the previous tile's partials and lists are consumed
      const uint16_t *tb = tiles + buf * (TR * RS);
      const uint2 *tt = tails + buf * TR;
      // ---- 4 x (16 + 1) MFMAs: block (rows 32 u .., this wave's 32 centroids); operand reads four k-steps ahead
      f32x16 acc[NSUB];
      float bm1[NSUB], bm2[NSUB];
      if (mine) {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) acc[u][rr] = 0.0f;
          const uint16_t *bp = tb + (32 * u + j) * RS + 8 * g;
          f16x8 bv[KS];
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) bv[s2] = *reinterpret_cast<const f16x8 *>(bp + 16 * s2);
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[s2], bv[s2], acc[u], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int s2 = 0; s2 < KS - 4; ++s2) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          if (has_tail) {
            const u32x4 tv = {g == 0 ? tt[32 * u + j].x : 0u, 0u, 0u, 0u};
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ta[KS], __builtin_bit_cast(f16x8, tv), acc[u], 0, 0, 0);
          }
        }
        // ---- tagged top-2 of the block per row (lane = row, 16 centroids per lane half), halves merged
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          float b1 = -INFINITY, b2 = -INFINITY;
          if (32 * (wu + 1) <= K) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const float v = __uint_as_float((__float_as_uint(acc[u][rr]) & ~15u) | (uint32_t)rr);
              b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
              b1 = max_raw(b1, v);
            }
          } else {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
              const int k = 32 * wu + (rr & 3) + 8 * (rr >> 2) + 4 * g;
              float v = __uint_as_float((__float_as_uint(acc[u][rr]) & ~15u) | (uint32_t)rr);
              v = k < K ? v : -INFINITY;
              b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
              b1 = max_raw(b1, v);
            }
          }
          bm1[u] = b1;
          bm2[u] = b2;
          const uint32_t tg = __float_as_uint(b1) & 15u;
          int ti = 32 * wu + (int)(tg & 3u) + 8 * (int)(tg >> 2) + 4 * g;
          float t1 = b1, t2 = b2;
          const float o1 = __shfl_xor(t1, 32), o2 = __shfl_xor(t2, 32);
          const int oi = __shfl_xor(ti, 32);
          if (o1 > t1 || (o1 == t1 && oi < ti)) { t2 = fmaxf(t1, o2); t1 = o1; ti = oi; }
          else { t2 = fmaxf(o1, t2); }
          if (g == 0) part[(u * 8 + wu) * 32 + j] = make_float4(t1, t2, __int_as_float(ti), 0.0f);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
          bm1[u] = bm2[u] = -INFINITY;
          if (g == 0) part[(u * 8 + wu) * 32 + j] = make_float4(-INFINITY, -INFINITY, __int_as_float(0x7fffffff), 0.0f);
        }
      }
      __syncthreads();                                 // B1: the eight blocks' top-2 of every row are in LDS
      // ---- every wave merges every row (block order: the lower block wins a tie) and appends ITS candidates
      bool ambu[NSUB];
      int tiu[NSUB];
#pragma unroll
      for (int u = 0; u < NSUB; ++u) {
        float t1 = -INFINITY, t2 = -INFINITY;
        int ti = 0;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) {
          const float4 pv = part[(u * 8 + ww) * 32 + j];
          const bool up = pv.x > t1;
          t2 = up ? fmaxf(t1, pv.y) : fmaxf(t2, pv.x);
          ti = up ? __float_as_int(pv.z) : ti;
          t1 = up ? pv.x : t1;
        }
        const int px = tile * TR + 32 * u + j;
        const bool valid = px < nrows;
        const float err = __uint_as_float(tt[32 * u + j].y);
        const float gap = half_wide_gap(err, errc_max) + 4.0e-6f;
        const bool amb = valid && !(t1 - t2 > gap);
        ambu[u] = amb;
        tiu[u] = ti;
        if (wu == 0 && g == 0 && valid) put_label(klab, crow0 + px, ti);   // provisional for ambiguous rows
        if (!mine || !__any(amb)) continue;
        const float thr = amb ? t1 - gap - 2.0e-6f : INFINITY;
        const float thr2 = thr - 2.0e-6f;
        // this lane's candidates: the block maximum by its tag, or -- its runner-up reaches the threshold as well
        // in some lane -- a scan of the lane's 16 scores (up to four kept; more: the row asks for all centroids)
        uint32_t hits = 0u;
        int nh = 0;
        if (!__any(bm2[u] >= thr2)) {
          const uint32_t tgm = __float_as_uint(bm1[u]) & 15u;
          const bool hit = bm1[u] >= thr2;
          hits = 32 * wu + (tgm & 3u) + 8 * (tgm >> 2) + 4 * g;
          nh = hit ? 1 : 0;
        } else {
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            const uint32_t k = 32 * wu + (rr & 3) + 8 * (rr >> 2) + 4 * g;
            const bool hit = k < (uint32_t)K && acc[u][rr] >= thr;
            hits = (hit && nh < 4) ? (hits | (k << (8 * nh))) : hits;
            nh += hit ? 1 : 0;
          }
        }
        if (nh > 0) {
          const int pos = atomicAdd(&ccnt[32 * u + j], nh > 4 ? 8 : nh);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nh && pos + q < 8) clist[(32 * u + j) * 8 + pos + q] = (unsigned char)((hits >> (8 * q)) & 255u);
        }
      }
      __syncthreads();                                 // B2: the candidate lists are complete
      // ---- wave u composes the queue entries of block u and clears its counters
      if (wu < NSUB) {
        bool amb = false;
        int ti = 0;
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
          if (u == wu) { amb = ambu[u]; ti = tiu[u]; }
        const int row = 32 * wu + j;
        if (g == 0) {
          const int tot = ccnt[row];
          if (amb) {
            uint32_t cand = 255u << 24, cand_hi = 0u;
            if (tot >= 1 && tot <= 7) {
              const unsigned long long all = *reinterpret_cast<const unsigned long long *>(clist + row * 8) &
                                             ((1ull << (8 * tot)) - 1ull);
              cand = (uint32_t)(all & 0xFFFFFFull) | ((uint32_t)tot << 24);
              cand_hi = (uint32_t)(all >> 24);
            }
            slice[atomicAdd(qnp, 1)] = SplitEntry{(int32_t)(crow0 + tile * TR + row), cand, cand_hi};
          }
          ccnt[row] = 0;
        }
        (void)ti;
      }
      // ---- the next tile into the other buffer (last read before B0 of this iteration), the one after into registers
      if (tile + 1 < ntile) HSGK_REGS_STORE(buf ^ 1)
      if (tile + 2 < ntile) HSGK_REGS_LOAD(tile + 2)
    }
#undef HSGK_REGS_LOAD
#undef HSGK_REGS_PUT
#undef HSGK_REGS_STORE
    r += nrows;
  }
  __syncthreads();
  if (tid == 0) seg.count[blockIdx.x] = qnp[0];
}

// exact pass over the per-workgroup slices of the queue (nseg <= 1024)
__global__ __launch_bounds__(256) void assign_requeue_seg_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    int32_t *__restrict__ klab, const SplitEntry *__restrict__ gqueue, SegQueue seg, int nseg,
    const int64_t *__restrict__ img_row0, int B, const HardList hl) {
  __shared__ int pre[1025];
  __shared__ int wsum[4];
  __shared__ __attribute__((aligned(16))) float exact_win[4 * kExactStageFloats];
  __shared__ int32_t hard_buf[4][2][kHardBuf];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  HardBuf hb{hard_buf[w][0], hard_buf[w][1], 0};
  // exclusive prefix of the segment counts (four per thread)
  int c[4], tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int sidx = tid * 4 + i; c[i] = sidx < nseg ? seg.count[sidx] : 0; tot += c[i]; }
  int inc = tot;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int i = 0; i < w; ++i) base += wsum[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) { pre[tid * 4 + i] = base; base += c[i]; }
  if (tid == 255) pre[1024] = base;
  __syncthreads();
  const int total = pre[1024];
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const int grp = lane >> 2;
  for (int e0 = wave * 16; e0 < total; e0 += nwaves * 16) {
    const int e = e0 + grp;
    SplitEntry ent{0, 0u, 0u};
    if (e < total) {
      int lo = 0, hi = 1024;                       // last segment with pre[seg] <= e
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= e) lo = mid; else hi = mid;
      }
      ent = gqueue[seg.row0[lo] + (e - pre[lo])];
    }
    int img = 0;
    {
      int hi = B;
      while (hi - img > 1) {
        const int mid = (img + hi) >> 1;
        if (img_row0[mid] <= (int64_t)ent.row) img = mid; else hi = mid;
      }
    }
    exact_rescore16(ent, e < total, img, x, d, cent, K, klab, exact_win + w * kExactStageFloats, hl, &hb);
  }
  if (hl.rows && hb.n) hard_flush(hb, hl, x, d, cent, K, klab);
}

static bool wide1_fits(int d) { return (d / 64 == 4 || d / 64 == 2) && half_lds_bytes<4, 8, 1, 1>(d) + 64 <= 160 * 1024; }

bool assign_half_wide2_eligible(int d, int K) {
  return K > 128 && K <= 256 && half_wide_shape_ok(d) &&
         half_lds_bytes<8, 4, 1, 1>(d) + (size_t)kSplitLdsList * 6 <= 160 * 1024;
}

// (the conditions under which launch_assign_half_wide2 below runs assign_half_t256_kernel when it is handed a
//  tile-ordered copy: api.hip then has the prep kernel write that copy only)
bool assign_half_wide2_tiles(int d, int K, int max_chunks) {
  if (!assign_half_wide2_eligible(d, K) || d / 64 != 4 || !wide1_fits(d) || max_chunks <= 0) return false;
  const char *two = getenv("HSGK_WIDE2");
  if (two) return false;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  (void)hipGetLastError();
  const int64_t max_tiles = ((int64_t)max_chunks * HSGK_CHUNK + 255) / 256;
  const int64_t grid = max_tiles < cus ? max_tiles : cus;
  return grid <= 1024;
}

// state: [rows] 16-byte scratch records
int launch_assign_half_wide2(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent,
                             float *errc, int K, int B, const ChunkTable &t, int max_chunks, int32_t *klab,
                             void *state, void *qrows, int32_t *qcount, const hsgk_segkm_meta *meta,
                             hipStream_t s, const _Float16 *xmT, bool table_ready, int32_t *hard_rows,
                             int32_t *hard_count, int64_t hard_cap) {
  if (max_chunks <= 0 || B <= 0) return 0;
  HardList hl{nullptr, nullptr, 0, 0, nullptr};
  if (hard_rows && hard_count && hard_rows_enabled() && hard_rows_fit(d)) {
    hl = HardList{hard_rows, hard_count, hard_cap, hard_skip(B), table_ready && hard_prev_enabled() ? hard_count + (size_t)(B + 1) * kHardStride : nullptr};
  }
  // (also with the lists switched off: hsgk_lloyd_requeued_rows sums these counters)
  if (!table_ready && hard_count) HSGK_CHECK_HIP(hipMemsetAsync(hard_count, 0, sizeof(int32_t) * (size_t)B * kHardStride, s));
  constexpr int NW = 8, TPX = NW * 32, MB = 4;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  const int64_t max_tiles = ((int64_t)max_chunks * HSGK_CHUNK + TPX - 1) / TPX;
  const int grid = (int)(max_tiles < n_cu ? max_tiles : n_cu);
  if (!table_ready) {                      // (inside the Lloyd loop launch_finalize_fx has measured errc and zeroed qcount)
    hipLaunchKernelGGL(centroid_half_err_kernel, dim3((unsigned)(((int64_t)B * K + 3) / 4)), dim3(256), 0, s,
                       cent, d, (int64_t)B * K, errc);
    HSGK_LAUNCH_CHECK();
  }
  const char *two = getenv("HSGK_WIDE2");               // "two": the two-half kernel (A/B; read per call)
  if (wide1_fits(d) && !(two && two[0] == 't') && grid <= 1024 &&
      (int64_t)max_chunks * HSGK_CHUNK * 16 >= (int64_t)grid * 16) {
    // one pass over the rows, all 256 centroids resident (see assign_half_wide1_kernel); the per-workgroup
    // entry counts and slice starts live at the head of the (otherwise unused) state records
    constexpr int NW1 = 4, MB1 = 8, TPX1 = NW1 * 32;
    const int64_t tiles1 = ((int64_t)max_chunks * HSGK_CHUNK + TPX1 - 1) / TPX1;
    const int grid1 = (int)(tiles1 < n_cu ? tiles1 : n_cu);
    SegQueue seg{reinterpret_cast<int32_t *>(state), reinterpret_cast<int64_t *>(static_cast<char *>(state) + 4096)};
    const size_t lds = half_lds_bytes<NW1, MB1, 1, 1>(d) + 64;
    if (xmT && d / 64 == 4 && !two) {
      // the rows in tile order: every wave on its own, four passes of 64 centroids over register-resident rows
      const int gridt = (int)(max_tiles < n_cu ? max_tiles : n_cu);
      const size_t ldst = (size_t)256 * (256 + 24) * 2 + 64;
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(assign_half_t256_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst));
      hipLaunchKernelGGL(assign_half_t256_kernel, dim3(gridt), dim3(512), ldst, s, xmT, xt, d, cent, errc, K, t.img_row0, B,
                         klab, reinterpret_cast<SplitEntry *>(qrows), seg, meta);
      HSGK_LAUNCH_CHECK();
      hipLaunchKernelGGL(assign_requeue_seg_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                         reinterpret_cast<const SplitEntry *>(qrows), seg, gridt, t.img_row0, B, hl);
      HSGK_LAUNCH_CHECK();
      if (int rc = launch_assign_hard_rows(x, d, cent, K, B, hl, klab, s)) return rc;
      return 0;
    }
    if (d / 64 == 4 && two && two[0] == 'r') {
      // HSGK_WIDE2=regs: the table in registers, 128-row tiles through LDS (A/B: slower than the pair kernel)
      constexpr int TRr = 128;
      const int64_t tilesr = ((int64_t)max_chunks * HSGK_CHUNK + TRr - 1) / TRr;
      const int gridr = (int)(tilesr < n_cu ? tilesr : n_cu);
      const size_t ldsr = (size_t)2 * TRr * (256 + 8) * 2 + 2 * TRr * 8 + 4 * 8 * 32 * 16 + TRr * 4 + TRr * 8 + 64;
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(assign_half_regs_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr));
      hipLaunchKernelGGL(assign_half_regs_kernel, dim3(gridr), dim3(512), ldsr, s, xm, xt, d, cent, errc, K, t.img_row0, B,
                         klab, reinterpret_cast<SplitEntry *>(qrows), seg, meta);
      HSGK_LAUNCH_CHECK();
      hipLaunchKernelGGL(assign_requeue_seg_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                         reinterpret_cast<const SplitEntry *>(qrows), seg, gridr, t.img_row0, B, hl);
      HSGK_LAUNCH_CHECK();
      if (int rc = launch_assign_hard_rows(x, d, cent, K, B, hl, klab, s)) return rc;
      return 0;
    }
    if (two && two[0] == 'o') {                       // "one": the four-wave kernel (one wave per SIMD)
      auto kern = d / 64 == 4 ? assign_half_wide1_kernel<NW1, 4, MB1, 4> : assign_half_wide1_kernel<NW1, 2, MB1, 2>;
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid1), dim3(NW1 * 64), lds, s, xm, xt, d, cent, errc, K, t.img_row0, B, klab,
                         reinterpret_cast<SplitEntry *>(qrows), seg, meta);
    } else {                                          // eight waves in pairs (two per SIMD), same LDS budget
      auto kern = d / 64 == 4 ? assign_half_pair_kernel<4> : assign_half_pair_kernel<2>;
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kern, dim3(grid1), dim3(512), lds, s, xm, xt, d, cent, errc, K, t.img_row0, B, klab,
                         reinterpret_cast<SplitEntry *>(qrows), seg, meta);
    }
    HSGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(assign_requeue_seg_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                       reinterpret_cast<const SplitEntry *>(qrows), seg, grid1, t.img_row0, B, hl);
    HSGK_LAUNCH_CHECK();
    if (int rc = launch_assign_hard_rows(x, d, cent, K, B, hl, klab, s)) return rc;
    return 0;
  }
  if (!table_ready) HSGK_CHECK_HIP(hipMemsetAsync(qcount, 0, sizeof(int32_t), s));
  {
    const bool deep = ((d / 64) & 3) == 0;
    auto kern = deep ? assign_half_wide2_kernel<NW, 4, MB> : assign_half_wide2_kernel<NW, 2, MB>;
    const size_t lds = half_lds_bytes<NW, MB, 1, 1>(d) + (size_t)kSplitLdsList * 6;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, xm, xt, d, cent, errc, K, t.img_row0, B,
                       klab, reinterpret_cast<uint4 *>(state), reinterpret_cast<SplitEntry *>(qrows), qcount,
                       meta);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                     reinterpret_cast<const SplitEntry *>(qrows), qcount, t.img_row0, B, hl);
  HSGK_LAUNCH_CHECK();
  return launch_assign_hard_rows(x, d, cent, K, B, hl, klab, s);
}

// Level 2: grid (T, B); workgroup (t, b) takes a contiguous slice of image b's queue.
template <int NW>
__global__ __launch_bounds__(NW * 64) void assign_split_rows_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    const int32_t *__restrict__ q1, const int32_t *__restrict__ q1count, int64_t q1cap,
    int32_t *__restrict__ klab, SplitEntry *__restrict__ gqueue, int32_t *__restrict__ gcount) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int TPX = NW * 32;
  const int b = blockIdx.y;
  const int n = q1count[b];
  const int ntile = (n + TPX - 1) / TPX;
  const int per = (ntile + (int)gridDim.x - 1) / (int)gridDim.x * TPX;
  const int r0 = blockIdx.x * per;
  const int nrows = min(per, n - r0);
  if (nrows <= 0) return;
  unsigned char *tail = lds_raw + split_lds_bytes<NW>(d);   // same carve as assign_split_kernel ...
  int *qnp = reinterpret_cast<int *>(tail - 16);
  uint32_t *qcand = reinterpret_cast<uint32_t *>(tail);
  uint16_t *qpx = reinterpret_cast<uint16_t *>(qcand + kSplitLdsList);
  int32_t *rl = reinterpret_cast<int32_t *>(qpx + kSplitLdsList);   // ... + [NW][2][32] row-id slots
  const int32_t *list = q1 + (int64_t)b * q1cap + r0;
  if (threadIdx.x == 0) qnp[0] = 0;
  SplitEpi epi{K, nrows, b, 0, klab, qpx, qcand, qnp, gqueue, gcount, rl + (threadIdx.x >> 6) * 64};
  score_tiles_split<NW, 4, SplitEpi, true>(x, d, cent + (int64_t)b * K * d, K, 0, nrows, lds_raw, epi,
                                           true, list, rl);
  __syncthreads();
  const int qn = min(qnp[0], kSplitLdsList);
  if (qn > 0) {
    if (threadIdx.x == 0) qnp[1] = atomicAdd(gcount, qn);
    __syncthreads();
    const int base = qnp[1];
    for (int i = threadIdx.x; i < qn; i += NW * 64)
      gqueue[base + i] = SplitEntry{list[qpx[i]], qcand[i], 0u};
  }
}

// fp32 rows -> fp16 copy (RNE): main columns xm[rows][DM], xt[rows] = {packed tail
// columns, measured rounding error of the row} (layout and bound: score_tiles_f16.h);
// workgroup per chunk, wave per row
__global__ __launch_bounds__(256) void to_half_rows_kernel(
    const float *__restrict__ x, const int64_t *__restrict__ chunk_row0,
    const int32_t *__restrict__ chunk_rows, int d, _Float16 *__restrict__ xm,
    uint2 *__restrict__ xt, const hsgk_segkm_meta *__restrict__ meta) {
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int DM = half_main_cols(d), G = DM / 4;
  const int64_t row0 = chunk_row0[c];
  const int nr = chunk_rows[c];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  for (int r = w; r < nr; r += 4) {
    const float *src = x + (row0 + r) * d;
    float e2 = 0.0f;
    for (int q = lane; q < G; q += 64) {
      const f4u v = *reinterpret_cast<const f4u *>(src + 4 * q);
      const h4 hv = __builtin_convertvector(v, h4);
      *reinterpret_cast<h4 *>(xm + (row0 + r) * DM + 4 * q) = hv;
      const f4u e = v - __builtin_convertvector(hv, f4u);        // exact residuals
      e2 = fmaf(e.x, e.x, e2); e2 = fmaf(e.y, e.y, e2); e2 = fmaf(e.z, e.z, e2); e2 = fmaf(e.w, e.w, e2);
    }
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if (lane == 0) {
      const float t0 = d > DM ? src[DM] : 0.0f, t1 = d > DM + 1 ? src[DM + 1] : 0.0f;
      const h2 t = {(_Float16)t0, (_Float16)t1};
      const float e0 = t0 - (float)t[0], e1 = t1 - (float)t[1];
      e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2);
      xt[row0 + r] = make_uint2(__builtin_bit_cast(uint32_t, t), __float_as_uint(sqrtf(e2) * 1.0001f));
    }
  }
}

int launch_to_half_rows(const float *x, const ChunkTable &t, int max_chunks, int d, _Float16 *xm,
                        uint2 *xt, const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  hipLaunchKernelGGL(to_half_rows_kernel, dim3(max_chunks), dim3(256), 0, s, x, t.chunk_row0,
                     t.chunk_rows, d, xm, xt, meta);
  HSGK_LAUNCH_CHECK();
  return 0;
}

bool assign_half_eligible(int d, int K) {
  return assign_split_eligible(d, K) && half_shape_ok(d) &&
         half_lds_bytes<8>(d) + (size_t)kHalfLdsList * 2 + 16 <= 160 * 1024 &&
         split_lds_bytes<8>(d) + (size_t)kSplitLdsList * 6 + 8 * 64 * 4 <= 160 * 1024;
}

// x: fp32 rows, xm / xt: their fp16 copy.  q1 [B][q1cap] / q1count [B]: per-image queues
// of the rows the first level could not decide; qrows / qcount: exact queue.
int launch_assign_half(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent, int K, int B,
                       const ChunkTable &t, int max_chunks, int32_t *klab, int32_t *q1,
                       int32_t *q1count, int64_t q1cap, void *qrows, int32_t *qcount,
                       const hsgk_segkm_meta *meta, hipStream_t s, bool counters_zeroed, const _Float16 *xmT) {
  if (max_chunks <= 0 || B <= 0) return 0;
  constexpr int NW = 8, TPX = NW * 32;
  static const int n_cu = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess)
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
  }();
  // one persistent workgroup per CU, fewer when the batch has fewer tiles
  const int64_t max_tiles = ((int64_t)max_chunks * HSGK_CHUNK + TPX - 1) / TPX;
  const int grid = (int)(max_tiles < n_cu ? max_tiles : n_cu);
  if (!counters_zeroed) {
    HSGK_CHECK_HIP(hipMemsetAsync(q1count, 0, sizeof(int32_t) * B, s));
    HSGK_CHECK_HIP(hipMemsetAsync(qcount, 0, sizeof(int32_t), s));
  }
  // Small batches are bound by their ~5 dependent launches per iteration, not by bytes: below
  // 1.5 M rows the undecided rows of the fp16 level go straight to the exact chains with their
  // candidate sets (one launch less: -10 % at training resolutions, -5 % at 16 x 224 x 224;
  // at 48 x 448 x 448 the bf16x3 level saves 3.6 ms per call).  HSGK_L2 = 0 / 1 forces either
  // way (read per call, so that the tests can cover both).
  const char *l2env = getenv("HSGK_L2");
  const bool direct = l2env ? l2env[0] == '0' : (int64_t)max_chunks * HSGK_CHUNK <= 1500000;
  if (direct && ((d / 64) & 3) == 0) {
    auto kern = assign_half_wide_kernel<NW, 4, 2, 2, 2>;
    const size_t lds = half_lds_bytes<NW, 2, 2, 2>(d) + (size_t)kSplitLdsList * 6 + 16;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, xm, xt, d, cent, (const float *)nullptr, K,
                       t.img_row0, B, klab, reinterpret_cast<SplitEntry *>(qrows), qcount, meta);
    HSGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                       reinterpret_cast<const SplitEntry *>(qrows), qcount, t.img_row0, B);
    HSGK_LAUNCH_CHECK();
    return 0;
  }
  if (xmT && (d / 64 == 2 || d / 64 == 4)) {
    auto kern = d / 64 == 4 ? assign_half_t_kernel<NW, 4> : assign_half_t_kernel<NW, 2>;
    const size_t lds = half_t_lds_bytes<2, 2>(d) + (size_t)kHalfLdsList * 2 + 16;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, xmT, xt, d, cent, K, t.img_row0, B,
                       klab, q1, q1count, q1cap, meta);
    HSGK_LAUNCH_CHECK();
  } else {
    const bool deep = ((d / 64) & 3) == 0;
    // rows of exactly four 64-column chunks (C = 256): six sets = one and a half tiles in flight per wave
    // (HSGK_HALF_DEPTH=4 keeps four, read per call)
    const char *de = getenv("HSGK_HALF_DEPTH");
    const bool six = d / 64 == 4 && !(de && de[0] == '4');
    auto kern = six ? assign_half_kernel<NW, 6> : deep ? assign_half_kernel<NW, 4> : assign_half_kernel<NW, 2>;
    const size_t lds = half_lds_bytes<NW>(d) + (size_t)kHalfLdsList * 2 + 16;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, xm, xt, d, cent, K, t.img_row0, B,
                       klab, q1, q1count, q1cap, meta);
    HSGK_LAUNCH_CHECK();
  }
  {
    // about one workgroup per CU in total (the table is staged once per workgroup and the
    // queues are short), in slices of <= 65535 rows (u16 queue offsets)
    int T = n_cu / B > 1 ? n_cu / B : 1;
    while ((int64_t)T * 32768 < q1cap) T *= 2;
    auto kern = assign_split_rows_kernel<NW>;
    const size_t lds = split_lds_bytes<NW>(d) + (size_t)kSplitLdsList * 6 + (size_t)NW * 64 * 4;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(T, B), dim3(NW * 64), lds, s, x, d, cent, K, q1, q1count, q1cap,
                       klab, reinterpret_cast<SplitEntry *>(qrows), qcount);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                     reinterpret_cast<const SplitEntry *>(qrows), qcount, t.img_row0, B);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// Small feature maps (the reference's TRAINING resolution: input / 16, e.g. 48 x 256 x 28 x 28): the
// whole Lloyd loop of one image in ONE workgroup, one launch per call.
// With the per-kernel route a call is ~5 dependent launches per iteration whose fixed costs (a
// persistent kernel's start-up, its table staging, its drain) add up to ~50 us per iteration for
// 37 K rows.  Here workgroup b owns image b for all iterations and the phases of an iteration are
// separated by workgroup barriers only:
//   M  exact fixed-point sums (order C2x, sums_fx.hip): the rows whose label changed are added to
//      / subtracted from an int64 [K][d] LDS table with ds_add_u64, the table is folded into the
//      image's running sums in global memory (L2), and the same threads leave the unnormalised
//      fp32 centroid in LDS (compact, above the engine's table planes: SmallLayout);
//   F  one lane per centroid walks the canonical norm chain, all threads divide -> cent (global)
//      and, from the same registers, the engine's fp16 hi / lo planes (no staging round trip);
//   E  the fp16 filter engine (score_tiles_f16.h, hi + lo table planes) over the image's rows,
//      ambiguous rows with their candidate sets into an LDS / global list;
//   X  the exact fp32 chains of those candidates (exact_rescore16).
// Same arithmetic, same labels as the per-kernel route (tests run both: HSGK_SMALL=0 / 1).
// One CU per image bounds it: 0.24 ms instead of 0.58 for 16 x 256 x 14 x 14, 0.43 instead of
// 0.66 for 48 x 256 x 28 x 28, slower than the per-kernel route from ~1500 rows per image on.
// (Tried and dropped: the table as MFMA A operands in the registers of four 512-register waves,
// rows straight from the fp16 copy as B operands, sums persistent in LDS -- every phase turned
// latency-bound with one wave per SIMD, 0.95 ms for the 28 x 28 batch.)
constexpr int kSmallRowsMax = 1024;       // rows per image the fused kernel accepts
#ifdef HSGK_SMALL_TIMING                   // tools/probes/small_timing.py: cycles per phase, workgroup 0
#define HSGK_STS(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
    atomicAdd(&g_small_ts[i], now_ - sts_); sts_ = now_; } } while (0)
#else
#define HSGK_STS(i) do { } while (0)
#endif

// LDS map of the fused kernel (bytes; the whole 160 KiB of the CU, one workgroup per CU):
//   [0, 8 K d)            int64 sums table                                   (M, flush)
//   [8 K d + 16, ...)     per-wave changed-row lists                         (M)
//   [fc0, fc0 + 4 K d)    unnormalised fp32 centroids, compact [K][d]        (flush -> F -> divide)
//   [planes, fc0)         K norms right after the planes                     (F -> divide)
//   [0, planes)           fp16 hi / lo table planes, then the row windows    (divide -> E)
//   [tail0, 160 KiB)      LDS list of ambiguous rows                         (E -> X)
// The fp32 centroids sit ABOVE the planes, so the divide step writes the planes straight from
// them (no global round trip, nothing held in registers across a barrier).  Writing them while
// the int64 table is still being read needs an order: the flush handles the table pairs whose
// fp32 slot lies past the table first, then -- after a barrier -- the rest, whose fp32 slots
// overwrite only pairs of the first batch (3 fc0 >= 16 K d, checked by small_layout_ok).
struct SmallLayout { int planes, nrm0, fc0, tail0, total, p1; };
__host__ __device__ inline SmallLayout small_layout(int d, int K) {
  SmallLayout L;
  L.total = 160 * 1024;
  L.tail0 = (L.total - (kSplitLdsList * 6 + 16)) & ~15;
  L.fc0 = (L.tail0 - 4 * K * d) & ~15;
  L.planes = 2 * 64 * (half_main_cols(d) + 16 + 8) * 2;
  L.nrm0 = L.planes;
  const int over = 8 * K * d - L.fc0;                  // bytes of the table the fp32 array overlaps
  L.p1 = over > 0 ? (over + 7) / 8 : 0;                // first table pair whose fp32 slot is past the table
  return L;
}
__host__ __device__ inline bool small_layout_ok(int d, int K, int NW) {
  const SmallLayout L = small_layout(d, K);
  const int pairs = K * d / 2;
  return ((K * d) & 1) == 0 && L.fc0 >= L.nrm0 + K * 4 && L.fc0 / 16 >= L.p1 && L.p1 <= pairs &&
         (int)half_lds_bytes<8, 2, 2, 2>(d) <= L.tail0 && 8 * K * d + 16 + NW * 64 * 4 <= L.tail0;
}

template <int NW, int DEPTH, int NV, int MB>
__global__ __launch_bounds__(NW * 64) void lloyd_small_kernel(
    const float *__restrict__ x, const _Float16 *__restrict__ xm, const uint2 *__restrict__ xt, int d, int K,
    int iterations, const int64_t *__restrict__ img_row0, int32_t *__restrict__ lab_a,
    int32_t *__restrict__ lab_b, long long *__restrict__ sumq, float *__restrict__ cent,
    SplitEntry *__restrict__ gqueue, int32_t *__restrict__ gcount, int first_sums_ready, float eps,
    int G, unsigned int *__restrict__ bar, hsgk_segkm_meta *meta, float *__restrict__ cent_multi) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // G workgroups per image (co-resident: the launcher keeps B * G within the CU count): workgroup g
  // owns a contiguous, 32-aligned share of the image's rows for the M, E and X phases; the
  // image's running sums are shared through device-scope atomics and a per-image tick counter.
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int nimg = (int)(img_row0[b + 1] - img_row0[b]);
  const int rpw = (((nimg + G - 1) / G) + 31) & ~31;
  const int rs = min(nimg, g * rpw);
  const int64_t r0 = img_row0[b] + rs;             // first row of THIS workgroup
  const int n = min(nimg, rs + rpw) - rs;          // its rows (0 for a trailing workgroup: barriers only)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const SmallLayout L = small_layout(d, K);
  // phase M / F view of the LDS
  unsigned long long *tab = reinterpret_cast<unsigned long long *>(lds_raw);            // [K][d]
  uint32_t *lists = reinterpret_cast<uint32_t *>(tab + (size_t)K * d + 2);              // [NW][64]
  float *fc = reinterpret_cast<float *>(lds_raw + L.fc0);                               // [K][d] fp32
  float *nrm = reinterpret_cast<float *>(lds_raw + L.nrm0);                             // [K]
  // phase E / X view
  unsigned char *etail = lds_raw + L.tail0;
  int *qnp = reinterpret_cast<int *>(etail);                                            // [0] LDS-list length
  uint32_t *qcand = reinterpret_cast<uint32_t *>(etail + 16);                           // [kSplitLdsList]
  uint16_t *qpx = reinterpret_cast<uint16_t *>(qcand + kSplitLdsList);
  long long *sq = sumq + (int64_t)b * K * d;
  // centroids in global memory (the exact chains read them from there).  With several workgroups
  // per image every workgroup keeps ITS OWN copy: the same address written by workgroups on
  // different XCDs sits dirty in several non-coherent L2s, and a late write-back of an OLDER
  // iteration's line from one of them can land after a newer line was evicted elsewhere -- a
  // re-fetch then returns stale centroids (seen once in ~600 calls with a second stream competing
  // for the caches, never alone).
  const int cslot = G > 1 ? (int)blockIdx.x : b;
  const float *cbase = G > 1 ? cent_multi : cent;
  float *ct = const_cast<float *>(cbase) + (int64_t)cslot * K * d;
  SplitEntry *gq = gqueue + r0;                   // this workgroup's overflow region (<= n entries)
  int32_t *gc = gcount + blockIdx.x;
  int32_t *cur = lab_a, *prev = lab_b;
  typedef float gvec_t __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const int nq = d / 4, tail0 = nq * 4, tw = d - tail0;
  constexpr int UNROLL = 8;
  const int tu = lane / max(tw, 1), tc = lane - tu * max(tw, 1);
  const bool tact = tw > 0 && tu < UNROLL;
  uint32_t *list = lists + w * 64;
#ifdef HSGK_SMALL_TIMING
  unsigned long long sts_ = __builtin_readcyclecounter();
#endif

  auto zero_table = [&]() {
    const int tot2 = (K * d + 1) / 2;
    u64x2 *t2 = reinterpret_cast<u64x2 *>(tab);
    for (int i = tid; i < tot2; i += NW * 64) t2[i] = u64x2{0ull, 0ull};
  };
  if (tid == 0) __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (first append: after barriers)
  // tick counter of the image: every workgroup ticks twice per iteration (sums added, sums read)
  int &bar_dead = qnp[2];                        // a wait timed out: stop waiting (the call reports error 3);
  if (tid == 0) bar_dead = 0;                    // (in the list header: no static LDS beside the 160 KiB array)
  // The launcher admits the grid only if it fits half of the device and keeps at most two such grids in
  // flight per device (small_admit), so the other shares of the image are running or will start as soon as
  // kernels of OTHER streams release their CUs: the wait is bounded by wall-clock time (10 s of the 100 MHz
  // real-time counter, far beyond any kernel a training step runs beside this one), not by a spin count a
  // profiler or a busy neighbour could exhaust.  On a time-out the call reports error 3 in its meta block
  // (the labels are then not valid); the mirror re-runs such a call with one workgroup per image or raises
  // it through the deferred-error path -- the kernel never traps.
  auto wait_ticks = [&](unsigned int target) {
    if (tid == 0 && !bar_dead) {
      unsigned long long t0 = 0;
      unsigned int spins = 0;
      while (__hip_atomic_load(bar + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);                // (polling without it is not faster)
        if ((++spins & 1023u) == 0) {
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          if (now - t0 > 1000000000ull) {
            if (meta) meta->error = 3;
            bar_dead = 1;
            break;
          }
        }
      }
    }
    __syncthreads();
  };
  for (int it = 0; it < iterations && nimg > 0; ++it) {
    // ---------------------------------------------------------------- M: exact sums of the changed rows
    const bool skip_m = it == 0 && first_sums_ready;
    // first iteration without prepared sums: no row has been added yet (previous label -1) and the
    // running sums are zero -- neither buffer is read, so the caller does not have to initialise them
    const bool fresh = it == 0 && !first_sums_ready;
    zero_table();
    __syncthreads();
    if (!skip_m) {
      for (int st = w; st * 64 < n; st += NW) {
        const int rb = st * 64, nr = min(64, n - rb);
        const int rr = min(lane, nr - 1);
        int pl = fresh ? -1 : prev[r0 + rb + rr];
        const int cl = cur[r0 + rb + rr];
        if (lane >= nr) pl = cl;
        const bool ch = pl != cl;
        const unsigned long long m = __ballot(ch);
        const int total = __popcll(m);
        if (total == 0) continue;
        if (ch)   // row (6 bits) | new label + 1 (11 bits) | old label + 1 (11 bits, 0: not added yet)
          list[__popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)lane << 22) | ((uint32_t)(cl + 1) << 11) | (uint32_t)(pl + 1);
        // (wave-private list: own LDS writes are visible to own reads in order)
        const float *xr = x + (r0 + rb) * d;
        auto entry = [&](int i) { return list[min(i, total - 1)]; };
        auto issue = [&](int i0, gvec_t (&v)[UNROLL][NV], float &t) {
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) {
            const float *src = xr + (int64_t)(entry(i0 + u) >> 22) * d;
#pragma unroll
            for (int h = 0; h < NV; ++h)
              v[u][h] = *reinterpret_cast<const gvec_t *>(src + 4 * min(lane + 64 * h, nq - 1));
          }
          t = xr[(int64_t)(entry(i0 + min(tu, UNROLL - 1)) >> 22) * d + min(tail0 + tc, d - 1)];
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
                for (int jj = 0; jj < 4; ++jj) q[h][jj] = to_fixed(v[u][h][jj]);
#pragma unroll
              for (int side = 0; side < 2; ++side) {
                if (labs[side] < 0) continue;                  // (wave-uniform)
                unsigned long long *rowp = tab + (size_t)labs[side] * d;
#pragma unroll
                for (int h = 0; h < NV; ++h) {
                  const int q4 = lane + 64 * h;
                  if (q4 < nq) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                      atomicAdd(rowp + 4 * q4 + jj, (unsigned long long)(side ? -q[h][jj] : q[h][jj]));
                  }
                }
              }
            }
          }
          if (tact && i0 + tu < total) {
            const uint32_t e = entry(i0 + tu);
            const long long qt = to_fixed(t);
            const int ln = (int)((e >> 11) & 2047u) - 1, lo = (int)(e & 2047u) - 1;
            if (ln >= 0) atomicAdd(tab + (size_t)ln * d + tail0 + tc, (unsigned long long)qt);
            if (lo >= 0) atomicAdd(tab + (size_t)lo * d + tail0 + tc, (unsigned long long)(-qt));
          }
        };
        gvec_t va[UNROLL][NV], vb[UNROLL][NV];
        float ta, tb;
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
      }
    }
    __syncthreads();
    HSGK_STS(0);
    { int32_t *t = cur; cur = prev; prev = t; }      // the sums now hold `prev`; the E-step writes `cur`
    if (G > 1) {
      // several workgroups per image: (a) nobody still reads the sums of the previous iteration,
      // (b) this workgroup's deltas go to the image's running sums with device-scope atomics
      // (performed at the memory side: the workgroups sit on different XCDs / L2s), (c) tick, wait
      // for the others, (d) read the sums back with device-scope loads -> fp32 array, tick.
      constexpr float kInvScale = 9.094947017729282e-13f;   // 2^-40, one rounding (finalize_fx_kernel)
      const int total = K * d;
      wait_ticks(2u * (unsigned)G * (unsigned)it);
      HSGK_STS(8);
      for (int i = tid; i < total; i += NW * 64) {
        const unsigned long long tv = tab[i];
        if (tv) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(sq) + i, tv, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_waitcnt(0);               // the atomics above have been performed
      __syncthreads();
      HSGK_STS(9);
      if (tid == 0) (void)__hip_atomic_fetch_add(bar + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ticks(2u * (unsigned)G * (unsigned)it + (unsigned)G);
      HSGK_STS(10);
      for (int i0 = tid; i0 < total; i0 += 16 * NW * 64) {      // (memory-side loads: 16 in flight per thread)
        long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          v[u] = __hip_atomic_load(sq + min(i0 + u * NW * 64, total - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int i = i0 + u * NW * 64;
          if (i < total) fc[i] = (float)v[u] * kInvScale;
        }
      }
      __syncthreads();
      HSGK_STS(11);
      if (tid == 0) (void)__hip_atomic_fetch_add(bar + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else
    // running sums += table -> global (L2) and, as one fp32 rounding of the sum, the compact
    // array above the planes.  Two slots per 16-byte access, twelve independent L2 loads per
    // thread in flight; the pairs whose fp32 slot lies past the table go first (see SmallLayout).
    {
      typedef long long i64x2 __attribute__((ext_vector_type(2), aligned(8)));
      constexpr int FU = 12;
      constexpr float kInvScale = 9.094947017729282e-13f;   // 2^-40, one rounding (finalize_fx_kernel)
      const int pairs = (K * d) >> 1;
      auto batch = [&](int pbeg, int pend) {
        for (int p0 = pbeg + tid; p0 < pend; p0 += FU * NW * 64) {
          i64x2 g[FU];
#pragma unroll
          for (int u = 0; u < FU; ++u)
            g[u] = fresh ? i64x2{0, 0} : *reinterpret_cast<const i64x2 *>(sq + 2 * (size_t)min(p0 + u * NW * 64, pend - 1));
#pragma unroll
          for (int u = 0; u < FU; ++u) {
            const int p = p0 + u * NW * 64;
            if (p < pend) {
              const u64x2 tv = *reinterpret_cast<const u64x2 *>(tab + 2 * (size_t)p);
              const i64x2 v = {g[u].x + (long long)tv.x, g[u].y + (long long)tv.y};
              if ((tv.x | tv.y) || fresh) *reinterpret_cast<i64x2 *>(sq + 2 * (size_t)p) = v;
              *reinterpret_cast<float2 *>(fc + 2 * (size_t)p) = float2{(float)v.x * kInvScale, (float)v.y * kInvScale};
            }
          }
        }
      };
      // (requesting the sums of both batches up front -- one L2 round trip instead of two -- costs
      // 96 registers at this point of a kernel that has ~50 to spare: 150 spilled, slower)
      batch(L.p1, pairs);
      if (L.p1 > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // (LDS order only)
        batch(0, L.p1);
      }
    }
    __syncthreads();
    HSGK_STS(1);
    // ---------------------------------------------------------------- F: norm chain, divide
    // one lane per centroid walks the canonical chain; its LDS reads are issued 16 at a time
    if (tid < K) {
      const float *r = fc + (size_t)tid * d;
      float ss = 0.0f;
      int i = 0;
      for (; i + 16 <= d; i += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = r[i + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) ss = fmaf(v[u], v[u], ss);
      }
      for (; i < d; ++i) ss = fmaf(r[i], r[i], ss);
      float nv = sqrtf(ss);
      if (!(nv >= eps)) nv = eps;
      nrm[tid] = nv;
    }
    if (tid == 0) qnp[0] = 0;
    __syncthreads();
    // divide: wave w owns the table rows w + NW*u, a lane the column pairs lane + 64*i (d is even:
    // half_shape_ok) -> cent (global: the exact chains, the caller) and the engine's fp16 hi / lo
    // planes (zero padded to 64 rows and to the plane row length), which no longer overlap `fc`
    {
      const int dp = d >> 1, RS2 = (half_main_cols(d) + 16 + 8) >> 1;
      uint32_t *ch32 = reinterpret_cast<uint32_t *>(lds_raw);
      uint32_t *cl32 = ch32 + 32 * MB * RS2;                  // (MB = 1: a 32-row table for K <= 32, half the MFMAs)
      for (int k = w; k < 32 * MB; k += NW) {
        const bool live = k < K;
        const float nv = nrm[min(k, K - 1)];
#pragma unroll
        for (int i = 0; i < 3; ++i) {                           // 3 x 64 pairs >= RS2 (d <= 322)
          const int pr = lane + 64 * i;
          if (pr < RS2) {
            uint32_t hi = 0u, lo = 0u;
            if (live && pr < dp) {
              const float2 f = *reinterpret_cast<const float2 *>(fc + (size_t)k * d + 2 * pr);
              const float2 c = float2{f.x / nv, f.y / nv};
              *reinterpret_cast<float2 *>(ct + (size_t)k * d + 2 * pr) = c;
              f16_split2(c.x, c.y, hi, lo);
            }
            ch32[k * RS2 + pr] = hi;
            cl32[k * RS2 + pr] = lo;
          }
        }
      }
    }
    HSGK_STS(2);
    // ---------------------------------------------------------------- E: fp16 filter over the image's rows
    {
      HalfWideEpi<MB, true> epi{K, n, b, r0, 0.0f, cur, qpx, qcand, qnp, gq, gc};
      score_tiles_half<NW, DEPTH, HalfWideEpi<MB, true>, MB, 2, 2>(xm, xt, d, ct, K, r0, n, lds_raw, epi, false);
    }
    __syncthreads();
    HSGK_STS(3);
    // ---------------------------------------------------------------- X: exact chains of the ambiguous rows
    {
      const int qn = min(qnp[0], kSplitLdsList);
      const int gn = (int)__hip_atomic_load(gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int total = qn + gn;
      const int grp = lane >> 2;
      for (int e0 = w * 16; e0 < total; e0 += NW * 16) {
        const int e = e0 + grp;
        SplitEntry ent{0, 0u, 0u};
        if (e < qn) ent = SplitEntry{(int32_t)(r0 + qpx[e]), qcand[e], 0u};
        else if (e < total) ent = gq[e - qn];
        exact_rescore16(ent, e < total, cslot, x, d, cbase, K, cur);
      }
      __syncthreads();
      if (tid == 0 && gn) __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    HSGK_STS(4);
  }
  // the caller expects the final labels in lab_a
  if (cur != lab_a)
    for (int i = tid; i < n; i += NW * 64) lab_a[r0 + i] = cur[r0 + i];
}

#ifdef HSGK_SMALL_TIMING
extern "C" __attribute__((visibility("default"))) int hsgk_debug_small_timing(unsigned long long *out) {
  unsigned long long h[12], z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_small_ts), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 12; ++i) out[i] = h[i];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_small_ts), z, sizeof(z));
  return 0;
}
#endif

// workgroups per image: one up to 512 rows; beyond that one per 256 rows (one engine tile each, at
// most kSmallGroupsMax), as many as keep all B * G workgroups co-resident (one workgroup per CU: they wait for each other).
// Half of the CUs at most, so that two such calls on two streams still fit side by side (two
// half-resident grids would wait for each other's CUs; a wait that times out reports error 3).
constexpr int kSmallGroupsMax = 16;
static int small_cu_count() {              // of the CURRENT device (cached per device: one thread per GPU calls in)
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); return 0; }
  if (dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
  int cu = 0;
  if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
  (void)hipGetLastError();
  if (dev < 64 && cu > 0) cache[dev].store(cu, std::memory_order_relaxed);
  return cu;
}
// The multi-workgroup grids are PLAIN launches (HSGK_SMALL_COOP=1 switches to hipLaunchCooperativeKernel: ~30 us per
// call on this runtime -- reftrain 0.294 -> 0.328 ms, train28 0.382 -> 0.412 ms, tools/probes/coop_ab.sh -- and the
// same residency: the co-operative API only adds the launch-time size check that small_resident_capacity repeats
// here).  What makes the inter-workgroup waits safe is that
//   (1) a grid is admitted only within HALF of what the device holds of this kernel, and
//   (2) at most TWO such grids of this process are in flight per device, whatever the number of streams
//       (small_admit below: a third launch first waits, on the device, for the one two launches back),
// so every admitted grid finds its CUs as soon as ordinary neighbour kernels (which wait for nobody) drain; two
// half-resident grids holding CUs while each waits for its own missing workgroups cannot occur.  The wall-clock
// bounded wait (error 3) remains for what the process cannot see: another PROCESS running such grids on this GPU.
static bool small_coop_enabled() {
  const char *e = getenv("HSGK_SMALL_COOP");
  return e && e[0] == '1';
}
// In-flight bound of the multi-workgroup launches (per device): a ring of two events.  Before launch n the
// stream waits for the event of launch n - 2; after it the stream records launch n's event.  Capturing streams
// skip the ring (an event recorded outside a capture cannot be waited for inside one): a captured call is
// ordered by its graph and callers that replay such graphs on more than two streams at once should pass
// HSGK_SEGKM_ONE_GROUP.
struct SmallRing { std::mutex mu; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; unsigned n = 0; };
static SmallRing &small_ring(int dev) { static SmallRing rings[64]; return rings[dev & 63]; }
static bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
template <typename F>
static int small_admit(hipStream_t s, F &&launch) {
  int dev = 0;
  if (stream_capturing(s) || hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return launch(); }
  SmallRing &r = small_ring(dev);
  std::lock_guard<std::mutex> g(r.mu);
  const unsigned slot = r.n++ & 1u;
  if (!r.ev[slot]) HSGK_CHECK_HIP(hipEventCreateWithFlags(&r.ev[slot], hipEventDisableTiming));
  if (r.used[slot]) HSGK_CHECK_HIP(hipStreamWaitEvent(s, r.ev[slot], 0));
  if (int rc = launch()) return rc;
  HSGK_CHECK_HIP(hipEventRecord(r.ev[slot], s));
  r.used[slot] = true;
  return 0;
}
// workgroups of `kern` (512 threads, `lds` bytes) the current device holds at once
static int small_resident_capacity(const void *kern, size_t lds) {
  struct Slot { std::atomic<const void *> k{nullptr}; std::atomic<int> cap{0}; };
  static Slot slots[64][8];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 0; }
  for (auto &sl : slots[dev])
    if (sl.k.load(std::memory_order_acquire) == kern) return sl.cap.load(std::memory_order_relaxed);
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 8 * 64, lds) != hipSuccess) nb = 0;
  (void)hipGetLastError();
  const int cap = nb * small_cu_count();
  for (auto &sl : slots[dev]) {
    const void *expect = nullptr;
    if (sl.k.compare_exchange_strong(expect, kern)) { sl.cap.store(cap, std::memory_order_relaxed); break; }
  }
  return cap;
}
int lloyd_small_rows_max() { return kSmallRowsMax; }
int lloyd_small_groups(int B, int64_t rows_per_image) {
  const char *fe = getenv("HSGK_SMALL_GROUPS");        // tests: force the number of workgroups per image (read per call)
  const int forced = fe ? atoi(fe) : 0;
  const int gmax = std::max(1, std::min(kSmallGroupsMax, B > 0 ? small_cu_count() / 2 / B : 1));
  if (forced > 0) return std::min(forced, gmax);
  if (rows_per_image <= 512) return 1;
  return (int)std::min<int64_t>(gmax, (rows_per_image + 255) / 256);
}

// floats of scratch for the per-workgroup centroid copies of the multi-workgroup route (B * G <= CUs / 2)
size_t lloyd_small_cent_floats(int d, int K, int B) {
  return ((size_t)small_cu_count() / 2 + (size_t)B) * K * d;
}

bool lloyd_small_eligible(int d, int K, int B, int64_t rows_per_image) {
  return assign_half_eligible(d, K) && small_layout_ok(d, K, 8) &&
         rows_per_image <= (int64_t)kSmallRowsMax * lloyd_small_groups(B, rows_per_image);
}

// lab_a: current labels (in / out), lab_b: the labels the sums hold; sumq / cent: [B][K][d];
// qrows: >= one SplitEntry per row; counters: >= B * (G + 1) int32 (per-workgroup queue counts, then
// the per-image tick counters).  single_group: one workgroup per image whatever the map size (no
// inter-workgroup wait at all).  first_sums_ready: sumq / lab_b are valid (first M-step done
// elsewhere); otherwise neither needs initialising (one workgroup per image) or sumq is zeroed here.
int launch_lloyd_small(const float *x, const _Float16 *xm, const uint2 *xt, int d, int K, int B, int iterations,
                       const ChunkTable &t, int32_t *lab_a, int32_t *lab_b, long long *sumq, float *cent,
                       void *qrows, int32_t *counters, bool first_sums_ready, hsgk_segkm_meta *meta,
                       int64_t rows_per_image, bool single_group, float *cent_multi, hipStream_t s) {
  if (B <= 0 || iterations <= 0) return 0;
  constexpr int NW = 8;
  const bool deep = ((d / 64) & 3) == 0;
  const size_t lds = (size_t)small_layout(d, K).total;
  int G = (cent_multi && !single_group) ? lloyd_small_groups(B, rows_per_image) : 1;   // (no scratch for private centroids: one workgroup per image)
  // Under stream capture the two-in-flight admission below cannot be applied (a graph may replay on any number of
  // streams at once): one workgroup per image where the map allows it, else the co-operative launch, whose grid the
  // runtime itself checks for co-residency.
  const bool capturing = G > 1 && stream_capturing(s);
  if (capturing && rows_per_image <= kSmallRowsMax) G = 1;
  auto go = [&](auto kern) -> int {
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // co-operating workgroups must all be resident: admitted only within HALF of what the device holds of
    // this kernel (a second such call on another stream still fits beside it); otherwise one workgroup per
    // image, which waits for nobody (maps beyond its row limit are refused: lloyd_small_groups sizes G from
    // the same CU count, so that only happens if the kernel's occupancy is not what it was built for).
    if (G > 1 && 2 * B * G > small_resident_capacity(reinterpret_cast<const void *>(kern), lds)) {
      HSGK_REQUIRE(rows_per_image <= kSmallRowsMax, "co-operating workgroups would not be co-resident");
      G = 1;
    }
    unsigned int *bar = reinterpret_cast<unsigned int *>(counters + (size_t)B * G);
    if (G > 1) {
      HSGK_CHECK_HIP(hipMemsetAsync(bar, 0, sizeof(unsigned int) * B, s));
      if (!first_sums_ready) HSGK_CHECK_HIP(hipMemsetAsync(sumq, 0, sizeof(long long) * (size_t)B * K * d, s));
    }
    auto plain = [&]() -> int {
      hipLaunchKernelGGL(kern, dim3(B * G), dim3(NW * 64), lds, s, x, xm, xt, d, K, iterations, t.img_row0, lab_a,
                         lab_b, sumq, cent, reinterpret_cast<SplitEntry *>(qrows), counters,
                         first_sums_ready ? 1 : 0, HSGK_EPS, G, bar, meta, cent_multi);
      HSGK_LAUNCH_CHECK();
      return 0;
    };
    if (G <= 1) return plain();                // one workgroup per image waits for nobody
    if (small_coop_enabled() || capturing) {
      // (the runtime rejects a grid that cannot be resident at once -- hipErrorCooperativeLaunchTooLarge)
      return small_admit(s, [&]() -> int {
        const int64_t *img_row0 = t.img_row0;
        SplitEntry *gq = reinterpret_cast<SplitEntry *>(qrows);
        int fsr = first_sums_ready ? 1 : 0, g = G;
        float eps = HSGK_EPS;
        void *kargs[] = {(void *)&x, (void *)&xm, (void *)&xt, (void *)&d, (void *)&K, (void *)&iterations,
                         (void *)&img_row0, (void *)&lab_a, (void *)&lab_b, (void *)&sumq, (void *)&cent, (void *)&gq,
                         (void *)&counters, (void *)&fsr, (void *)&eps, (void *)&g, (void *)&bar, (void *)&meta,
                         (void *)&cent_multi};
        HSGK_CHECK_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(kern), dim3(B * G), dim3(NW * 64),
                                                  kargs, (unsigned int)lds, s));
        return 0;
      });
    }
    return small_admit(s, plain);
  };
  if (K <= 32) {
    if (d <= 259) return deep ? go(lloyd_small_kernel<NW, 4, 1, 1>) : go(lloyd_small_kernel<NW, 2, 1, 1>);
    return deep ? go(lloyd_small_kernel<NW, 4, 2, 1>) : go(lloyd_small_kernel<NW, 2, 2, 1>);
  }
  if (d <= 259) return deep ? go(lloyd_small_kernel<NW, 4, 1, 2>) : go(lloyd_small_kernel<NW, 2, 1, 2>);
  return deep ? go(lloyd_small_kernel<NW, 4, 2, 2>) : go(lloyd_small_kernel<NW, 2, 2, 2>);
}

int launch_assign_fast(const float *x, int d, const float *cent, int K, int B, const ChunkTable &t,
                       int max_chunks, int32_t *klab, float *best, void *qrows,
                       int32_t *qcount, const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  static const int mode = [] {
    const char *e = getenv("HSGK_ASSIGN");          // "fp32" forces the exact kernel only
    return (e && e[0] == 'f') ? 0 : 1;
  }();
  if (mode == 1 && qrows && assign_split_eligible(d, K))
    return launch_assign_split(x, d, cent, K, B, t, max_chunks, klab,
                               reinterpret_cast<SplitEntry *>(qrows), qcount, meta, s);
  return launch_assign(x, d, cent, K, t, max_chunks, klab, best, meta, s);
}

template <int KB, int NW, int KC, bool EVEN_D>
static int launch_assign_cfg2(const float *x, int d, const float *cent, int K,
                             const ChunkTable &t, int max_chunks, int32_t *klab,
                             float *best, const hsgk_segkm_meta *meta, hipStream_t s) {
  constexpr int TPX = NW * 32;
  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  size_t lds = ((size_t)KB * DP + (size_t)2 * TPX * (KC + 1)) * 4;
  auto kern = assign_kernel<KB, NW, KC, EVEN_D>;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  constexpr int kTiles = HSGK_CHUNK / TPX;
  int split = 1;
  while (split < kTiles && (int64_t)max_chunks * split < 2048) split *= 2;
  const int grid = max_chunks * split;
  const bool multi = K > KB;
  for (int kb0 = 0; kb0 < K; kb0 += KB) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, x, d, cent, K, kb0,
                       t.chunk_row0, t.chunk_rows, t.chunk_img, klab,
                       multi ? best : nullptr, kb0 == 0 ? 1 : 0, split, meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

template <int KB, int NW, int KC>
static int launch_assign_cfg(const float *x, int d, const float *cent, int K,
                             const ChunkTable &t, int max_chunks, int32_t *klab,
                             float *best, const hsgk_segkm_meta *meta, hipStream_t s) {
  if ((d & 1) == 0)
    return launch_assign_cfg2<KB, NW, KC, true>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
  return launch_assign_cfg2<KB, NW, KC, false>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
}

int launch_assign(const float *x, int d, const float *cent, int K, const ChunkTable &t,
                  int max_chunks, int32_t *klab, float *best,
                  const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  const size_t cap = 160 * 1024;
  auto fits = [&](int KB, int NW, int KC) {
    return ((size_t)KB * DP + (size_t)2 * NW * 32 * (KC + 1)) * 4 <= cap;
  };
  if (K <= 32) {
    if (fits(32, 8, 32)) return launch_assign_cfg<32, 8, 32>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(32, 8, 16)) return launch_assign_cfg<32, 8, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(32, 4, 16)) return launch_assign_cfg<32, 4, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
  } else {
    if (fits(64, 8, 32)) return launch_assign_cfg<64, 8, 32>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(64, 8, 16)) return launch_assign_cfg<64, 8, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(64, 4, 16)) return launch_assign_cfg<64, 4, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(32, 8, 16)) return launch_assign_cfg<32, 8, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
    if (fits(32, 4, 16)) return launch_assign_cfg<32, 4, 16>(x, d, cent, K, t, max_chunks, klab, best, meta, s);
  }
  set_error("assign: row length %d does not fit the LDS centroid block", d);
  return -1;
}

}  // namespace hsgk

#ifdef HSGK_Q_STATS
extern "C" __attribute__((visibility("default"))) int hsgk_debug_qstats(unsigned long long *out) {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(hsgk::g_qstats), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out[i] = h[i];
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hsgk::g_qstats), z, sizeof(z));
  return 0;
}
#endif
