// kmeans.hip -- Lloyd iteration of the spherical k-means (reference
// hsg/utils/segsort/common.py:67-97): M-step = calculate_prototypes_from_labels
// (:11-41), E-step = find_nearest_prototypes (:44-64).
//
// Canonical arithmetic:
//   M  per chunk (<= 2048 consecutive rows of one image) every (cluster,
//      column) is summed sequentially in row order from +0.0f; chunk partials
//      are summed sequentially in chunk order from +0.0f; the centroid is the
//      sum divided by its norm (fmaf chain over columns, sqrtf, '/').
//   E  <x, c_k> is one fmaf chain over ascending column index from +0.0f --
//      exactly what v_mfma_f32_32x32x2_f32 computes (k-ordered fmaf chain) --
//      argmax takes the first maximal index.
#include "common.h"
#include "accumulate.h"
#include "score_tiles.h"
#include "score_tiles_bf16.h"

namespace hsgk {


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
template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void accumulate_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ klab,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    int K, int kb0, int kbn, float *__restrict__ partial,
    const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ float sums[];   // [kbn][DS] then the row list
  __shared__ int wcount[4];
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int DS = (d + VEC - 1) / VEC * VEC;
  const int tot = kbn * DS;
  uint32_t *rlist = reinterpret_cast<uint32_t *>(sums + tot);   // [HSGK_CHUNK] (row << 10 | label)
  for (int i = tid; i < tot; i += 256) sums[i] = 0.0f;
  const int64_t row0 = chunk_row0[c];
  chunk_accumulate<VEC, UNROLL, int32_t>(x + row0 * d, d, DS, klab + row0, chunk_rows[c], kb0,
                                         kbn, sums, rlist, wcount);
  __syncthreads();
  float *out = partial + ((int64_t)c * K + kb0) * d;
  for (int k = w; k < kbn; k += 4)
    for (int i = lane; i < d; i += 64) out[k * d + i] = sums[k * DS + i];
}

int launch_accumulate(const float *x, int d, const int32_t *klab, const ChunkTable &t,
                      int max_chunks, int K, float *partial,
                      const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  const bool wide = d >= 256;
  const int DS = wide ? (d + 3) / 4 * 4 : d;
  // cluster rows per pass so that the table fits LDS (two workgroups per CU
  // when possible: <= 76 KiB each).
  const size_t list_bytes = (size_t)HSGK_CHUNK * 4;
  const size_t budget2 = 78 * 1024 - list_bytes, budget1 = 156 * 1024 - list_bytes;
  int kbn = K;
  if ((size_t)kbn * DS * 4 > budget2) {
    kbn = (int)(budget1 / ((size_t)DS * 4));
    if (kbn > K) kbn = K;
  }
  if (kbn > 1024) kbn = 1024;               // label field of the row list is 10 bits
  HSGK_REQUIRE(kbn >= 1, "row too long for the LDS segment table");
  // 16-byte global loads only need dword alignment on gfx950.
  auto kern = wide ? accumulate_kernel<4, 8> : accumulate_kernel<1, 16>;
  size_t lds = (size_t)kbn * DS * 4 + list_bytes;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(158 * 1024)));
  for (int kb0 = 0; kb0 < K; kb0 += kbn) {
    int cur = K - kb0 < kbn ? K - kb0 : kbn;
    hipLaunchKernelGGL(kern, dim3(max_chunks), dim3(256), lds, s, x, d, klab,
                       t.chunk_row0, t.chunk_rows, K, kb0, cur, partial, meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

// ===========================================================================
// M-step, stage 2: sum the chunk partials of one (image, cluster) in chunk
// order, normalise, write the centroid row.  grid = (K, B).
__global__ __launch_bounds__(256) void finalize_kernel(
    const float *__restrict__ partial, int d, int K,
    const int32_t *__restrict__ img_chunk0, float eps, float *__restrict__ cent) {
  extern __shared__ float row[];    // [d] + 1
  const int k = blockIdx.x, b = blockIdx.y;
  const int c0 = img_chunk0[b], c1 = img_chunk0[b + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < d; i += 256) {
    float tsum = 0.0f;
    for (int c = c0; c < c1; ++c) tsum = tsum + partial[((int64_t)c * K + k) * d + i];
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

int launch_finalize(const float *partial, int d, int K, int B, const ChunkTable &t,
                    float eps, float *cent, hipStream_t s) {
  if (B <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(finalize_kernel, dim3(K, B), dim3(256), (size_t)(d + 1) * 4, s,
                     partial, d, K, t.img_chunk0, eps, cent);
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
        klab[row] = bi;
        if (best) best[row] = bv;
      } else if (bv > best[row]) {              // later blocks win only strictly
        klab[row] = bi;
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
struct SplitEpi {
  int K, nrows;
  int64_t crow0;
  int32_t *klab;
  uint16_t *qrows;       // LDS: this workgroup's queue (row offsets inside its range)
  int *qn;               // LDS counter
  __device__ inline void operator()(int tile, const f32x16 (&mainacc)[2],
                                    const f32x16 (&corracc)[2]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    // branch-free running top-2: b2 = max(b2, min(b1, v)); b1 = max(b1, v)
    float b1 = -INFINITY, b2 = -INFINITY;
    int bi = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = k < K ? mainacc[m][r] + corracc[m][r] : -INFINITY;
        b2 = fmaxf(b2, fminf(b1, v));
        bi = v > b1 ? k : bi;
        b1 = fmaxf(b1, v);
      }
    const float o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
    const int oi = __shfl_xor(bi, 32);
    // merged top-2 of the two halves
    float t1, t2;
    int ti;
    if (o1 > b1) { t1 = o1; ti = oi; t2 = fmaxf(b1, o2); }
    else { t1 = b1; ti = bi; t2 = fmaxf(o1, b2); }
    const int px = tile * TPX + w * 32 + j;
    if (h == 0 && px < nrows) {
      klab[crow0 + px] = ti;
      if (!(t1 - t2 > kSplitGap)) {           // ambiguous (or NaN): exact pass decides
        const int pos = atomicAdd(qn, 1);
        qrows[pos] = (uint16_t)px;
      }
    }
  }
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void assign_split_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int32_t *__restrict__ klab,
    int2 *__restrict__ gqueue, int32_t *__restrict__ gcount, int split,
    const hsgk_segkm_meta *__restrict__ meta) {
  constexpr int TPX = NW * 32;
  // All LDS comes from ONE dynamic array: a static __shared__ object in front
  // of it would leave the base only 4-byte aligned and every ds_read_b128 of the
  // engine would be split (measured: LDS 87 % busy, 2x slower kernel).
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char *tail = lds_raw + split_lds_bytes<NW>(d);
  int *qnp = reinterpret_cast<int *>(tail - 16);            // [0] count, [1] global base
  uint16_t *qlist = reinterpret_cast<uint16_t *>(tail);     // [HSGK_CHUNK]
  if (threadIdx.x == 0) qnp[0] = 0;
  const int c = blockIdx.x / split;
  if (c >= meta->n_chunks) return;
  const int part = blockIdx.x - c * split;
  const int tps = (HSGK_CHUNK / TPX + split - 1) / split;
  const int nrows = min(chunk_rows[c] - part * tps * TPX, tps * TPX);
  if (nrows <= 0) return;
  const int64_t crow0 = chunk_row0[c] + (int64_t)part * tps * TPX;
  const int b = chunk_img[c];
  SplitEpi epi{K, nrows, crow0, klab, qlist, qnp};
  score_tiles_split<NW>(x, d, cent + (int64_t)b * K * d, K, crow0, nrows, lds_raw, epi);
  __syncthreads();
  // publish this workgroup's ambiguous rows to the global queue (one atomic per
  // workgroup); the re-score pass balances them over the whole chip
  const int qn = qnp[0];
  if (qn > 0) {
    if (threadIdx.x == 0) qnp[1] = atomicAdd(gcount, qn);
    __syncthreads();
    const int base = qnp[1];
    for (int i = threadIdx.x; i < qn; i += NW * 64)
      gqueue[base + i] = make_int2((int)(crow0 + qlist[i]), b);
  }
}

// Exact re-score of the queued rows (typically < 1 % of a chunk): one WAVE per
// row, lane k owns centroid k and runs the canonical C1 chain
// acc = fmaf(c_k[dd], x[dd], acc) over ascending dd -- bit-identical to the
// fp32 MFMA engine.  The row is held across the wave (4 floats per lane) and
// broadcast one element at a time with v_readlane; centroid rows are read
// straight from global memory as 16-byte pieces (64 rows x 128-byte lines = an
// 8 KiB working set that lives in the CU's L1), so there is no LDS staging and
// no per-workgroup setup cost.  Ties keep the lowest index.
__global__ __launch_bounds__(256) void assign_requeue_rows_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K,
    int32_t *__restrict__ klab, const int2 *__restrict__ gqueue,
    const int32_t *__restrict__ gcount) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  const int total = *gcount;
  const int d4 = d & ~3;
  for (int e = wave; e < total; e += nwaves) {
    const int2 ent = gqueue[e];
    const int64_t row = ent.x;
    const float *xr = x + row * d;
    const float *ck = cent + ((int64_t)ent.y * K + min(lane, K - 1)) * d;   // this lane's centroid
    float acc = 0.0f;
    for (int base = 0; base < d4; base += 256) {
      // lanes hold 256 consecutive row elements, 4 per lane
      const int mine = base + 4 * lane;
      float xv[4] = {0.f, 0.f, 0.f, 0.f};
      if (mine < d4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = xr[mine + i];
      }
      const int cnt = min(256, d4 - base);
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));    // rows are 8-byte aligned
      for (int t0 = 0; t0 < cnt; t0 += 32) {                                 // 8 pieces in flight
        f4u cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          cv[u] = *reinterpret_cast<const f4u *>(ck + base + min(t0 + 4 * u, cnt - 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + 4 * u;
          if (t < cnt) {
            const int src = t >> 2;
            acc = fmaf(cv[u].x, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[0]), src)), acc);
            acc = fmaf(cv[u].y, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[1]), src)), acc);
            acc = fmaf(cv[u].z, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[2]), src)), acc);
            acc = fmaf(cv[u].w, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[3]), src)), acc);
          }
        }
      }
    }
    for (int dd = d4; dd < d; ++dd) acc = fmaf(ck[dd], xr[dd], acc);
    // wave argmax, first index on ties
    float bv = lane < K ? acc : -INFINITY;
    int bi = lane;
    if (!(bv == bv)) bv = -INFINITY;               // NaN never wins
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) klab[row] = bi < K ? bi : 0;
  }
}

static int launch_assign_split(const float *x, int d, const float *cent, int K,
                               const ChunkTable &t, int max_chunks, int32_t *klab,
                               int2 *gqueue, int32_t *gcount, const hsgk_segkm_meta *meta,
                               hipStream_t s) {
  constexpr int NW = 8, TPX = NW * 32, kTiles = HSGK_CHUNK / TPX;
  int split = 1;
  while (split < kTiles && (int64_t)max_chunks * split < 2048) split *= 2;
  const int grid = max_chunks * split;
  HSGK_CHECK_HIP(hipMemsetAsync(gcount, 0, sizeof(int32_t), s));
  {
    auto kern = assign_split_kernel<NW>;
    const size_t lds = split_lds_bytes<NW>(d) + (size_t)HSGK_CHUNK * sizeof(uint16_t);
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, x, d, cent, K, t.chunk_row0,
                       t.chunk_rows, t.chunk_img, klab, gqueue, gcount, split, meta);
    HSGK_LAUNCH_CHECK();
  }
  // exact pass over the (chip-wide balanced) queue; its length is only known on
  // the device, so a fixed grid strides over it
  hipLaunchKernelGGL(assign_requeue_rows_kernel, dim3(2048), dim3(256), 0, s, x, d, cent, K, klab,
                     gqueue, gcount);
  HSGK_LAUNCH_CHECK();
  return 0;
}

bool assign_split_eligible(int d, int K) {
  return K <= 64 && split_shape_ok(d) &&
         split_lds_bytes<8>(d) + (size_t)HSGK_CHUNK * sizeof(uint16_t) <= 160 * 1024;
}

int launch_assign_fast(const float *x, int d, const float *cent, int K, const ChunkTable &t,
                       int max_chunks, int32_t *klab, float *best, int2 *qrows,
                       int32_t *qcount, const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  static const int mode = [] {
    const char *e = getenv("HSGK_ASSIGN");          // "fp32" forces the exact kernel only
    return (e && e[0] == 'f') ? 0 : 1;
  }();
  if (mode == 1 && qrows && assign_split_eligible(d, K))
    return launch_assign_split(x, d, cent, K, t, max_chunks, klab, qrows, qcount, meta, s);
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
