// segreduce.hip -- stand-alone segment reductions over flat row sets:
//   mode 0  calculate_prototypes_from_labels (hsg/utils/segsort/common.py:11-41):
//           L2-normalised segment sums (empty segment -> exact zero row)
//   mode 1  segment_mean (hsg/utils/general/common.py:123-147): sum / count,
//           empty segments divide by 1
//   mode 2  raw sums (the per-rank payload of the RCCL prototype exchange)
// plus their backward passes.  Summation order = C2 with chunks of 2048
// consecutive rows of the flat row list.
//
// Any labels are supported.  A chunk's partial sums are stored by the RANK of a
// segment id among the distinct ids of the chunk (a presence bitmap over the chunk's id
// range, prefix popcounts), so the id range of a chunk does not matter as long as it has
// at most RMAX (= min(P, 512)) distinct ids -- every realistic input: sorted ranks of
// (image, cluster, label) over image-major rows, label-split clusters, several small
// images per chunk.  A chunk with more distinct ids than that (fewer than 4 rows per
// segment on average) or an id range beyond the bitmap is left to the per-segment
// kernel, which scans such a chunk's labels itself and adds up the matching rows in row
// order: the same canonical order C2, no extra memory, just slower.
// Labels outside [0, P) are skipped and flagged in *status (bit 1).
#include "accumulate.h"
#include "common.h"

namespace hsgk {

constexpr int kSegRmax = 512;
constexpr int kSegBitWords = 1024;                 // presence bitmap: 32768 ids per chunk
constexpr int kSegBitRange = kSegBitWords * 32;

// lmin / lmax: id range of the chunk's valid labels (lmax < lmin: none); ndist: distinct ids
// (fast chunks: their partial rows are partial[c][0 .. ndist), ids in seg_ids[c][.]); slow = 1:
// no partial rows, the per-segment kernel scans the chunk
struct SegWin { int64_t lmin; int64_t lmax; int32_t ndist; int32_t slow; };

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void segreduce_chunk_kernel(
    const float *__restrict__ x, int64_t n, int d, const int64_t *__restrict__ labels,
    int64_t P, int rmax, int rlds, int tab_floats, float *__restrict__ partial, SegWin *__restrict__ win,
    int64_t *__restrict__ seg_ids, int32_t *__restrict__ status) {
  extern __shared__ float sums[];   // [rlds][DS], the row list, the bitmap, its prefix, the slots
  __shared__ int wcount[4];
  __shared__ int64_t red[8];
  __shared__ int wtot[4];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int DS = (d + VEC - 1) / VEC * VEC;
  uint32_t *rlist = reinterpret_cast<uint32_t *>(sums + tab_floats);     // [HSGK_CHUNK]; tab_floats >= rlds * DS
  int16_t *slots = reinterpret_cast<int16_t *>(rlist + HSGK_CHUNK);       // [HSGK_CHUNK]
  // the presence bitmap and its prefix are only needed until the slots are known: they live in the
  // (not yet zeroed) sum table, which keeps the workgroup at <= 78 KiB for 64 table rows of d = 258
  uint32_t *bits = reinterpret_cast<uint32_t *>(sums);                    // [kSegBitWords]
  int32_t *wpre = reinterpret_cast<int32_t *>(bits + kSegBitWords);       // [kSegBitWords]
  const int64_t row0 = (int64_t)c * HSGK_CHUNK;
  const int nrows = (int)((n - row0) < HSGK_CHUNK ? (n - row0) : HSGK_CHUNK);
  const int64_t *lab = labels + row0;

  // label range of this chunk (labels outside [0,P) are skipped and flagged)
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  bool bad = false;
  for (int r = tid; r < nrows; r += 256) {
    const int64_t l = lab[r];
    if (l >= 0 && l < P) { lo = l < lo ? l : lo; hi = l > hi ? l : hi; } else bad = true;
  }
  if (__ballot(bad) && lane == 0) atomicOr(status, 2);
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (lane == 0) { red[w] = lo; red[4 + w] = hi; }
  for (int i = tid; i < kSegBitWords; i += 256) bits[i] = 0u;
  __syncthreads();
  lo = red[0]; hi = red[4];
  for (int i = 1; i < 4; ++i) { lo = red[i] < lo ? red[i] : lo; hi = red[4 + i] > hi ? red[4 + i] : hi; }
  if (hi < lo) {                                   // no valid row in this chunk
    if (tid == 0) win[c] = SegWin{0, -1, 0, 0};
    return;
  }
  if (hi - lo >= kSegBitRange) {
    if (tid == 0) win[c] = SegWin{lo, hi, 0, 1};
    return;
  }
  // presence bitmap over [lo, hi] -> rank of every id among the chunk's distinct ids
  for (int r = tid; r < nrows; r += 256) {
    const int64_t l = lab[r];
    if (l >= 0 && l < P) atomicOr(&bits[(l - lo) >> 5], 1u << ((l - lo) & 31));
  }
  __syncthreads();
  {
    int cnt[kSegBitWords / 256], tot = 0;
#pragma unroll
    for (int i = 0; i < kSegBitWords / 256; ++i) { cnt[i] = __popc(bits[tid * (kSegBitWords / 256) + i]); tot += cnt[i]; }
    int inc = tot;                                  // inclusive scan over the 256 threads
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int i = 0; i < w; ++i) base += wtot[i];
#pragma unroll
    for (int i = 0; i < kSegBitWords / 256; ++i) { wpre[tid * (kSegBitWords / 256) + i] = base; base += cnt[i]; }
  }
  __syncthreads();
  const int ndist = wtot[0] + wtot[1] + wtot[2] + wtot[3];
  if (ndist > rmax) {
    if (tid == 0) win[c] = SegWin{lo, hi, ndist, 1};
    return;
  }
  if (tid == 0) win[c] = SegWin{lo, hi, ndist, 0};
  int64_t *ids = seg_ids + (int64_t)c * rmax;
  for (int i = tid; i < kSegBitWords; i += 256) {
    uint32_t m = bits[i];
    int pos = wpre[i];
    while (m) {
      const int b = __ffs(m) - 1;
      ids[pos++] = lo + 32 * i + b;
      m &= m - 1;
    }
  }
  for (int r = tid; r < HSGK_CHUNK; r += 256) {
    int sl = -1;
    if (r < nrows) {
      const int64_t l = lab[r];
      if (l >= 0 && l < P) {
        const int64_t o = l - lo;
        sl = wpre[o >> 5] + __popc(bits[o >> 5] & ((1u << (o & 31)) - 1u));
      }
    }
    slots[r] = (int16_t)sl;
  }

  float *out = partial + (int64_t)c * rmax * d;
  for (int p0 = 0; p0 < ndist; p0 += rlds) {
    const int cur = (ndist - p0) < rlds ? (ndist - p0) : rlds;
    __syncthreads();
    for (int i = tid; i < cur * DS; i += 256) sums[i] = 0.0f;
    chunk_accumulate<VEC, UNROLL, int16_t>(x + row0 * d, d, DS, slots, nrows, p0, cur, sums,
                                           rlist, wcount);
    __syncthreads();
    for (int k = w; k < cur; k += 4)
      for (int i = lane; i < d; i += 64) out[(int64_t)(p0 + k) * d + i] = sums[k * DS + i];
  }
}

// lds_slots >= P: a workgroup counts in LDS and adds its non-zero slots to the global counters once (one global
// atomic per row on ~1 K hot counters took 0.39 ms for 2.4 M rows); otherwise straight to global memory
__global__ void count_labels_kernel(const int64_t *__restrict__ labels, int64_t n, int64_t P,
                                    int32_t *__restrict__ counts, int lds_slots) {
  extern __shared__ int cl_hist[];
  const bool local = lds_slots >= P;
  if (local) {
    for (int i = threadIdx.x; i < (int)P; i += blockDim.x) cl_hist[i] = 0;
    __syncthreads();
  }
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = labels[r];
    if (l >= 0 && l < P) atomicAdd(local ? &cl_hist[l] : &counts[l], 1);
  }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < (int)P; i += blockDim.x) {
      const int v = cl_hist[i];
      if (v) atomicAdd(&counts[i], v);
    }
  }
}
static void launch_count_labels(const int64_t *labels, int64_t n, int64_t P, int32_t *counts, hipStream_t s) {
  const int64_t g = (n + 255) / 256;
  const bool local = P <= 8192 && n >= 16 * P;
  const int64_t cap = local ? 512 : 2048;
  hipLaunchKernelGGL(count_labels_kernel, dim3((unsigned)(g > cap ? cap : g)), dim3(256), local ? (size_t)P * 4 : 0, s,
                     labels, n, P, counts, local ? (int)P : 0);
}

// One workgroup per segment: chunk partials in chunk order (C2), then the
// mode's epilogue.  aux[k] = clamped norm (mode 0) or count (mode 1).  A slow chunk has no
// partial row: its labels are scanned here and the matching rows summed in row order.
__global__ __launch_bounds__(256) void segreduce_final_kernel(
    const float *__restrict__ partial, const SegWin *__restrict__ win, const int64_t *__restrict__ seg_ids,
    int nchunks, int rmax, int d, int mode, float eps, const int32_t *__restrict__ counts,
    const float *__restrict__ x, const int64_t *__restrict__ labels, int64_t n,
    float *__restrict__ out, float *__restrict__ aux) {
  extern __shared__ float row[];             // [d + 1], the chunk list, the row list of a slow chunk
  __shared__ int wn[4];
  __shared__ int total;
  int32_t *clist = reinterpret_cast<int32_t *>(row + d + 1);   // [nchunks][2]: chunk, slot (-1: slow)
  uint16_t *rows = reinterpret_cast<uint16_t *>(clist + 2 * (size_t)nchunks);   // [HSGK_CHUNK]
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int c = c0 + tid;
    bool hit = false;
    int slot = -1;
    if (c < nchunks) {
      const SegWin wv = win[c];
      if (k >= wv.lmin && k <= wv.lmax) {
        if (wv.slow) {
          hit = true;
        } else {                                       // binary search in the chunk's sorted id list
          const int64_t *ids = seg_ids + (int64_t)c * rmax;
          int a = 0, b = wv.ndist - 1;
          while (a < b) {
            const int m = (a + b) >> 1;
            if (ids[m] < k) a = m + 1; else b = m;
          }
          if (ids[a] == k) { hit = true; slot = a; }
        }
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wn[w] = __popcll(m);
    __syncthreads();
    int base = total;
    for (int i = 0; i < w; ++i) base += wn[i];
    if (hit) {
      const int q = base + __popcll(m & ((1ull << lane) - 1ull));
      clist[2 * q] = c;
      clist[2 * q + 1] = slot;
    }
    __syncthreads();
    if (tid == 0) total += wn[0] + wn[1] + wn[2] + wn[3];
    __syncthreads();
  }
  const int nl = total;
  float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};              // columns tid, tid + 256, ... (d <= 1024 here; wider: loop)
  const int ncol = (d + 255) / 256;
  for (int cb = 0; cb < ncol; cb += 4) {
    for (int u = 0; u < 4; ++u) t[u] = 0.0f;
    for (int q = 0; q < nl; ++q) {
      const int c = clist[2 * q], slot = clist[2 * q + 1];
      if (slot >= 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (cb + u) * 256 + tid;
          if (i < d) t[u] = t[u] + partial[((int64_t)c * rmax + slot) * d + i];
        }
      } else {
        // slow chunk: ordered list of its rows that carry label k, then their sum in row order
        const int64_t r0 = (int64_t)c * HSGK_CHUNK;
        const int nr = (int)((n - r0) < HSGK_CHUNK ? (n - r0) : HSGK_CHUNK);
        __syncthreads();
        if (tid == 0) total = 0;
        __syncthreads();
        for (int b0 = 0; b0 < nr; b0 += 256) {
          const int r = b0 + tid;
          const bool mine = r < nr && labels[r0 + r] == k;
          const unsigned long long m = __ballot(mine);
          if (lane == 0) wn[w] = __popcll(m);
          __syncthreads();
          int base = total;
          for (int i = 0; i < w; ++i) base += wn[i];
          if (mine) rows[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)r;
          __syncthreads();
          if (tid == 0) total += wn[0] + wn[1] + wn[2] + wn[3];
          __syncthreads();
        }
        const int nm = total;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = (cb + u) * 256 + tid;
          if (i < d) {
            float sacc = 0.0f;
            for (int e = 0; e < nm; ++e) sacc = sacc + x[(r0 + rows[e]) * d + i];
            t[u] = t[u] + sacc;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = (cb + u) * 256 + tid;
      if (i < d) row[i] = t[u];
    }
  }
  __syncthreads();
  if (tid == 0) {
    float a = 1.0f;
    if (mode == 0) {
      float ss = 0.0f;
      for (int i = 0; i < d; ++i) ss = fmaf(row[i], row[i], ss);
      a = sqrtf(ss);
      if (!(a >= eps)) a = eps;
    } else if (mode == 1) {
      a = (float)counts[k];
      if (a == 0.0f) a = 1.0f;
    }
    row[d] = a;
    if (aux) aux[k] = a;
  }
  __syncthreads();
  const float a = row[d];
  for (int i = tid; i < d; i += 256) out[k * d + i] = mode == 2 ? row[i] : row[i] / a;
}

// ---- backward -------------------------------------------------------------
// d(out)/d(sum): mode 0  (g - out * <out, g>) / norm   (norm >= eps branch; the
// eps branch degenerates to g / eps because out = s / eps is linear there);
// mode 1  g / count;  mode 2  g.
__global__ __launch_bounds__(256) void segreduce_bwd_seg_kernel(
    const float *__restrict__ gout, const float *__restrict__ out, const float *__restrict__ aux,
    int d, int mode, float eps, float *__restrict__ gseg) {
  __shared__ float wsum[4];
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x;
  const float a = mode == 2 ? 1.0f : aux[k];
  float dot = 0.0f;
  if (mode == 0 && a > eps) {
    for (int i = tid; i < d; i += 256) dot += out[k * d + i] * gout[k * d + i];
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((tid & 63) == 0) wsum[tid >> 6] = dot;
    __syncthreads();
    dot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  for (int i = tid; i < d; i += 256) {
    float g = gout[k * d + i];
    if (mode == 0 && a > eps) g = g - out[k * d + i] * dot;
    gseg[k * d + i] = g / a;
  }
}

__global__ __launch_bounds__(256) void segreduce_bwd_rows_kernel(
    const float *__restrict__ gseg, const int64_t *__restrict__ labels, int64_t n, int d,
    int64_t P, float *__restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n; r += nwaves) {
    const int64_t l = labels[r];
    const bool ok = l >= 0 && l < P;
    for (int i = lane; i < d; i += 64) gx[r * d + i] = ok ? gseg[l * d + i] : 0.0f;
  }
}

// raw sums -> the mode's output: 0 L2-normalised (C1 norm chain, eps clamp; aux = norm), 1 mean (aux = count,
// 0 -> 1), 2 raw.  One workgroup per segment.
__global__ __launch_bounds__(256) void segreduce_epilogue_kernel(const float *__restrict__ sums, int d, int mode, float eps,
                                                                 const int32_t *__restrict__ counts,
                                                                 float *__restrict__ out, float *__restrict__ aux) {
  extern __shared__ float row[];             // [d + 1]
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < d; i += 256) row[i] = sums[k * d + i];
  __syncthreads();
  if (tid == 0) {
    float a = 1.0f;
    if (mode == 0) {
      float ss = 0.0f;
      for (int i = 0; i < d; ++i) ss = fmaf(row[i], row[i], ss);
      a = sqrtf(ss);
      if (!(a >= eps)) a = eps;
    } else if (mode == 1) {
      a = (float)counts[k];
      if (a == 0.0f) a = 1.0f;
    }
    row[d] = a;
    if (aux) aux[k] = a;
  }
  __syncthreads();
  const float a = row[d];
  for (int i = tid; i < d; i += 256) out[k * d + i] = row[i] / a;
}

}  // namespace hsgk

using namespace hsgk;

// Round 3: rows of up to 1024 columns take the sorted-run kernels of exchange.hip (one row set: the rows of a chunk
// stably sorted by id, a wave per column slice, runs summed in registers -- no LDS table, any ids; 5.3 TB/s against
// 3.7 for the LDS-table kernel below, same order C2, same bits).  Wider rows keep the LDS-table kernels.
static bool seg_sorted_path(int d) { return d <= 1024; }
static int64_t seg_pool_rows(int64_t n, int64_t P) {
  const int64_t nch = (n + HSGK_CHUNK - 1) / HSGK_CHUNK;
  int64_t r = nch * (P < 256 ? P : 256);
  if (r > n) r = n;
  return r < 1 ? 1 : r;
}

static int seg_rmax(int64_t P) { return (int)(P < kSegRmax ? (P < 1 ? 1 : P) : kSegRmax); }

extern "C" {

size_t hsgk_segment_reduce_workspace_bytes(int64_t n, int d, int64_t P) {
  const int64_t nch = (n + HSGK_CHUNK - 1) / HSGK_CHUNK;
  if (seg_sorted_path(d)) {
    Carver cv(nullptr);
    const int64_t pr = seg_pool_rows(n, P);
    cv.take<char>((size_t)(nch > 0 ? nch : 1) * sorted_sums_chunk_bytes());
    cv.take<int32_t>((size_t)2 * (P > 0 ? P : 1));
    cv.take<int64_t>((size_t)pr);
    cv.take<float>((size_t)pr * d);
    cv.take<int32_t>(64);
    cv.take<int32_t>((size_t)(P > 0 ? P : 1));
    cv.take<float>((size_t)(P > 0 ? P : 1) * d);
    return cv.off + 256;
  }
  Carver cv(nullptr);
  cv.take<float>((size_t)(nch > 0 ? nch : 1) * seg_rmax(P) * d);
  cv.take<SegWin>((size_t)(nch > 0 ? nch : 1));
  cv.take<int64_t>((size_t)(nch > 0 ? nch : 1) * seg_rmax(P));
  cv.take<int32_t>((size_t)(P > 0 ? P : 1));
  return cv.off + 256;
}

int hsgk_segment_reduce(const float *x, int64_t n, int d, const int64_t *labels, int64_t P,
                        int mode, float eps, float *out, float *aux, int32_t *status,
                        void *workspace, size_t workspace_bytes, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && d >= 1 && P >= 0 && mode >= 0 && mode <= 2, "bad arguments");
  HSGK_REQUIRE(status != nullptr, "status pointer required");
  HSGK_REQUIRE(workspace_bytes >= hsgk_segment_reduce_workspace_bytes(n, d, P), "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  HSGK_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  if (P == 0) return 0;
  if (seg_sorted_path(d)) {
    const int64_t nchl = (n + HSGK_CHUNK - 1) / HSGK_CHUNK;
    const int64_t pr = seg_pool_rows(n, P);
    Carver cv(workspace);
    char *chunks = cv.take<char>((size_t)(nchl > 0 ? nchl : 1) * sorted_sums_chunk_bytes());
    int32_t *seg_range = cv.take<int32_t>((size_t)2 * P);
    int64_t *pool_ids = cv.take<int64_t>((size_t)pr);
    float *pool = cv.take<float>((size_t)pr * d);
    int32_t *pool_used = cv.take<int32_t>(64);
    int32_t *counts = cv.take<int32_t>((size_t)P);
    float *sums = cv.take<float>((size_t)P * d);
    HSGK_CHECK_HIP(hipMemsetAsync(pool_used, 0, sizeof(int32_t), s));
    float *table = mode == 2 ? out : sums;
    if (int rc = launch_sorted_sums(x, d, nullptr, 0, labels, n, P, chunks, seg_range, pool_ids, pool, pr, pool_used,
                                    table, status, s))
      return rc;
    if (mode == 1) {
      HSGK_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)P * 4, s));
      if (n > 0) {
        launch_count_labels(labels, n, P, counts, s);
        HSGK_LAUNCH_CHECK();
      }
    }
    if (mode != 2) {
      hipLaunchKernelGGL(segreduce_epilogue_kernel, dim3((unsigned)P), dim3(256), (size_t)(d + 1) * 4, s, sums, d, mode, eps,
                         counts, out, aux);
      HSGK_LAUNCH_CHECK();
    }
    return 0;
  }
  const int nch = (int)((n + HSGK_CHUNK - 1) / HSGK_CHUNK);
  const int rmax = seg_rmax(P);
  Carver cv(workspace);
  float *partial = cv.take<float>((size_t)(nch > 0 ? nch : 1) * rmax * d);
  SegWin *win = cv.take<SegWin>((size_t)(nch > 0 ? nch : 1));
  int64_t *seg_ids = cv.take<int64_t>((size_t)(nch > 0 ? nch : 1) * rmax);
  int32_t *counts = cv.take<int32_t>((size_t)P);

  // floats per lane and row load: the first 64 * VEC columns of a row are prefetched 16 rows deep, further
  // full passes are load-use (one exposed latency per row: d = 128 with one float per lane took 293 us for
  // 12.5 K rows), so VEC grows with d (the sums do not depend on it: every column adds its rows in order)
  const int vec = d >= 256 ? 4 : d >= 128 ? 2 : 1;
  const int DS = (d + vec - 1) / vec * vec;
  const size_t list_bytes = (size_t)HSGK_CHUNK * (4 + 2);      // row list, slots (bitmap + prefix alias the table)
  // table rows per pass: two workgroups per CU when that still covers a useful
  // window, otherwise one workgroup with the whole LDS
  const int rl2 = (int)((78 * 1024 - list_bytes) / ((size_t)DS * 4));
  const int rl1 = (int)((150 * 1024 - list_bytes) / ((size_t)DS * 4));
  // (chunks of image-major rows with sorted ids hold a few dozen distinct ids: 32 table rows per pass
  //  are worth the second workgroup per CU; a chunk with more takes further passes over its rows)
  int rlds = rl2 >= (rmax < 32 ? rmax : 32) ? rl2 : rl1;
  if (rlds > rmax) rlds = rmax;
  if (rlds > 1024) rlds = 1024;
  HSGK_REQUIRE(rlds >= 1, "row too long for the LDS segment table");
  size_t tab_bytes = (size_t)rlds * DS * 4;
  if (tab_bytes < (size_t)kSegBitWords * 8) tab_bytes = (size_t)kSegBitWords * 8;      // room for the bitmap + prefix
  if (nch > 0) {
    auto kern = vec == 4 ? segreduce_chunk_kernel<4, 16> : vec == 2 ? segreduce_chunk_kernel<2, 16> : segreduce_chunk_kernel<1, 16>;
    const size_t lds = tab_bytes + list_bytes;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(158 * 1024)));
    hipLaunchKernelGGL(kern, dim3(nch), dim3(256), lds, s, x, n, d, labels, P, rmax, rlds, (int)(tab_bytes / 4), partial,
                       win, seg_ids, status);
    HSGK_LAUNCH_CHECK();
  }
  if (mode == 1) {
    HSGK_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)P * 4, s));
    if (n > 0) {
      launch_count_labels(labels, n, P, counts, s);
      HSGK_LAUNCH_CHECK();
    }
  }
  const size_t lds = (size_t)(d + 1) * 4 + (size_t)(nch > 0 ? nch : 1) * 8 + (size_t)HSGK_CHUNK * 2;
  HSGK_REQUIRE(lds <= 150 * 1024, "too many chunks for the finalize list");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(segreduce_final_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)));
  hipLaunchKernelGGL(segreduce_final_kernel, dim3((unsigned)P), dim3(256), lds, s, partial, win, seg_ids,
                     nch, rmax, d, mode, eps, counts, x, labels, n, out, aux);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_segment_reduce_bwd(const float *gout, const float *out, const float *aux,
                            const int64_t *labels, int64_t n, int d, int64_t P, int mode, float eps,
                            float *gseg, float *gx, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && d >= 1 && P >= 0 && mode >= 0 && mode <= 2, "bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  if (P > 0) {
    hipLaunchKernelGGL(segreduce_bwd_seg_kernel, dim3((unsigned)P), dim3(256), 0, s, gout, out, aux, d,
                       mode, eps, gseg);
    HSGK_LAUNCH_CHECK();
  }
  if (n > 0) {
    int64_t g = (n + 3) / 4;
    hipLaunchKernelGGL(segreduce_bwd_rows_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0,
                       s, gseg, labels, n, d, P, gx);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
