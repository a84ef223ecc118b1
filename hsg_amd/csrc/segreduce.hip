// segreduce.hip -- stand-alone segment reductions over flat row sets:
//   mode 0  calculate_prototypes_from_labels (hsg/utils/segsort/common.py:11-41):
//           L2-normalised segment sums (empty segment -> exact zero row)
//   mode 1  segment_mean (hsg/utils/general/common.py:123-147): sum / count,
//           empty segments divide by 1
//   mode 2  raw sums (the per-rank payload of the RCCL prototype exchange)
// plus their backward passes.  Summation order = C2 with chunks of 2048
// consecutive rows of the flat row list.
//
// Any P is supported as long as the labels inside one chunk span at most
// RMAX (= min(P, 512)) consecutive ids -- true for every call site of the
// reference (ids are sorted ranks of (image, cluster, label), rows are image
// major); a wider chunk raises status 1.
#include "accumulate.h"
#include "common.h"

namespace hsgk {

constexpr int kSegRmax = 512;

struct SegWin { int64_t lmin; int64_t range; };

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void segreduce_chunk_kernel(
    const float *__restrict__ x, int64_t n, int d, const int64_t *__restrict__ labels,
    int64_t P, int rmax, int rlds, float *__restrict__ partial, SegWin *__restrict__ win,
    int32_t *__restrict__ status) {
  extern __shared__ float sums[];   // [rlds][DS] then the row list
  __shared__ int wcount[4];
  __shared__ int64_t red[8];
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int DS = (d + VEC - 1) / VEC * VEC;
  uint32_t *rlist = reinterpret_cast<uint32_t *>(sums + rlds * DS);
  const int64_t row0 = (int64_t)c * HSGK_CHUNK;
  const int nrows = (int)((n - row0) < HSGK_CHUNK ? (n - row0) : HSGK_CHUNK);
  const int64_t *lab = labels + row0;

  // label range of this chunk (labels outside [0,P) are ignored)
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  for (int r = tid; r < nrows; r += 256) {
    const int64_t l = lab[r];
    if (l >= 0 && l < P) { lo = l < lo ? l : lo; hi = l > hi ? l : hi; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (lane == 0) { red[w] = lo; red[4 + w] = hi; }
  __syncthreads();
  lo = red[0]; hi = red[4];
  for (int i = 1; i < 4; ++i) { lo = red[i] < lo ? red[i] : lo; hi = red[4 + i] > hi ? red[4 + i] : hi; }
  int64_t range = hi >= lo ? hi - lo + 1 : 0;
  if (range > rmax) {
    if (tid == 0) atomicMax(status, 1);
    range = rmax;
  }
  if (tid == 0) { win[c].lmin = range ? lo : 0; win[c].range = range; }

  float *out = partial + (int64_t)c * rmax * d;
  for (int64_t p0 = 0; p0 < range; p0 += rlds) {
    const int cur = (int)((range - p0) < rlds ? (range - p0) : rlds);
    __syncthreads();
    for (int i = tid; i < cur * DS; i += 256) sums[i] = 0.0f;
    chunk_accumulate<VEC, UNROLL, int64_t>(x + row0 * d, d, DS, lab, nrows, lo + p0, cur, sums,
                                           rlist, wcount);
    __syncthreads();
    for (int k = w; k < cur; k += 4)
      for (int i = lane; i < d; i += 64) out[(p0 + k) * d + i] = sums[k * DS + i];
  }
}

__global__ void count_labels_kernel(const int64_t *__restrict__ labels, int64_t n, int64_t P,
                                    int32_t *__restrict__ counts) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = labels[r];
    if (l >= 0 && l < P) atomicAdd(&counts[l], 1);
  }
}

// One workgroup per segment: chunk partials in chunk order (C2), then the
// mode's epilogue.  aux[k] = clamped norm (mode 0) or count (mode 1).
__global__ __launch_bounds__(256) void segreduce_final_kernel(
    const float *__restrict__ partial, const SegWin *__restrict__ win, int nchunks, int rmax,
    int d, int mode, float eps, const int32_t *__restrict__ counts, float *__restrict__ out,
    float *__restrict__ aux) {
  extern __shared__ float row[];             // [d + 1] then the chunk list
  __shared__ int wn[4];
  __shared__ int total;
  int32_t *clist = reinterpret_cast<int32_t *>(row + d + 1);   // [nchunks]
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int c = c0 + tid;
    bool hit = false;
    if (c < nchunks) {
      const SegWin wv = win[c];
      hit = k >= wv.lmin && k < wv.lmin + wv.range;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wn[w] = __popcll(m);
    __syncthreads();
    int base = total;
    for (int i = 0; i < w; ++i) base += wn[i];
    if (hit) clist[base + __popcll(m & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    if (tid == 0) total += wn[0] + wn[1] + wn[2] + wn[3];
    __syncthreads();
  }
  const int nl = total;
  for (int i = tid; i < d; i += 256) {
    float t = 0.0f;
    for (int q = 0; q < nl; ++q) {
      const int c = clist[q];
      t = t + partial[((int64_t)c * rmax + (k - win[c].lmin)) * d + i];
    }
    row[i] = t;
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

}  // namespace hsgk

using namespace hsgk;

static int seg_rmax(int64_t P) { return (int)(P < kSegRmax ? (P < 1 ? 1 : P) : kSegRmax); }

extern "C" {

size_t hsgk_segment_reduce_workspace_bytes(int64_t n, int d, int64_t P) {
  const int64_t nch = (n + HSGK_CHUNK - 1) / HSGK_CHUNK;
  Carver cv(nullptr);
  cv.take<float>((size_t)(nch > 0 ? nch : 1) * seg_rmax(P) * d);
  cv.take<SegWin>((size_t)(nch > 0 ? nch : 1));
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
  const int nch = (int)((n + HSGK_CHUNK - 1) / HSGK_CHUNK);
  const int rmax = seg_rmax(P);
  Carver cv(workspace);
  float *partial = cv.take<float>((size_t)(nch > 0 ? nch : 1) * rmax * d);
  SegWin *win = cv.take<SegWin>((size_t)(nch > 0 ? nch : 1));
  int32_t *counts = cv.take<int32_t>((size_t)P);

  const bool wide = d >= 256;
  const int DS = wide ? (d + 3) / 4 * 4 : d;
  const size_t list_bytes = (size_t)HSGK_CHUNK * 4;
  // table rows per pass: two workgroups per CU when that still covers a useful
  // window, otherwise one workgroup with the whole LDS
  const int rl2 = (int)((76 * 1024 - list_bytes) / ((size_t)DS * 4));
  const int rl1 = (int)((150 * 1024 - list_bytes) / ((size_t)DS * 4));
  int rlds = rl2 >= (rmax < 64 ? rmax : 64) ? rl2 : rl1;
  if (rlds > rmax) rlds = rmax;
  if (rlds > 1024) rlds = 1024;
  HSGK_REQUIRE(rlds >= 1, "row too long for the LDS segment table");
  if (nch > 0) {
    auto kern = wide ? segreduce_chunk_kernel<4, 8> : segreduce_chunk_kernel<1, 16>;
    const size_t lds = (size_t)rlds * DS * 4 + list_bytes;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(158 * 1024)));
    hipLaunchKernelGGL(kern, dim3(nch), dim3(256), lds, s, x, n, d, labels, P, rmax, rlds, partial,
                       win, status);
    HSGK_LAUNCH_CHECK();
  }
  if (mode == 1) {
    HSGK_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)P * 4, s));
    if (n > 0) {
      int64_t g = (n + 255) / 256;
      hipLaunchKernelGGL(count_labels_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, s,
                         labels, n, P, counts);
      HSGK_LAUNCH_CHECK();
    }
  }
  const size_t lds = (size_t)(d + 1) * 4 + (size_t)(nch > 0 ? nch : 1) * 4;
  HSGK_REQUIRE(lds <= 150 * 1024, "too many chunks for the finalize list");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(segreduce_final_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)));
  hipLaunchKernelGGL(segreduce_final_kernel, dim3((unsigned)P), dim3(256), lds, s, partial, win, nch,
                     rmax, d, mode, eps, counts, out, aux);
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
