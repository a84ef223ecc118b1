// relabel.hip -- back half of segment_by_kmeans (reference
// hsg/utils/segsort/common.py:398-405): the two nested sorted-`unique`
// relabels.  The final id of a pixel is the rank of its (image, cluster,
// label) triple among the triples PRESENT in the batch, in lexicographic
// order.  Instead of sorting N keys we mark a bounded presence table indexed
// by (image*K + cluster)*L + label and take an exclusive prefix sum over it.
// When the raw label range would make the table too large, labels are first
// replaced by their rank among the distinct label values (monotone, so the
// order of triples is unchanged).  All decisions are taken on the device, so
// the host never synchronises in the middle of the operator.
#include "common.h"

namespace hsgk {

constexpr int kScanBlock = 2048;   // elements per workgroup (256 threads x 8)

__device__ inline int64_t scan_len(const hsgk_segkm_meta *meta, int which, int64_t BK) {
  if (meta->error) return 0;
  if (which == 0) return meta->relabel_mode == 1 ? meta->label_max + 1 : 0;
  return BK * meta->relabel_L;
}

__global__ void decide_kernel(hsgk_segkm_meta *meta, int64_t BK, int64_t cap) {
  if (meta->error) return;
  int64_t L = meta->n_rows > 0 ? meta->label_max + 1 : 1;
  if (BK * L <= cap) {
    meta->relabel_mode = 0;
    meta->relabel_L = L;
  } else {
    meta->relabel_mode = 1;
    meta->relabel_L = 0;
    if (L > cap) meta->error = 2;
  }
}

__global__ void decide2_kernel(hsgk_segkm_meta *meta, int64_t BK, int64_t cap) {
  if (meta->error || meta->relabel_mode != 1) return;
  // relabel_L was set to the number of distinct labels by the label scan
  if (BK * meta->relabel_L > cap) meta->error = 2;
}

__global__ void mark_labels_kernel(const int64_t *__restrict__ labels,
                                   int32_t *__restrict__ lrank,
                                   const hsgk_segkm_meta *__restrict__ meta) {
  if (meta->error || meta->relabel_mode != 1) return;
  const int64_t n = meta->n_rows;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x)
    lrank[labels[r]] = 1;
}

// ---- three-kernel exclusive scan over a device-sized int32 array ----------
__global__ __launch_bounds__(256) void scan_reduce_kernel(
    const int32_t *__restrict__ a, int32_t *__restrict__ bsum,
    const hsgk_segkm_meta *__restrict__ meta, int which, int64_t BK) {
  __shared__ int32_t ws[4];
  const int64_t len = scan_len(meta, which, BK);
  const int64_t base = (int64_t)blockIdx.x * kScanBlock;
  if (base >= len) return;
  int s = 0;
  for (int i = 0; i < 8; ++i) {
    int64_t idx = base + threadIdx.x + 256 * i;
    if (idx < len) s += a[idx];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(
    int32_t *__restrict__ bsum, hsgk_segkm_meta *meta, int which, int64_t BK) {
  __shared__ int32_t ws[4];
  __shared__ int32_t carry;
  const int64_t len = scan_len(meta, which, BK);
  const int nblk = (int)((len + kScanBlock - 1) / kScanBlock);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nblk; b0 += 256) {
    int i = b0 + tid;
    int v = i < nblk ? bsum[i] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) ws[w] = incl;
    __syncthreads();
    int base = carry;
    for (int k = 0; k < w; ++k) base += ws[k];
    if (i < nblk) bsum[i] = base + incl - v;
    __syncthreads();
    if (tid == 255) carry = base + incl;
    __syncthreads();
  }
  if (tid == 0 && !meta->error) {
    if (which == 0) { if (meta->relabel_mode == 1) meta->relabel_L = carry; }
    else meta->n_segments = carry;
  }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(
    int32_t *__restrict__ a, const int32_t *__restrict__ bsum,
    const hsgk_segkm_meta *__restrict__ meta, int which, int64_t BK) {
  __shared__ int32_t ws[4];
  const int64_t len = scan_len(meta, which, BK);
  const int64_t base = (int64_t)blockIdx.x * kScanBlock;
  if (base >= len) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // thread owns 8 consecutive elements
  int v[8];
  int s = 0;
  const int64_t i0 = base + (int64_t)tid * 8;
  for (int i = 0; i < 8; ++i) {
    v[i] = (i0 + i < len) ? a[i0 + i] : 0;
    s += v[i];
  }
  int incl = s;
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  int run = bsum[blockIdx.x] + incl - s;
  for (int k = 0; k < w; ++k) run += ws[k];
  for (int i = 0; i < 8; ++i) {
    if (i0 + i < len) a[i0 + i] = run;
    run += v[i];
  }
}

static int launch_scan(int32_t *a, int32_t *bsum, int64_t cap, hsgk_segkm_meta *meta,
                       int which, int64_t BK, hipStream_t s) {
  int nblk = (int)((cap + kScanBlock - 1) / kScanBlock);
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblk), dim3(256), 0, s, a, bsum, meta, which, BK);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, s, bsum, meta, which, BK);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblk), dim3(256), 0, s, a, bsum, meta, which, BK);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// ---- presence marks and final ids (one workgroup per chunk) ----------------
template <bool WRITE>
__global__ __launch_bounds__(256) void table_kernel(
    const int32_t *__restrict__ klab, const int64_t *__restrict__ labels,
    const int32_t *__restrict__ lrank, int32_t *__restrict__ table,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int K, int64_t batch_offset,
    int64_t *__restrict__ out_cluster, int64_t *__restrict__ out_batch,
    const hsgk_segkm_meta *__restrict__ meta) {
  const int c = blockIdx.x;
  if (meta->error || c >= meta->n_chunks) return;
  const int64_t row0 = chunk_row0[c];
  const int n = chunk_rows[c];
  const int64_t b = chunk_img[c];
  const int64_t L = meta->relabel_L;
  const bool ranked = meta->relabel_mode == 1;
  for (int r = threadIdx.x; r < n; r += 256) {
    const int64_t row = row0 + r;
    int64_t lab = labels[row];
    if (ranked) lab = lrank[lab];
    const int64_t key = (b * K + klab[row]) * L + lab;
    if (WRITE) {
      out_cluster[row] = table[key];
      out_batch[row] = b + batch_offset;
    } else {
      table[key] = 1;
    }
  }
}

int launch_relabel(const hsgk_segkm_args &a, const ChunkTable &t, int max_chunks,
                   const int32_t *klab, int32_t *table, int32_t *scan_tmp,
                   hipStream_t s) {
  const int64_t cap = a.table_cap;
  const int64_t BK = (int64_t)a.B * a.K;
  HSGK_REQUIRE(cap >= BK, "relabel table smaller than B*K");
  int32_t *lrank = table + cap;                 // second region of `cap` entries
  int32_t *bsum = scan_tmp;
  const bool has_labels = a.labels != nullptr;
  HSGK_CHECK_HIP(hipMemsetAsync(table, 0, (size_t)cap * 4 * (has_labels ? 2 : 1), s));
  hipLaunchKernelGGL(decide_kernel, dim3(1), dim3(1), 0, s, a.meta, BK, cap);
  HSGK_LAUNCH_CHECK();
  if (has_labels) {
    hipLaunchKernelGGL(mark_labels_kernel, dim3(1024), dim3(256), 0, s, a.out_labels, lrank,
                       a.meta);
    HSGK_LAUNCH_CHECK();
    if (int rc = launch_scan(lrank, bsum, cap, a.meta, 0, BK, s)) return rc;
    hipLaunchKernelGGL(decide2_kernel, dim3(1), dim3(1), 0, s, a.meta, BK, cap);
    HSGK_LAUNCH_CHECK();
  }
  if (max_chunks > 0) {
    hipLaunchKernelGGL(table_kernel<false>, dim3(max_chunks), dim3(256), 0, s, klab,
                       a.out_labels, lrank, table, t.chunk_row0, t.chunk_rows, t.chunk_img,
                       a.K, a.batch_offset, a.out_cluster, a.out_batch, a.meta);
    HSGK_LAUNCH_CHECK();
  }
  if (int rc = launch_scan(table, bsum, cap, a.meta, 1, BK, s)) return rc;
  if (max_chunks > 0) {
    hipLaunchKernelGGL(table_kernel<true>, dim3(max_chunks), dim3(256), 0, s, klab,
                       a.out_labels, lrank, table, t.chunk_row0, t.chunk_rows, t.chunk_img,
                       a.K, a.batch_offset, a.out_cluster, a.out_batch, a.meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

// ---- label width conversions for the stand-alone entry points --------------
__global__ void i64_to_i32_kernel(const int64_t *__restrict__ in, int64_t n,
                                  int32_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)in[i];
}
__global__ void i32_to_i64_kernel(const int32_t *__restrict__ in, int64_t n,
                                  int64_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}
static int conv_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
int launch_i64_to_i32(const int64_t *in, int64_t n, int32_t *out, hipStream_t s) {
  hipLaunchKernelGGL(i64_to_i32_kernel, dim3(conv_grid(n)), dim3(256), 0, s, in, n, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}
int launch_i32_to_i64(const int32_t *in, int64_t n, int64_t *out, hipStream_t s) {
  hipLaunchKernelGGL(i32_to_i64_kernel, dim3(conv_grid(n)), dim3(256), 0, s, in, n, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // namespace hsgk
