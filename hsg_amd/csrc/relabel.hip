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
// Launch count matters here (a training-resolution call is ~0.45 ms in total): the tables are
// zeroed over their USED extent by the deciding kernels themselves (the table is sized for the
// worst case, 100 MB for 48 images x 64 clusters with labels), and each prefix sum is ONE
// launch (chained scan with decoupled look-back): 4 launches without labels, 7 with.
#include "common.h"

namespace hsgk {

constexpr int kScanBlock = 2048;   // elements per workgroup (256 threads x 8)

__device__ inline int64_t scan_len(const hsgk_segkm_meta *meta, int which, int64_t BK) {
  if (meta->error) return 0;
  if (which == 0) return meta->relabel_mode == 1 ? meta->label_max + 1 : 0;
  return BK * meta->relabel_L;
}

// state of the chained scans: [2][nblk] packed (flag << 32 | value) words, then 2 tickets
__host__ __device__ inline int64_t scan_blocks_for(int64_t cap) { return (cap + kScanBlock - 1) / kScanBlock + 1; }
__host__ __device__ inline size_t scan_state_words(int64_t cap) { return (size_t)2 * scan_blocks_for(cap) + 2; }

__device__ inline void zero_span(int32_t *p, int64_t n) {
  // grid-stride; 16 bytes per thread over the 16-byte aligned middle
  const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gn = (int64_t)gridDim.x * blockDim.x;
  int64_t head = (int64_t)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;
  if (head > n) head = n;
  const int64_t n4 = (n - head) >> 2;
  int4 *p4 = reinterpret_cast<int4 *>(p + head);
  for (int64_t i = gt; i < n4; i += gn) p4[i] = make_int4(0, 0, 0, 0);
  for (int64_t i = gt; i < head; i += gn) p[i] = 0;
  for (int64_t i = head + (n4 << 2) + gt; i < n; i += gn) p[i] = 0;
}

// decide the table layout (every thread from the same meta fields; block 0 records it) and zero
// what the chosen mode uses first: the whole presence table (direct keys) or the label-rank table
__global__ __launch_bounds__(256) void relabel_begin_kernel(hsgk_segkm_meta *meta, int64_t BK, int64_t cap,
                                                          int32_t *table, int32_t *lrank,
                                                          unsigned long long *state, size_t state_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < state_words; i += (size_t)gridDim.x * blockDim.x)
    state[i] = 0ull;
  if (meta->error) return;
  const int64_t L = meta->n_rows > 0 ? meta->label_max + 1 : 1;
  const bool direct = BK * L <= cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    meta->relabel_mode = direct ? 0 : 1;
    meta->relabel_L = direct ? L : 0;
    if (!direct && L > cap) meta->error = 2;
  }
  if (direct) zero_span(table, BK * L);
  else if (L <= cap) zero_span(lrank, L);
}

// ranked mode: relabel_L = number of distinct labels (left by the label scan); zero the table
__global__ __launch_bounds__(256) void relabel_ranked_kernel(hsgk_segkm_meta *meta, int64_t BK, int64_t cap,
                                                           int32_t *table) {
  if (meta->error || meta->relabel_mode != 1) return;
  const int64_t D = meta->relabel_L;
  if (BK * D > cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) meta->error = 2;
    return;
  }
  zero_span(table, BK * D);
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

// ---- exclusive scan over a device-sized int32 array, one launch ------------
// Chained scan with decoupled look-back: a workgroup takes a ticket (so every predecessor it
// may wait for is already running), publishes the sum of its 2048 elements, walks back over its
// predecessors' published words until it meets an inclusive prefix, publishes its own.  Length
// and total live on the device (meta); workgroups past the length leave at once.
__global__ __launch_bounds__(256) void scan_chained_kernel(
    int32_t *__restrict__ a, unsigned long long *__restrict__ st, unsigned long long *__restrict__ ticket,
    hsgk_segkm_meta *meta, int which, int64_t BK) {
  __shared__ int32_t ws[4];
  __shared__ int32_t sbid, sexcl;
  const int64_t len = scan_len(meta, which, BK);
  const int nblk = (int)((len + kScanBlock - 1) / kScanBlock);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  auto write_total = [&](int total) {
    if (meta->error) return;
    if (which == 0) { if (meta->relabel_mode == 1) meta->relabel_L = total; }
    else meta->n_segments = total;
  };
  if (nblk == 0) {
    if (blockIdx.x == 0 && tid == 0) write_total(0);
    return;
  }
  if ((int)blockIdx.x >= nblk) return;      // (the grid is sized for the table's capacity: no ticket, no traffic)
  if (tid == 0) sbid = (int)atomicAdd(ticket, 1ull);
  __syncthreads();
  const int bid = sbid;                     // exactly nblk tickets are drawn
  const int64_t i0 = (int64_t)bid * kScanBlock + (int64_t)tid * 8;    // thread owns 8 consecutive elements
  int v[8];
  int s = 0;
  for (int i = 0; i < 8; ++i) {
    v[i] = (i0 + i < len) ? a[i0 + i] : 0;
    s += v[i];
  }
  int incl = s;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) ws[w] = incl;
  __syncthreads();
  if (w == 0) {                               // wave 0 looks back, 64 predecessors per step
    const int total = ws[0] + ws[1] + ws[2] + ws[3];
    int excl = 0;
    if (bid > 0) {
      if (lane == 0)
        __hip_atomic_store(st + bid, (1ull << 32) | (unsigned int)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      for (int j0 = bid - 1; j0 >= 0; j0 -= 64) {
        const int j = j0 - lane;
        unsigned long long word = 2ull << 32;          // (lanes before block 0: an inclusive prefix of 0)
        for (;;) {
          if (j >= 0) word = __hip_atomic_load(st + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (__ballot((word >> 32) == 0ull) == 0ull) break;
          __builtin_amdgcn_s_sleep(1);
        }
        const unsigned long long full = __ballot((word >> 32) == 2ull);   // lanes holding an inclusive prefix
        const int stop = full ? __builtin_ctzll(full) : 63;                // nearest one (lane 0 = nearest block)
        int v = lane <= stop ? (int)(unsigned int)word : 0;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        excl += v;
        if (full) break;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(st + bid, (2ull << 32) | (unsigned int)(excl + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      sexcl = excl;
      if (bid == nblk - 1) write_total(excl + total);
    }
  }
  __syncthreads();
  int run = sexcl + incl - s;
  for (int k = 0; k < w; ++k) run += ws[k];
  for (int i = 0; i < 8; ++i) {
    if (i0 + i < len) a[i0 + i] = run;
    run += v[i];
  }
}

static int launch_scan(int32_t *a, unsigned long long *state, int64_t cap, hsgk_segkm_meta *meta,
                       int which, int64_t BK, hipStream_t s) {
  const int64_t nb = scan_blocks_for(cap);
  unsigned long long *st = state + (size_t)which * nb;
  unsigned long long *ticket = state + (size_t)2 * nb + which;
  hipLaunchKernelGGL(scan_chained_kernel, dim3((unsigned)nb), dim3(256), 0, s, a, st, ticket, meta, which, BK);
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
    const int64_t key = (b * K + get_label(klab, row)) * L + lab;      // (int32 or byte labels: common.h)
    if (WRITE) {
      out_cluster[row] = table[key];
      out_batch[row] = b + batch_offset;
    } else {
      table[key] = 1;
    }
  }
}

size_t relabel_scan_bytes(int64_t cap) { return scan_state_words(cap) * 8; }

int launch_relabel(const hsgk_segkm_args &a, const ChunkTable &t, int max_chunks,
                   const int32_t *klab, int32_t *table, int32_t *scan_tmp,
                   hipStream_t s) {
  const int64_t cap = a.table_cap;
  const int64_t BK = (int64_t)a.B * a.K;
  HSGK_REQUIRE(cap >= BK, "relabel table smaller than B*K");
  int32_t *lrank = table + cap;                 // second region of `cap` entries
  unsigned long long *state = reinterpret_cast<unsigned long long *>(scan_tmp);
  const bool has_labels = a.labels != nullptr;
  const int zgrid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (cap + 4095) / 4096));
  hipLaunchKernelGGL(relabel_begin_kernel, dim3(zgrid), dim3(256), 0, s, a.meta, BK, cap, table, lrank, state,
                     scan_state_words(cap));
  HSGK_LAUNCH_CHECK();
  if (has_labels) {
    hipLaunchKernelGGL(mark_labels_kernel, dim3(1024), dim3(256), 0, s, a.out_labels, lrank,
                       a.meta);
    HSGK_LAUNCH_CHECK();
    if (int rc = launch_scan(lrank, state, cap, a.meta, 0, BK, s)) return rc;
    hipLaunchKernelGGL(relabel_ranked_kernel, dim3(zgrid), dim3(256), 0, s, a.meta, BK, cap, table);
    HSGK_LAUNCH_CHECK();
  }
  if (max_chunks > 0) {
    hipLaunchKernelGGL(table_kernel<false>, dim3(max_chunks), dim3(256), 0, s, klab,
                       a.out_labels, lrank, table, t.chunk_row0, t.chunk_rows, t.chunk_img,
                       a.K, a.batch_offset, a.out_cluster, a.out_batch, a.meta);
    HSGK_LAUNCH_CHECK();
  }
  if (int rc = launch_scan(table, state, cap, a.meta, 1, BK, s)) return rc;
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
