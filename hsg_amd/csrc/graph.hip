// graph.hip -- k-NN affinity graph of the DMon clustering loss
// (hsg/utils/graph/common.py:39-125 affinity_matrix_as_attention with
//  exp_inner_product_kernel :23-36).
//
// The reference builds A = exp(conc * X^T X) with an einsum, then loops in Python over
// images and over the segments of every image (mask, masked_select, topk, compare,
// masked_fill: ~10 tiny launches per (image, segment)).  Here: one launch for A, one
// launch (16 rows per workgroup, a wave per row) for padding / self-loop masking, the
// per-segment k-th-largest cut and the binarisation.
//
// k-th largest without sorting: an entry of row i in segment s survives iff fewer than
// k_s = min(#valid nodes of s, knn) entries of that row and segment are STRICTLY larger
// (equivalent to "not (A < kth_val)", ties with the k-th value survive).
#include "common.h"

namespace hsgk {

// A[b,i,j] = exp(conc * sum_c x[b,c,i] x[b,c,j])   (C1 chain over c; x is [B,C,N])
__global__ __launch_bounds__(256) void affinity_kernel(const float *__restrict__ x, int C, int N,
                                                       float conc, float *__restrict__ A) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const float *xb = x + (int64_t)b * C * N;
  float acc = 0.0f;
  for (int c = 0; c < C; ++c) acc = fmaf(xb[(int64_t)c * N + i], xb[(int64_t)c * N + j], acc);
  A[((int64_t)b * N + i) * N + j] = expf(acc * conc);
}

// grid (ceil(N / 16), B): 16 rows per workgroup, a wave per row (4 rows each)
__global__ __launch_bounds__(256) void knn_graph_kernel(
    const float *__restrict__ A, int N, const uint8_t *__restrict__ pad,
    const int64_t *__restrict__ seg, int knn, int remove_self_loop, int binarize,
    float *__restrict__ out) {
  extern __shared__ unsigned char lds_raw[];
  int64_t *sl = reinterpret_cast<int64_t *>(lds_raw);                     // [N] segment label
  int *sid = reinterpret_cast<int *>(sl + N);                             // [N] first node with the same label, -1 = padded
  int *kseg = sid + N;                                                    // [N] k of the node's segment
  float *rows = reinterpret_cast<float *>(kseg + N);                      // [4 waves][N] masked row
  __shared__ int nvalid;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) nvalid = 0;
  __syncthreads();
  int cnt = 0;
  for (int j = tid; j < N; j += 256) {
    const bool v = !(pad && pad[(int64_t)b * N + j]);
    sl[j] = seg ? seg[(int64_t)b * N + j] : 0;
    sid[j] = v ? 0 : -1;
    cnt += v;
  }
  atomicAdd(&nvalid, cnt);
  __syncthreads();
  const bool drop_self = remove_self_loop && nvalid > 1;
  int my_sid[4], my_k[4];                       // (N <= 1024: at most 4 nodes per thread)
  for (int t = 0; t < 4; ++t) {
    const int j = tid + 256 * t;
    my_sid[t] = -1;
    my_k[t] = 0;
    if (j < N && sid[j] >= 0) {
      int first = -1, m = 0;
      for (int o = 0; o < N; ++o)
        if (sid[o] >= 0 && sl[o] == sl[j]) { if (first < 0) first = o; ++m; }
      my_sid[t] = first;
      my_k[t] = m < knn ? m : knn;              // k_s = min(#valid nodes with the same label, knn)
    }
  }
  __syncthreads();
  for (int t = 0; t < 4; ++t) {
    const int j = tid + 256 * t;
    if (j < N) { sid[j] = my_sid[t]; kseg[j] = my_k[t]; }
  }
  __syncthreads();
  float *row = rows + w * N;
  const float *Ab = A + (int64_t)b * N * N;
  float *ob = out + (int64_t)b * N * N;
  const int i_end = min(N, ((int)blockIdx.x + 1) * 16);
  for (int i = blockIdx.x * 16 + w; i < i_end; i += 4) {                  // wave per row
    const bool vi = sid[i] >= 0;
    for (int j = lane; j < N; j += 64) {
      float v = Ab[(int64_t)i * N + j];
      if (!vi || sid[j] < 0) v = 0.0f;
      if (drop_self && i == j) v = 0.0f;
      row[j] = v;
    }
    // (wave-private LDS row: the writes above are visible to the reads below in order)
    for (int j = lane; j < N; j += 64) {
      float v = row[j];
      const int s = sid[j];
      if (knn > 0 && s >= 0) {
        int greater = 0;
#pragma unroll 8
        for (int o = 0; o < N; ++o) greater += (sid[o] == s && row[o] > v) ? 1 : 0;   // LDS broadcast reads
        if (greater >= kseg[j]) v = 0.0f;
      }
      ob[(int64_t)i * N + j] = binarize ? (v > 0.0f ? 1.0f : 0.0f) : v;
    }
  }
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

int hsgk_knn_affinity(const float *x, const float *affinity_in, int B, int C, int N,
                      float concentration, const uint8_t *padding_mask,
                      const int64_t *segment_labels, int knn, int remove_self_loop, int binarize,
                      float *affinity_tmp, float *out, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && N >= 1 && knn >= 0, "bad shape");
  HSGK_REQUIRE(out && (affinity_in || (x && affinity_tmp && C >= 1)), "null argument");
  if (B == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  const float *A = affinity_in;
  if (!A) {
    hipLaunchKernelGGL(affinity_kernel, dim3((N + 255) / 256, N, B), dim3(256), 0, s, x, C, N,
                       concentration, affinity_tmp);
    HSGK_LAUNCH_CHECK();
    A = affinity_tmp;
  }
  HSGK_REQUIRE(N <= 1024, "too many nodes for the k-NN graph kernel");
  const size_t lds = (size_t)N * 16 + (size_t)4 * N * 4 + 16;
  hipLaunchKernelGGL(knn_graph_kernel, dim3((N + 15) / 16, B), dim3(256), lds, s, A, N, padding_mask,
                     segment_labels, knn, remove_self_loop, binarize, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
