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

// grid (ceil(N / RPW), B): RPW rows per workgroup (4 when the whole launch has few rows -- 256 nodes of two images are
// 128 workgroups instead of 32 -- else 16), a wave per row
__global__ __launch_bounds__(256) void knn_graph_kernel(
    const float *__restrict__ A, int N, const uint8_t *__restrict__ pad,
    const int64_t *__restrict__ seg, int knn, int remove_self_loop, int binarize,
    float *__restrict__ out, int rpw) {
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
  const int i_end = min(N, ((int)blockIdx.x + 1) * rpw);
  for (int i = blockIdx.x * rpw + w; i < i_end; i += 4) {                 // wave per row
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


// =============================================================================
// DMon pooling objective (hsg/utils/graph/loss.py:27-96) for an adjacency WITHOUT gradient
// =============================================================================
// Per image, with S [N][K] (the masked assignments), d = A 1, 2m = 2 sum(d):
//   t = (Tr(S^T A S) - |S^T d|^2 / 2m) / 2m        (the modularity term; the reference forms d d^T [N][N] and
//                                                    two more batched GEMMs for the second trace -- it is the
//                                                    squared norm of the K-vector S^T d)
//   c = |sum_i S_i|_2                              (the collapse term before its constant factor)
// One workgroup per image: a wave per row of A (coalesced), the row's products with the K columns of S (LDS)
// and its degree reduced across the wave; per-row results kept for the backward:
//   dt/dS = ((A + A^T) S - 2 d (S^T d)^T / 2m) / 2m,   dc/dS[i][k] = colsum_k / c.
constexpr int kDmonMaxK = 32;

struct DmonStats { float tr1, q, two_m, c; float v[kDmonMaxK]; float colsum[kDmonMaxK]; };   // per image

// KT: K rounded up to 4 / 8 / 16 / 32 (the LDS copy of S is zero padded).  Forward: 16 lanes per row of A, one
// float4 per lane and step, S transposed in LDS so that a lane reads the four assignments of its columns as one
// float4 per cluster; 16 rows in flight per workgroup.
template <int KT>
__global__ __launch_bounds__(256) void dmon_pool_fwd_kernel(const float *__restrict__ adj, const float *__restrict__ s,
                                                            const uint8_t *__restrict__ valid, int N, int K,
                                                            float *__restrict__ as_out, float *__restrict__ deg_out,
                                                            DmonStats *__restrict__ stats, float *__restrict__ t_out,
                                                            float *__restrict__ c_out) {
  extern __shared__ __attribute__((aligned(16))) float dm_lds[];
  const int NP = (N + 3) & ~3;               // padded node count (float4 columns)
  float *st = dm_lds;                        // [KT][NP] masked assignments, transposed
  float *red = dm_lds + (size_t)KT * NP;     // [16 row slots][2 KT + 2]
  const int b = blockIdx.x, tid = threadIdx.x, sub = tid & 15, slot = tid >> 4;
  const float *A = adj + (size_t)b * N * N;
  const float *S = s + (size_t)b * N * K;
  for (int i = tid; i < KT * NP; i += 256) {
    const int k = i / NP, j = i - k * NP;
    st[i] = (k < K && j < N && (valid == nullptr || valid[(size_t)b * N + j])) ? S[(size_t)j * K + k] : 0.0f;
  }
  __syncthreads();
  const bool vec = (N & 3) == 0;
  float tr1 = 0.f, m = 0.f, v[KT], cs[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) v[k] = cs[k] = 0.f;
  for (int i = slot; i < N; i += 16) {
    float acc[KT], deg = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    for (int j = 4 * sub; j < N; j += 64) {
      float4 a;
      if (vec) {
        a = *reinterpret_cast<const float4 *>(A + (size_t)i * N + j);
      } else {
        a.x = A[(size_t)i * N + j];
        a.y = j + 1 < N ? A[(size_t)i * N + j + 1] : 0.f;
        a.z = j + 2 < N ? A[(size_t)i * N + j + 2] : 0.f;
        a.w = j + 3 < N ? A[(size_t)i * N + j + 3] : 0.f;
      }
      deg += (a.x + a.y) + (a.z + a.w);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const float4 sv = *reinterpret_cast<const float4 *>(st + k * NP + j);
        acc[k] = fmaf(a.x, sv.x, fmaf(a.y, sv.y, fmaf(a.z, sv.z, fmaf(a.w, sv.w, acc[k]))));
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      deg += __shfl_xor(deg, o);
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[k] += __shfl_xor(acc[k], o);
    }
    if (sub == 0) deg_out[(size_t)b * N + i] = deg;
    m += deg;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float sik = st[k * NP + i];
      if (sub == 0 && k < K) as_out[((size_t)b * N + i) * K + k] = acc[k];
      tr1 = fmaf(sik, acc[k], tr1);
      v[k] = fmaf(sik, deg, v[k]);
      cs[k] += sik;
    }
  }
  // the 16 row slots' partials, summed in slot order by thread 0
  if (sub == 0) {
    float *r = red + slot * (2 * KT + 2);
    r[0] = tr1; r[1] = m;
#pragma unroll
    for (int k = 0; k < KT; ++k) { r[2 + k] = v[k]; r[2 + KT + k] = cs[k]; }
  }
  __syncthreads();
  if (tid == 0) {
    DmonStats so;
    float T = 0.f, M = 0.f;
    for (int ww = 0; ww < 16; ++ww) { T += red[ww * (2 * KT + 2)]; M += red[ww * (2 * KT + 2) + 1]; }
    float q = 0.f, c2 = 0.f;
    for (int k = 0; k < KT; ++k) {
      float vk = 0.f, ck = 0.f;
      for (int ww = 0; ww < 16; ++ww) { vk += red[ww * (2 * KT + 2) + 2 + k]; ck += red[ww * (2 * KT + 2) + 2 + KT + k]; }
      so.v[k] = vk; so.colsum[k] = ck;
      q = fmaf(vk, vk, q);
      c2 = fmaf(ck, ck, c2);
    }
    so.tr1 = T; so.q = q; so.two_m = 2.0f * M; so.c = sqrtf(c2);
    stats[b] = so;
    t_out[b] = (T - q / so.two_m) / so.two_m;
    c_out[b] = so.c;
  }
}

// grad_s[b][i][k] = g_t[b] * ((A S + A^T S)[i][k] - 2 d_i v_k / 2m) / 2m + g_c[b] * colsum_k / c   (masked rows: 0)
// thread i walks column i of A (coalesced across threads), eight rows in flight
template <int KT>
__global__ __launch_bounds__(256) void dmon_pool_bwd_kernel(const float *__restrict__ adj, const float *__restrict__ s,
                                                            const uint8_t *__restrict__ valid, int N, int K,
                                                            const float *__restrict__ as_in, const float *__restrict__ deg_in,
                                                            const DmonStats *__restrict__ stats, const float *__restrict__ g_t,
                                                            const float *__restrict__ g_c, float *__restrict__ grad_s) {
  extern __shared__ __attribute__((aligned(16))) float dm_lds[];
  float *sm = dm_lds;                        // [N][KT]
  const int b = blockIdx.y, tid = threadIdx.x;
  const float *A = adj + (size_t)b * N * N;
  const float *S = s + (size_t)b * N * K;
  for (int i = tid; i < N * KT; i += 256) {
    const int j = i / KT, k = i - j * KT;
    sm[i] = (k < K && (valid == nullptr || valid[(size_t)b * N + j])) ? S[(size_t)j * K + k] : 0.0f;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  if (i >= N) return;
  float acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = 0.f;
  int j = 0;
  for (; j + 8 <= N; j += 8) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = A[(size_t)(j + u) * N + i];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int k = 0; k < KT; ++k) acc[k] = fmaf(a[u], sm[(j + u) * KT + k], acc[k]);
  }
  for (; j < N; ++j) {
    const float a = A[(size_t)j * N + i];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = fmaf(a, sm[j * KT + k], acc[k]);
  }
  const DmonStats so = stats[b];
  const float gt = g_t[b], gc = g_c[b];
  const float inv = 1.0f / so.two_m;
  const float deg = deg_in[(size_t)b * N + i];
  const bool live = valid == nullptr || valid[(size_t)b * N + i];
#pragma unroll
  for (int k = 0; k < KT; ++k)
    if (k < K) {
      const float dt = ((as_in[((size_t)b * N + i) * K + k] + acc[k]) - 2.0f * deg * so.v[k] * inv) * inv;
      const float dc = so.c > 0.0f ? so.colsum[k] / so.c : 0.0f;
      grad_s[((size_t)b * N + i) * K + k] = live ? gt * dt + gc * dc : 0.0f;
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
  const int rpw = (int64_t)B * N <= 16384 ? 4 : 16;
  hipLaunchKernelGGL(knn_graph_kernel, dim3((N + rpw - 1) / rpw, B), dim3(256), lds, s, A, N, padding_mask,
                     segment_labels, knn, remove_self_loop, binarize, out, rpw);
  HSGK_LAUNCH_CHECK();
  return 0;
}

size_t hsgk_dmon_pool_workspace_bytes(int B, int N, int K) {
  Carver cv(nullptr);
  cv.take<float>((size_t)(B > 0 ? B : 1) * N * K);
  cv.take<float>((size_t)(B > 0 ? B : 1) * N);
  cv.take<DmonStats>((size_t)(B > 0 ? B : 1));
  return cv.off + 256;
}

int hsgk_dmon_pool_fwd(const float *adj, const float *s, const uint8_t *valid, int B, int N, int K, float *t_out,
                       float *c_out, void *saved, size_t saved_bytes, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= kDmonMaxK, "bad shape (1 <= clusters <= 32)");
  HSGK_REQUIRE(adj && s && t_out && c_out && saved, "null argument");
  HSGK_REQUIRE(saved_bytes >= hsgk_dmon_pool_workspace_bytes(B, N, K), "saved-state buffer too small");
  const int KT = K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : 32;
  const size_t lds = ((size_t)KT * ((N + 3) & ~3) + 16 * (2 * KT + 2)) * sizeof(float);
  HSGK_REQUIRE(lds <= 64 * 1024, "too many nodes x clusters for the DMon pooling kernel");
  if (B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  Carver cv(saved);
  float *as = cv.take<float>((size_t)B * N * K);
  float *deg = cv.take<float>((size_t)B * N);
  DmonStats *stats = cv.take<DmonStats>((size_t)B);
#define HSGK_DMON_FWD(KTV)                                                                                   \
  hipLaunchKernelGGL(dmon_pool_fwd_kernel<KTV>, dim3(B), dim3(256), lds, st, adj, s, valid, N, K, as, deg, stats, t_out, c_out)
  if (KT == 4) HSGK_DMON_FWD(4);
  else if (KT == 8) HSGK_DMON_FWD(8);
  else if (KT == 16) HSGK_DMON_FWD(16);
  else HSGK_DMON_FWD(32);
#undef HSGK_DMON_FWD
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_dmon_pool_bwd(const float *adj, const float *s, const uint8_t *valid, int B, int N, int K, const void *saved,
                       const float *g_t, const float *g_c, float *grad_s, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && N >= 1 && K >= 1 && K <= kDmonMaxK, "bad shape (1 <= clusters <= 32)");
  HSGK_REQUIRE(adj && s && saved && g_t && g_c && grad_s, "null argument");
  if (B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  Carver cv(const_cast<void *>(saved));
  const float *as = cv.take<float>((size_t)B * N * K);
  const float *deg = cv.take<float>((size_t)B * N);
  const DmonStats *stats = cv.take<DmonStats>((size_t)B);
  const int KT = K <= 4 ? 4 : K <= 8 ? 8 : K <= 16 ? 16 : 32;
  const size_t lds = (size_t)N * KT * sizeof(float);
  HSGK_REQUIRE(lds <= 64 * 1024, "too many nodes x clusters for the DMon pooling kernel");
#define HSGK_DMON_BWD(KTV)                                                                                          \
  hipLaunchKernelGGL(dmon_pool_bwd_kernel<KTV>, dim3((N + 255) / 256, B), dim3(256), lds, st, adj, s, valid, N, K, as, deg, \
                     stats, g_t, g_c, grad_s)
  if (KT == 4) HSGK_DMON_BWD(4);
  else if (KT == 8) HSGK_DMON_BWD(8);
  else if (KT == 16) HSGK_DMON_BWD(16);
  else HSGK_DMON_BWD(32);
#undef HSGK_DMON_BWD
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
