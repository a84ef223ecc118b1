// topk.hip -- top-k prototype retrieval (reference hsg/utils/segsort/eval.py:9-52
// top_k_ranking; the same contraction feeds Segsort.predictions,
// hsg/models/predictions/segsort.py:66-123).  The reference builds the full
// [N,P] affinity matrix and argsorts every row; here the fp32-MFMA engine
// (score_tiles.h) streams the query rows against 64-prototype blocks and the
// epilogue keeps only each row's k best of the block; a second kernel merges the
// per-block candidates.  Order: descending score, lower prototype index first on
// exact ties (torch.argsort leaves ties unspecified).
#include "common.h"
#include "score_tiles.h"

namespace hsgk {

constexpr int kTopkMax = 32;

struct TopkEpi {
  int kb0, nrows, pb, topk;
  int64_t P, N, crow0;
  float *cval;                  // [npb][N][topk]
  int32_t *cidx;
  const int64_t *qgroup, *pgroup;    // optional: only prototypes of the query's group compete
  template <int MB>
  __device__ inline void operator()(int tile, const f32x16 (&acc)[MB]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    float v[MB][16];
    const int64_t qg = qgroup ? qgroup[crow0 + (valid ? px : 0)] : 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t p = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bool ok = p < P;
        if (ok && pgroup) ok = pgroup[p] == qg;
        v[m][r] = ok ? acc[m][r] : -INFINITY;
      }
    const int64_t base = ((int64_t)pb * N + crow0 + (valid ? px : 0)) * topk;
    for (int t = 0; t < topk; ++t) {
      float bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (v[m][r] > bv) { bv = v[m][r]; bi = k; }       // ascending k inside the lane
        }
      const float ov = __shfl_xor(bv, 32);
      const int oi = __shfl_xor(bi, 32);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (k == bi) v[m][r] = -INFINITY;                  // retire the winner
        }
      if (h == 0 && valid) {
        cval[base + t] = bv;
        cidx[base + t] = bi == 0x7fffffff ? -1 : bi;
      }
    }
  }
};

template <int KB, int NW, int KC, bool EVEN_D>
__global__ __launch_bounds__(NW * 64) void topk_tiles_kernel(
    const float *__restrict__ q, int c, const float *__restrict__ proto, int64_t P, int64_t N,
    int split, TopkEpi epi_proto) {
  constexpr int TPX = NW * 32;
  extern __shared__ float lds[];
  const int chunk = blockIdx.x / split, part = blockIdx.x - chunk * split;
  const int pb = blockIdx.y;
  const int tps = (HSGK_CHUNK / TPX + split - 1) / split;
  const int64_t c_row0 = (int64_t)chunk * HSGK_CHUNK;
  const int c_rows = (int)((N - c_row0) < HSGK_CHUNK ? (N - c_row0) : HSGK_CHUNK);
  const int nrows = min(c_rows - part * tps * TPX, tps * TPX);
  if (nrows <= 0) return;
  TopkEpi epi = epi_proto;
  epi.kb0 = pb * KB;
  epi.pb = pb;
  epi.nrows = nrows;
  epi.crow0 = c_row0 + (int64_t)part * tps * TPX;
  const int kvalid = (int)((P - (int64_t)pb * KB) < KB ? (P - (int64_t)pb * KB) : KB);
  score_tiles<KB, NW, KC, EVEN_D>(q, c, proto + (int64_t)pb * KB * c, kvalid, epi.crow0, nrows, lds, epi);
}

// one thread per row: k rounds of selection over the npb*topk block candidates
__global__ void topk_merge_kernel(const float *__restrict__ cval, const int32_t *__restrict__ cidx,
                                  int npb, int64_t N, int topk, int64_t *__restrict__ out_idx,
                                  float *__restrict__ out_val) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  int head[64];                                   // per-block cursor (lists are sorted)
  const int nb = npb < 64 ? npb : 64;
  for (int b = 0; b < nb; ++b) head[b] = 0;
  for (int t = 0; t < topk; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff, bb = -1;
    for (int b = 0; b < npb; ++b) {
      // blocks beyond the 64 cursors are scanned in full (rare: P > 4096)
      if (b < 64) {
        if (head[b] >= topk) continue;
        const int64_t o = ((int64_t)b * N + r) * topk + head[b];
        const float v = cval[o];
        const int i = cidx[o];
        if (i >= 0 && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; bb = b; }
      } else {
        for (int u = 0; u < topk; ++u) {
          const int64_t o = ((int64_t)b * N + r) * topk + u;
          const float v = cval[o];
          const int i = cidx[o];
          bool taken = false;
          for (int s = 0; s < t; ++s) taken |= out_idx[r * topk + s] == i;
          if (i >= 0 && !taken && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; bb = b; }
        }
      }
    }
    if (bb >= 0 && bb < 64) head[bb]++;
    out_idx[r * topk + t] = bi == 0x7fffffff ? 0 : bi;
    out_val[r * topk + t] = bv;
  }
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

size_t hsgk_topk_workspace_bytes(int64_t n, int c, int64_t P, int topk) {
  (void)c;
  const int64_t npb = (P + 63) / 64;
  return (size_t)(npb > 0 ? npb : 1) * (size_t)(n > 0 ? n : 1) * topk * 8 + 512;
}

int hsgk_topk_prototypes(const float *queries, int64_t n, int c, const float *proto, int64_t P,
                         int topk, int64_t *out_idx, float *out_val, void *workspace,
                         size_t workspace_bytes, hsgk_stream_t stream) {
  return hsgk_topk_prototypes_grouped(queries, n, c, proto, P, topk, nullptr, nullptr, out_idx, out_val,
                                      workspace, workspace_bytes, stream);
}

int hsgk_topk_prototypes_grouped(const float *queries, int64_t n, int c, const float *proto, int64_t P,
                                 int topk, const int64_t *query_group, const int64_t *proto_group,
                                 int64_t *out_idx, float *out_val, void *workspace,
                                 size_t workspace_bytes, hsgk_stream_t stream) {
  HSGK_REQUIRE((query_group == nullptr) == (proto_group == nullptr), "both group vectors or neither");
  HSGK_REQUIRE(n >= 0 && c >= 1 && P >= 1, "bad shape");
  HSGK_REQUIRE(topk >= 1 && topk <= kTopkMax && topk <= P, "top_k must be in [1, min(32, P)]");
  HSGK_REQUIRE(workspace_bytes >= hsgk_topk_workspace_bytes(n, c, P, topk), "workspace too small");
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  const int npb = (int)((P + 63) / 64);
  float *cval = static_cast<float *>(workspace);
  int32_t *cidx = reinterpret_cast<int32_t *>(cval + (size_t)npb * n * topk + 64);
  TopkEpi epi{0, 0, 0, topk, P, n, 0, cval, cidx, query_group, proto_group};
  const int nch = (int)((n + HSGK_CHUNK - 1) / HSGK_CHUNK);
  const bool even = (c & 1) == 0;
  auto go = [&](auto kern, size_t lds) -> int {
    int split = 1;
    while (split < HSGK_CHUNK / 256 && (int64_t)nch * npb * split < 1024) split *= 2;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nch * split, npb), dim3(512), lds, s, queries, c, proto, P, n, split, epi);
    HSGK_LAUNCH_CHECK();
    return 0;
  };
  const size_t l32 = score_tiles_lds_bytes<64, 8, 32>(c), l16 = score_tiles_lds_bytes<64, 8, 16>(c);
  int rc;
  if (l32 <= 160 * 1024)
    rc = even ? go(topk_tiles_kernel<64, 8, 32, true>, l32) : go(topk_tiles_kernel<64, 8, 32, false>, l32);
  else if (l16 <= 160 * 1024)
    rc = even ? go(topk_tiles_kernel<64, 8, 16, true>, l16) : go(topk_tiles_kernel<64, 8, 16, false>, l16);
  else {
    set_error("top-k: embedding dimension %d does not fit the LDS prototype block", c);
    return -1;
  }
  if (rc) return rc;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, cval, cidx,
                     npb, n, topk, out_idx, out_val);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
