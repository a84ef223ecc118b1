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
// update_sums_kernel: one workgroup (8 waves) per 2048-row chunk; the chunk's changed
// rows are compacted into an LDS list, each is loaded once (16-byte row loads, 8 rows per
// wave in flight, double buffered), converted, and added to / subtracted from a
// [kbn][d] int64 LDS table with ds_add_u64 (measured 2.5x faster than a plain 64-bit
// LDS read-add-write and 20x faster than ds_add_f32, tools/probes/lds_atomics.hip); the
// touched table rows are flushed to the per-image table with global 64-bit atomics.
#include "common.h"

namespace hsgk {

__device__ inline long long to_fixed(float x) {
  const float v = x * 65536.0f;                      // exact
  const float hi = rintf(v);
  const float lo = rintf((v - hi) * 16777216.0f);    // exact remainder, exact scaling
  return (long long)(int)hi * 16777216ll + (long long)(int)lo;
}

template <int NW, int UNROLL>
__global__ __launch_bounds__(NW * 64) void update_sums_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ prev,
    const int32_t *__restrict__ cur, const int64_t *__restrict__ chunk_row0,
    const int32_t *__restrict__ chunk_rows, const int32_t *__restrict__ chunk_img, int K, int kb0,
    int kbn, unsigned long long *__restrict__ sumq, const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned long long *tab = reinterpret_cast<unsigned long long *>(lds_raw);          // [kbn][d]
  uint32_t *list = reinterpret_cast<uint32_t *>(tab + (size_t)kbn * d);               // [HSGK_CHUNK]
  int *wcount = reinterpret_cast<int *>(list + HSGK_CHUNK);                           // [NW + 1]
  unsigned char *touched = reinterpret_cast<unsigned char *>(wcount + NW + 1);        // [kbn]
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t row0 = chunk_row0[c];
  const int n = chunk_rows[c];
  // ---- changed rows of this chunk that touch the cluster window [kb0, kb0 + kbn)
  constexpr int PER = HSGK_CHUNK / (NW * 64);
  int pl[PER], cl[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = (w * PER + i) * 64 + lane;                   // wave-contiguous rows: ascending list
    const int rr = min(r, n - 1);
    pl[i] = prev[row0 + rr];
    cl[i] = cur[row0 + rr];
    if (r >= n) pl[i] = cl[i];                                 // past the end: unchanged
  }
  int cnt = 0;
  auto in_win = [&](int l) { return l >= kb0 && l < kb0 + kbn; };
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool ch = pl[i] != cl[i] && (in_win(pl[i]) || in_win(cl[i]));
    cnt += __popcll(__ballot(ch));
  }
  if (lane == 0) wcount[w] = cnt;
  // zero the table while the counts settle
  {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int tot2 = (kbn * d + 1) / 2;                          // (the list behind absorbs an odd tail)
    u64x2 *t2 = reinterpret_cast<u64x2 *>(tab);
    for (int i = tid; i < tot2; i += NW * 64) t2[i] = u64x2{0ull, 0ull};
    for (int i = tid; i < kbn; i += NW * 64) touched[i] = 0;
  }
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
      const bool ch = pl[i] != cl[i] && (in_win(pl[i]) || in_win(cl[i]));
      const unsigned long long m = __ballot(ch);
      if (ch) {
        // row (11 bits) | new label + 1 in the window, 0 = outside (10+1 bits) | old label + 1 (10+1 bits)
        const uint32_t nw = in_win(cl[i]) ? (uint32_t)(cl[i] - kb0 + 1) : 0u;
        const uint32_t od = in_win(pl[i]) ? (uint32_t)(pl[i] - kb0 + 1) : 0u;
        list[pos + __popcll(m & ((1ull << lane) - 1ull))] =
            ((uint32_t)((w * PER + i) * 64 + lane) << 21) | (nw << 10) | od;
        if (nw) touched[nw - 1] = 1;
        if (od) touched[od - 1] = 1;
      }
      pos += __popcll(m);
    }
  }
  __syncthreads();
  // ---- every wave takes entries w, w + NW, ...: load the row once, add / subtract
  typedef float gvec_t __attribute__((ext_vector_type(4), aligned(4)));
  const float *xr = x + row0 * d;
  const int nq = d / 4, tail0 = nq * 4;                         // quads, then d mod 4 scalar columns
  const int nmine = (total - w + NW - 1) / NW;
  auto entry = [&](int i) { return list[min(w + i * NW, total - 1)]; };
  auto issue = [&](int i0, gvec_t (&v)[UNROLL][2], float (&t)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int r = (int)(entry(i0 + u) >> 21);
      const float *src = xr + (int64_t)r * d;
      v[u][0] = *reinterpret_cast<const gvec_t *>(src + 4 * min(lane, nq - 1));
      v[u][1] = *reinterpret_cast<const gvec_t *>(src + 4 * min(lane + 64, nq - 1));
      t[u] = src[min(tail0 + lane, d - 1)];
    }
  };
  auto fold = [&](int i0, const gvec_t (&v)[UNROLL][2], const float (&t)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (i0 + u < nmine) {
        const uint32_t e = entry(i0 + u);
        const int nw = (int)((e >> 10) & 2047u), od = (int)(e & 1023u);
        long long q[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) q[h][j] = to_fixed(v[u][h][j]);
        const long long qt = to_fixed(t[u]);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const int lab = side == 0 ? nw : od;
          if (lab == 0) continue;
          unsigned long long *rowp = tab + (size_t)(lab - 1) * d;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int q4 = lane + 64 * h;
            if (q4 < nq) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                atomicAdd(rowp + 4 * q4 + j, (unsigned long long)(side ? -q[h][j] : q[h][j]));
            }
          }
          if (tail0 + lane < d) atomicAdd(rowp + tail0 + lane, (unsigned long long)(side ? -qt : qt));
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
  // ---- flush the touched table rows into the image's table
  unsigned long long *gq = sumq + ((int64_t)chunk_img[c] * K + kb0) * d;
  for (int k = w; k < kbn; k += NW)
    if (touched[k])
      for (int i = lane; i < d; i += 64) {
        const unsigned long long v = tab[(size_t)k * d + i];
        if (v) atomicAdd(gq + (size_t)k * d + i, v);
      }
}

// centroid row = normalise((float)sum * 2^-40): one rounding per element, then the C1 norm chain
__global__ __launch_bounds__(256) void finalize_fx_kernel(const long long *__restrict__ sumq, int d,
                                                          int K, float eps, float *__restrict__ cent) {
  extern __shared__ float row[];    // [d] + 1
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long long *src = sumq + ((int64_t)b * K + k) * d;
  for (int i = tid; i < d; i += 256) row[i] = (float)src[i] * 9.094947017729282e-13f;   // 2^-40
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

// rows of d <= 512 columns (two 16-byte loads per lane + one scalar tail) and a cluster
// window that fits the LDS table
bool sums_fx_eligible(int d) { return d >= 8 && d <= 515; }

int sums_fx_window(int d, int K) {
  const size_t budget = 150 * 1024 - (size_t)HSGK_CHUNK * 4 - 64 - 1024;
  int kbn = (int)(budget / ((size_t)d * 8));
  if (kbn > K) kbn = K;
  if (kbn > 1023) kbn = 1023;               // 10-bit label fields of the list
  return kbn;
}

int launch_update_sums(const float *x, int d, const int32_t *prev, const int32_t *cur,
                       const ChunkTable &t, int max_chunks, int K, long long *sumq,
                       const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  const int kbn = sums_fx_window(d, K);
  HSGK_REQUIRE(kbn >= 1, "row too long for the exact-sum table");
  constexpr int NW = 8;
  auto kern = update_sums_kernel<NW, 8>;
  const size_t lds = (size_t)kbn * d * 8 + (size_t)HSGK_CHUNK * 4 + (NW + 1) * 4 + 1024 + 32;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)(158 * 1024)));
  for (int kb0 = 0; kb0 < K; kb0 += kbn) {
    const int curk = K - kb0 < kbn ? K - kb0 : kbn;
    hipLaunchKernelGGL(kern, dim3(max_chunks), dim3(NW * 64), lds, s, x, d, prev, cur, t.chunk_row0,
                       t.chunk_rows, t.chunk_img, K, kb0, curk,
                       reinterpret_cast<unsigned long long *>(sumq), meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

int launch_finalize_fx(const long long *sumq, int d, int K, int B, float eps, float *cent, hipStream_t s) {
  if (B <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(finalize_fx_kernel, dim3(K, B), dim3(256), (size_t)(d + 1) * 4, s, sumq, d, K, eps,
                     cent);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // namespace hsgk
