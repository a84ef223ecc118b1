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

namespace hsgk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ===========================================================================
// M-step, stage 1: chunk partial sums.
// One workgroup per chunk.  Thread t owns columns t, t+256, ... of EVERY
// cluster row, so a given LDS word is only ever touched by one thread, in
// program (= row) order: ds_add_f32 gives the sequential sum of order C2
// without any cross-thread ordering question.
template <int UNROLL>
__global__ __launch_bounds__(256) void accumulate_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ klab,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    int K, int kb0, int kbn, float *__restrict__ partial,
    const hsgk_segkm_meta *__restrict__ meta) {
  extern __shared__ float sums[];   // [kbn][d]
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int tid = threadIdx.x;
  const int tot = kbn * d;
  for (int i = tid; i < tot; i += 256) sums[i] = 0.0f;
  __syncthreads();

  const int64_t row0 = chunk_row0[c];
  const int n = chunk_rows[c];
  const int32_t *lab = klab + row0;
  const float *xr = x + row0 * d;

  for (int dcol = tid; dcol < d; dcol += 256) {
    int r = 0;
    for (; r + UNROLL <= n; r += UNROLL) {
      float v[UNROLL];
      int l[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        l[u] = lab[r + u] - kb0;
        v[u] = xr[(int64_t)(r + u) * d + dcol];
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (l[u] >= 0 && l[u] < kbn)
          __hip_atomic_fetch_add(&sums[l[u] * d + dcol], v[u], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    for (; r < n; ++r) {
      int l = lab[r] - kb0;
      float v = xr[(int64_t)r * d + dcol];
      if (l >= 0 && l < kbn)
        __hip_atomic_fetch_add(&sums[l * d + dcol], v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float *out = partial + ((int64_t)c * K + kb0) * d;
  for (int i = tid; i < tot; i += 256) out[i] = sums[i];
}

int launch_accumulate(const float *x, int d, const int32_t *klab, const ChunkTable &t,
                      int max_chunks, int K, float *partial,
                      const hsgk_segkm_meta *meta, hipStream_t s) {
  if (max_chunks <= 0) return 0;
  // cluster rows per pass so that the table fits LDS (two workgroups per CU
  // when possible: <= 72 KiB each).
  const size_t budget2 = 72 * 1024, budget1 = 150 * 1024;
  int kbn = K;
  if ((size_t)kbn * d * 4 > budget2) {
    kbn = (int)(budget1 / ((size_t)d * 4));
    if (kbn > K) kbn = K;
    if ((size_t)K * d * 4 <= budget1) kbn = K;
  }
  HSGK_REQUIRE(kbn >= 1, "row too long for the LDS segment table");
  auto kern = accumulate_kernel<16>;
  size_t lds = (size_t)kbn * d * 4;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(150 * 1024)));
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
// E-step: fp32 MFMA assign kernel.
//
// Workgroup = NW waves; tile = NW*32 rows.  Centroid block (KB rows, KB = 32
// or 64) is staged once per workgroup in LDS with an odd row stride (bank
// conflict free A-operand reads).  Pixel rows stream through LDS in column
// chunks of KC floats, double buffered, padded to KC+1 floats per row.
// Per wave: 32 rows on the MFMA N dimension, centroids on M, so every lane
// ends with all KB scores of ONE row split between lanes l and l^32: the
// argmax is in-register plus one cross-half exchange.
//
// v_mfma_f32_32x32x2_f32 operand map: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
// D[i][j] at lane (j, h = l>>5), register r <-> i = (r&3) + 8*(r>>2) + 4*h.
template <int KB, int NW, int KC>
__global__ __launch_bounds__(NW * 64) void assign_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K, int kb0,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int32_t *__restrict__ klab,
    float *__restrict__ best, int first_block,
    const hsgk_segkm_meta *__restrict__ meta) {
  constexpr int NT = NW * 64;
  constexpr int TPX = NW * 32;
  constexpr int XS = KC + 1;
  constexpr int MB = KB / 32;
  constexpr int TILES_PER_CHUNK = HSGK_CHUNK / TPX;
  constexpr int F2_PER_ROW = KC / 2;
  constexpr int LOADS = (TPX * F2_PER_ROW) / NT;
  static_assert((TPX * F2_PER_ROW) % NT == 0, "staging must divide evenly");

  extern __shared__ float lds[];
  const int c = blockIdx.x / TILES_PER_CHUNK;
  if (c >= meta->n_chunks) return;
  const int tt = blockIdx.x % TILES_PER_CHUNK;
  const int n = min(TPX, chunk_rows[c] - tt * TPX);
  if (n <= 0) return;
  const int64_t row0 = chunk_row0[c] + (int64_t)tt * TPX;
  const int b = chunk_img[c];

  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  float *cent_s = lds;                     // [KB][DP]
  float *xs = lds + KB * DP;               // [2][TPX][XS]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;

  // ---- stage the centroid block (rows >= K and the pad column are zero)
  for (int i = w; i < KB; i += NW) {
    const int k = kb0 + i;
    const float *src = cent + ((int64_t)b * K + k) * d;
    for (int dd = lane; dd < dpad; dd += 64)
      cent_s[i * DP + dd] = (k < K && dd < d) ? src[dd] : 0.0f;
  }

  const bool even_d = (d & 1) == 0;
  float2 pre[LOADS];
  auto load_chunk = [&](int q) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int e = tid + NT * i;
      const int px = e / F2_PER_ROW, f2 = e % F2_PER_ROW;
      const int dd = q * KC + 2 * f2;
      float2 v = make_float2(0.0f, 0.0f);
      if (px < n && dd < d) {
        const float *src = x + (row0 + px) * (int64_t)d + dd;
        if (even_d) {
          v = *reinterpret_cast<const float2 *>(src);
        } else {
          v.x = src[0];
          if (dd + 1 < d) v.y = src[1];
        }
      }
      pre[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float *dst = xs + buf * (TPX * XS);
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int e = tid + NT * i;
      const int px = e / F2_PER_ROW, f2 = e % F2_PER_ROW;
      dst[px * XS + 2 * f2] = pre[i].x;
      dst[px * XS + 2 * f2 + 1] = pre[i].y;
    }
  };

  f32x16 acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

  const int nq = (dpad + KC - 1) / KC;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int q = 0; q < nq; ++q) {
    if (q + 1 < nq) load_chunk(q + 1);
    const float *xb = xs + (q & 1) * (TPX * XS) + (w * 32 + j) * XS + h;
    const float *cb = cent_s + j * DP + q * KC + h;
    const int rem = dpad - q * KC;
    if (rem >= KC) {
#pragma unroll
      for (int st = 0; st < KC / 2; ++st) {
        const float bv = xb[2 * st];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const float av = cb[m * 32 * DP + 2 * st];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
        }
      }
    } else {
      for (int st = 0; st < rem / 2; ++st) {
        const float bv = xb[2 * st];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const float av = cb[m * 32 * DP + 2 * st];
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
        }
      }
    }
    if (q + 1 < nq) store_chunk((q + 1) & 1);
    __syncthreads();
  }

  // ---- argmax over this lane's rows (ascending index, strict >)
  float bv = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = acc[m][r];
      if (k < K && v > bv) { bv = v; bi = k; }
    }
  const float ov = __shfl_xor(bv, 32);
  const int oi = __shfl_xor(bi, 32);
  if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }

  const int px = w * 32 + j;
  if (h == 0 && px < n) {
    const int64_t row = row0 + px;
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

template <int KB, int NW, int KC>
static int launch_assign_cfg(const float *x, int d, const float *cent, int K,
                             const ChunkTable &t, int max_chunks, int32_t *klab,
                             float *best, const hsgk_segkm_meta *meta, hipStream_t s) {
  constexpr int TPX = NW * 32;
  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  size_t lds = ((size_t)KB * DP + (size_t)2 * TPX * (KC + 1)) * 4;
  auto kern = assign_kernel<KB, NW, KC>;
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  const int grid = max_chunks * (HSGK_CHUNK / TPX);
  const bool multi = K > KB;
  for (int kb0 = 0; kb0 < K; kb0 += KB) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, x, d, cent, K, kb0,
                       t.chunk_row0, t.chunk_rows, t.chunk_img, klab,
                       multi ? best : nullptr, kb0 == 0 ? 1 : 0, meta);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
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
