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
__device__ inline int owner_wave(int label) {
  return (label ^ (label >> 2) ^ (label >> 4) ^ (label >> 6)) & 3;
}

template <int VEC, int UNROLL>
__global__ __launch_bounds__(256) void accumulate_kernel(
    const float *__restrict__ x, int d, const int32_t *__restrict__ klab,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    int K, int kb0, int kbn, float *__restrict__ partial,
    const hsgk_segkm_meta *__restrict__ meta) {
  typedef float gvec_t __attribute__((ext_vector_type(VEC), aligned(4)));       // global: dword aligned
  typedef float lvec_t __attribute__((ext_vector_type(VEC), aligned(4 * VEC))); // LDS: natural
  extern __shared__ float sums[];   // [kbn][DS]
  const int c = blockIdx.x;
  if (c >= meta->n_chunks) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int DS = (d + VEC - 1) / VEC * VEC;
  const int tot = kbn * DS;
  for (int i = tid; i < tot; i += 256) sums[i] = 0.0f;
  __syncthreads();

  const int64_t row0 = chunk_row0[c];
  const int n = chunk_rows[c];
  const int32_t *lab = klab + row0;
  const float *xr = x + row0 * d;
  constexpr int PW = 64 * VEC;
  const int npass = d / PW;
  const int tail0 = npass * PW;
  const int tail = d - tail0;          // < 64*VEC columns, one float per lane per step

  for (int base = 0; base < n; base += 64) {
    int l = -1;
    if (base + lane < n) l = lab[base + lane] - kb0;
    const bool mine = l >= 0 && l < kbn && owner_wave(l + kb0) == w;
    unsigned long long mask = __ballot(mine);
    while (mask) {
      int rr[UNROLL], ll[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        rr[u] = -1;
        ll[u] = 0;
        if (mask) {
          const int bit = __builtin_ctzll(mask);
          mask &= mask - 1;
          rr[u] = base + bit;
          ll[u] = __builtin_amdgcn_readlane(l, bit);
        }
      }
      // tail columns are requested first so they share the latency of the
      // wide row loads
      float tv[UNROLL];
      const bool on = lane < tail;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (rr[u] >= 0 && on) tv[u] = xr[(int64_t)rr[u] * d + tail0 + lane];
      for (int p = 0; p < npass; ++p) {
        gvec_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
          if (rr[u] >= 0)
            v[u] = *reinterpret_cast<const gvec_t *>(xr + (int64_t)rr[u] * d + p * PW + lane * VEC);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
          if (rr[u] >= 0) {
            lvec_t *dst = reinterpret_cast<lvec_t *>(sums + ll[u] * DS + p * PW + lane * VEC);
            lvec_t acc = *dst;
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = acc[i] + v[u][i];
            *dst = acc;
          }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
        if (rr[u] >= 0 && on) {
          float *dst = sums + ll[u] * DS + tail0 + lane;
          *dst = *dst + tv[u];
        }
      for (int t0 = 64; t0 < tail; t0 += 64) {       // wider tails (VEC > 1 only)
        if (t0 + lane < tail) {
#pragma unroll
          for (int u = 0; u < UNROLL; ++u)
            if (rr[u] >= 0) {
              float *dst = sums + ll[u] * DS + tail0 + t0 + lane;
              *dst = *dst + xr[(int64_t)rr[u] * d + tail0 + t0 + lane];
            }
        }
      }
    }
  }
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
  const size_t budget2 = 76 * 1024, budget1 = 150 * 1024;
  int kbn = K;
  if ((size_t)kbn * DS * 4 > budget2) {
    kbn = (int)(budget1 / ((size_t)DS * 4));
    if (kbn > K) kbn = K;
  }
  HSGK_REQUIRE(kbn >= 1, "row too long for the LDS segment table");
  // 16-byte global loads only need dword alignment on gfx950.
  auto kern = wide ? accumulate_kernel<4, 8> : accumulate_kernel<1, 16>;
  size_t lds = (size_t)kbn * DS * 4;
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
template <int KB, int NW, int KC, bool EVEN_D>
__global__ __launch_bounds__(NW * 64) void assign_kernel(
    const float *__restrict__ x, int d, const float *__restrict__ cent, int K, int kb0,
    const int64_t *__restrict__ chunk_row0, const int32_t *__restrict__ chunk_rows,
    const int32_t *__restrict__ chunk_img, int32_t *__restrict__ klab,
    float *__restrict__ best, int first_block, int split,
    const hsgk_segkm_meta *__restrict__ meta) {
  constexpr int NT = NW * 64;
  constexpr int TPX = NW * 32;
  constexpr int XS = KC + 1;
  constexpr int MB = KB / 32;
  constexpr int F2_PER_ROW = KC / 2;
  constexpr int LOADS = (TPX * F2_PER_ROW) / NT;
  static_assert((TPX * F2_PER_ROW) % NT == 0, "staging must divide evenly");

  extern __shared__ float lds[];
  // `split` workgroups share one chunk (split = 1 when there are plenty of
  // chunks; > 1 keeps all CUs busy on small batches); each takes a contiguous
  // range of the chunk's tiles and stages the centroid block once.
  const int c = blockIdx.x / split;
  if (c >= meta->n_chunks) return;
  const int part = blockIdx.x - c * split;
  const int tps = (HSGK_CHUNK / TPX + split - 1) / split;      // tiles per workgroup
  const int all_rows = chunk_rows[c];
  const int nrows = min(all_rows - part * tps * TPX, tps * TPX);
  if (nrows <= 0) return;
  const int64_t crow0 = chunk_row0[c] + (int64_t)part * tps * TPX;
  const int b = chunk_img[c];

  const int dpad = (d + 1) & ~1;
  const int DP = dpad | 1;
  float *cent_s = lds;                     // [KB][DP]
  float *xs = lds + KB * DP;               // [2][TPX][XS]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;

  // ---- stage the centroid block once per chunk (rows >= K and the pad
  //      column are zero); batches of 8 independent 8-byte loads per thread
  {
    const float *src = cent + ((int64_t)b * K + kb0) * d;
    const int kvalid = min(KB, K - kb0);
    if constexpr (EVEN_D) {
      const int half = d >> 1;
      const int total = KB * half;                     // float2 elements
      for (int f0 = 0; f0 < total; f0 += NT * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + tid + NT * u;
          const int k = f / half;
          v[u] = (f < total && k < kvalid) ? *reinterpret_cast<const float2 *>(src + 2 * (int64_t)f)
                                           : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int f = f0 + tid + NT * u;
          if (f < total) {
            const int k = f / half, col = 2 * (f - k * half);
            cent_s[k * DP + col] = v[u].x;
            cent_s[k * DP + col + 1] = v[u].y;
          }
        }
      }
    } else {
      for (int i = w; i < KB; i += NW)
        for (int dd = lane; dd < dpad; dd += 64)
          cent_s[i * DP + dd] = (i < kvalid && dd < d) ? src[(int64_t)i * d + dd] : 0.0f;
    }
  }

  const int nq = (dpad + KC - 1) / KC;
  const int ntile = (nrows + TPX - 1) / TPX;
  const int nsteps = ntile * nq;

  // Staging loads are branch-free: out-of-range rows / columns are clamped to
  // a valid address and zeroed by a select, so all LOADS loads of a chunk are
  // in flight together.
  float2 pre[LOADS];
  auto load_chunk = [&](int g) {
    const int tile = g / nq, q = g - tile * nq;
    const int n = min(TPX, nrows - tile * TPX);
    const int64_t row0 = crow0 + (int64_t)tile * TPX;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int e = tid + NT * i;
      const int px = e / F2_PER_ROW, f2 = e % F2_PER_ROW;
      const int dd = q * KC + 2 * f2;
      const bool ok = px < n && dd < d;
      const int pxc = px < n ? px : n - 1;
      const int ddc = dd < d ? dd : 0;
      const float *src = x + (row0 + pxc) * (int64_t)d + ddc;
      float2 v;
      if constexpr (EVEN_D) {
        v = *reinterpret_cast<const float2 *>(src);
      } else {
        v.x = src[0];
        v.y = src[ddc + 1 < d ? 1 : 0];
        if (dd + 1 >= d) v.y = 0.0f;
      }
      pre[i] = ok ? v : make_float2(0.0f, 0.0f);
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

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  int g = 0;
  for (int tile = 0; tile < ntile; ++tile) {
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;

    for (int q = 0; q < nq; ++q, ++g) {
      if (g + 1 < nsteps) load_chunk(g + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of, and its LDS
                                           // write-back behind, this chunk's MFMAs
      const float *xb = xs + (g & 1) * (TPX * XS) + (w * 32 + j) * XS + h;
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
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < nsteps) store_chunk((g + 1) & 1);
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
