// prep.hip -- front half of segment_by_kmeans (reference
// hsg/utils/segsort/common.py:306-365): NCHW -> normalised NHWC rows, append
// the two location channels and re-normalise, drop ignore-labelled pixels
// (stable compaction), and emit the grid-seed label of every kept pixel.
//
// Canonical arithmetic: every sum of squares is one fmaf
// chain in ascending channel order starting from +0.0f; sqrtf and '/' are the
// correctly rounded IEEE operations (hipcc default, no fast-math).
#include <algorithm>

#include "common.h"

namespace hsgk {

// --------------------------------------------------------------------------
// Row L2 normalisation of an [n,d] matrix (normalize_embedding,
// general/common.py:101-120); this entry point serves small tables (prototypes, centroid rows, tests), not the
// pixel stream.  One WAVE per row: the lanes fetch the row coalesced into a wave-private LDS window, lane 0 walks
// the C1 chain from there (a thread per row walked its own strided global loads: 22 us for sixteen rows of 128),
// all lanes divide and store coalesced.  Rows longer than the window: the thread-per-row form.
constexpr int kNormRowWindow = 4096;                 // floats per wave
__global__ __launch_bounds__(256) void normalize_rows_wave_kernel(const float *__restrict__ x, int64_t n, int d,
                                                                  float eps, float *__restrict__ out,
                                                                  float *__restrict__ norms) {
  extern __shared__ float nr_win[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + w;
  if (r >= n) return;
  float *row = nr_win + (size_t)w * d;
  const float *xr = x + r * d;
  for (int i = lane; i < d; i += 64) row[i] = xr[i];
  // (wave-private window: own writes are visible to own reads in order)
  float nrm = 0.0f;
  if (lane == 0) {
    float ss = 0.0f;
    for (int i = 0; i < d; ++i) ss = fmaf(row[i], row[i], ss);
    nrm = sqrtf(ss);
    if (!(nrm >= eps)) nrm = eps;
    if (norms) norms[r] = nrm;
  }
  nrm = __shfl(nrm, 0);
  float *yr = out + r * d;
  for (int i = lane; i < d; i += 64) yr[i] = row[i] / nrm;
}

__global__ void normalize_rows_kernel(const float *__restrict__ x, int64_t n, int d,
                                      float eps, float *__restrict__ out,
                                      float *__restrict__ norms) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float *xr = x + r * d;
  float ss = 0.0f;
  for (int i = 0; i < d; ++i) ss = fmaf(xr[i], xr[i], ss);
  float nrm = sqrtf(ss);
  if (!(nrm >= eps)) nrm = eps;
  if (norms) norms[r] = nrm;
  float *yr = out + r * d;
  for (int i = 0; i < d; ++i) yr[i] = xr[i] / nrm;
}

int launch_normalize_rows(const float *x, int64_t n, int d, float eps, float *out,
                          float *norms, hipStream_t s) {
  if (n <= 0) return 0;
  if (d <= kNormRowWindow && n <= (int64_t)1 << 22) {
    hipLaunchKernelGGL(normalize_rows_wave_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)4 * d * 4, s, x, n, d,
                       eps, out, norms);
    HSGK_LAUNCH_CHECK();
    return 0;
  }
  int64_t blocks = (n + 63) / 64;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)blocks), dim3(64), 0, s, x,
                     n, d, eps, out, norms);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------
// Per 64-pixel tile: number of kept pixels (+ min/max of kept labels).  A workgroup walks
// gridDim.x-strided groups of four tiles and issues ONE atomic min / max pair at the end (one
// pair per tile on the same two words serialises in L2: 3.4 ms at 48 x 448 x 448).
__global__ __launch_bounds__(256) void count_valid_kernel(const int64_t *__restrict__ labels, int64_t HW,
                                                          int ntiles, int has_ignore, int64_t ignore,
                                                          int32_t *__restrict__ tile_cnt,
                                                          hsgk_segkm_meta *meta) {
  __shared__ long long red[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.y;
  int64_t lo = INT64_MAX, hi = INT64_MIN;
  for (int t = blockIdx.x * 4 + w; t < ntiles; t += gridDim.x * 4) {
    const int64_t pix = (int64_t)t * kTilePix + lane;
    bool keep = false;
    int64_t lab = 0;
    if (pix < HW) {
      lab = labels[(int64_t)b * HW + pix];
      keep = !(has_ignore && lab == ignore);
    }
    const unsigned long long m = __ballot(keep);
    if (keep) { lo = lab < lo ? lab : lo; hi = lab > hi ? lab : hi; }
    if (lane == 0) tile_cnt[(int64_t)b * ntiles + t] = __popcll(m);
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (lane == 0) { red[0][w] = lo; red[1][w] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { lo = red[0][i] < lo ? red[0][i] : lo; hi = red[1][i] > hi ? red[1][i] : hi; }
    if (lo <= hi) {
      atomicMin(reinterpret_cast<long long *>(&meta->label_min), (long long)lo);
      atomicMax(reinterpret_cast<long long *>(&meta->label_max), (long long)hi);
    }
  }
}

__global__ void init_meta_kernel(hsgk_segkm_meta *meta, int has_labels) {
  meta->n_rows = 0;
  meta->n_segments = 0;
  meta->label_min = has_labels ? INT64_MAX : 0;
  meta->label_max = has_labels ? INT64_MIN : 0;
  meta->n_chunks = 0;
  meta->error = 0;
  meta->relabel_mode = 0;
  meta->relabel_L = 1;
}

int launch_count_valid(const int64_t *labels, int B, int64_t HW, int has_ignore,
                       int64_t ignore, int32_t *tile_cnt, hsgk_segkm_meta *meta,
                       hipStream_t s) {
  hipLaunchKernelGGL(init_meta_kernel, dim3(1), dim3(1), 0, s, meta, labels != nullptr);
  HSGK_LAUNCH_CHECK();
  if (!labels) return 0;
  int ntiles = (int)((HW + kTilePix - 1) / kTilePix);
  const int gx = (ntiles + 3) / 4;
  dim3 grid(gx < 32 ? gx : 32, B);
  hipLaunchKernelGGL(count_valid_kernel, grid, dim3(256), 0, s, labels, HW, ntiles,
                     has_ignore, ignore, tile_cnt, meta);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------
// Exclusive prefix of the tile counts inside each image (one WG per image).
__global__ void scan_tiles_kernel(const int32_t *__restrict__ tile_cnt, int ntiles,
                                  int32_t *__restrict__ tile_off,
                                  int64_t *__restrict__ img_cnt) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < ntiles; t0 += 256) {
    int t = t0 + tid;
    int v = t < ntiles ? tile_cnt[(int64_t)b * ntiles + t] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = carry;
    for (int i = 0; i < w; ++i) base += wsum[i];
    if (t < ntiles) tile_off[(int64_t)b * ntiles + t] = base + incl - v;
    __syncthreads();
    if (tid == 255) carry = base + incl;
    __syncthreads();
  }
  if (tid == 0) img_cnt[b] = carry;
}

// Image row offsets + chunk table.  img_cnt == nullptr -> every image keeps HW.
// Up to kTablesLds images: block-wide prefix sums (wave shuffles + LDS) and one thread per chunk
// with a binary search for its image -- the first version, a serial loop over the images in
// thread 0 followed by a serial loop over the images for the chunks, took 18 us for 48 images
// (4 % of a training-resolution call).  Larger batches keep the serial form.
constexpr int kTablesLds = 2048;
__global__ __launch_bounds__(256) void build_tables_kernel(const int64_t *__restrict__ img_cnt, int64_t HW,
                                                          int B, ChunkTable t, int max_chunks,
                                                          hsgk_segkm_meta *meta) {
  __shared__ int64_t srow[kTablesLds + 1];
  __shared__ int32_t sch[kTablesLds + 1];
  __shared__ int64_t wrow[4];
  __shared__ int32_t wch[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (B <= kTablesLds) {
    int64_t crow = 0;                         // running totals of the images before this batch of 256
    int32_t cch = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {
      const int b = b0 + tid;
      const int64_t c = b < B ? (img_cnt ? img_cnt[b] : HW) : 0;
      const int32_t n = (int32_t)((c + HSGK_CHUNK - 1) / HSGK_CHUNK);
      int64_t ir = c;
      int32_t ic = n;
      for (int off = 1; off < 64; off <<= 1) {
        const int64_t orow = __shfl_up(ir, off);
        const int32_t och = __shfl_up(ic, off);
        if (lane >= off) { ir += orow; ic += och; }
      }
      if (lane == 63) { wrow[w] = ir; wch[w] = ic; }
      __syncthreads();
      int64_t br = crow;
      int32_t bc = cch;
      for (int k = 0; k < w; ++k) { br += wrow[k]; bc += wch[k]; }
      if (b < B) { srow[b] = br + ir - c; sch[b] = bc + ic - n; }
      crow += wrow[0] + wrow[1] + wrow[2] + wrow[3];
      cch += wch[0] + wch[1] + wch[2] + wch[3];
      __syncthreads();
    }
    if (tid == 0) {
      srow[B] = crow;
      sch[B] = cch;
      meta->n_rows = crow;
      meta->n_chunks = cch;
      if (crow == 0) { meta->label_min = 0; meta->label_max = 0; }
      if (meta->label_min < 0) meta->error = 1;
    }
    __syncthreads();
    for (int b = tid; b <= B; b += 256) {
      t.img_row0[b] = srow[b];
      t.img_chunk0[b] = sch[b];
    }
    const int total = min(sch[B], max_chunks);
    for (int g = tid; g < total; g += 256) {
      int lo = 0, hi = B - 1;                 // last image whose first chunk is <= g (empty images share a start)
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (sch[mid] <= g) lo = mid; else hi = mid - 1;
      }
      while (sch[lo + 1] <= g) ++lo;          // (skip images without rows)
      const int64_t off = (int64_t)(g - sch[lo]) * HSGK_CHUNK;
      const int64_t left = srow[lo + 1] - srow[lo] - off;
      t.chunk_row0[g] = srow[lo] + off;
      t.chunk_rows[g] = (int32_t)(left < HSGK_CHUNK ? left : HSGK_CHUNK);
      t.chunk_img[g] = lo;
    }
    return;
  }
  if (threadIdx.x == 0) {
    int64_t row = 0;
    int32_t ch = 0;
    for (int b = 0; b < B; ++b) {
      int64_t c = img_cnt ? img_cnt[b] : HW;
      t.img_row0[b] = row;
      t.img_chunk0[b] = ch;
      row += c;
      ch += (int32_t)((c + HSGK_CHUNK - 1) / HSGK_CHUNK);
    }
    t.img_row0[B] = row;
    t.img_chunk0[B] = ch;
    meta->n_rows = row;
    meta->n_chunks = ch;
    if (row == 0) { meta->label_min = 0; meta->label_max = 0; }
    if (meta->label_min < 0) meta->error = 1;
  }
  __syncthreads();
  for (int b = 0; b < B; ++b) {
    const int64_t r0 = t.img_row0[b];
    const int64_t cnt = t.img_row0[b + 1] - r0;
    const int32_t c0 = t.img_chunk0[b];
    const int32_t nc = t.img_chunk0[b + 1] - c0;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
      if (c0 + c >= max_chunks) continue;
      int64_t off = (int64_t)c * HSGK_CHUNK;
      int64_t left = cnt - off;
      t.chunk_row0[c0 + c] = r0 + off;
      t.chunk_rows[c0 + c] = (int32_t)(left < HSGK_CHUNK ? left : HSGK_CHUNK);
      t.chunk_img[c0 + c] = b;
    }
  }
}

int launch_build_tables(const int32_t *tile_cnt, int B, int64_t HW, int ntiles,
                        int32_t *tile_off, ChunkTable t, int max_chunks,
                        hsgk_segkm_meta *meta, hipStream_t s) {
  // tile_off holds align_up(B*ntiles, 64) int32 prefixes followed by the B
  // int64 per-image kept-pixel counts (see the workspace carve in api.hip)
  int64_t *img_cnt = nullptr;
  if (tile_cnt) {
    img_cnt = reinterpret_cast<int64_t *>(tile_off + align_up((size_t)B * ntiles, 64));
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(B), dim3(256), 0, s, tile_cnt, ntiles,
                       tile_off, img_cnt);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(build_tables_kernel, dim3(1), dim3(256), 0, s, img_cnt, HW, B, t,
                     max_chunks, meta);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// Single "image" of n rows (stand-alone kmeans / assign entry points).
__global__ void flat_table_kernel(int64_t n, ChunkTable t, int max_chunks,
                                  hsgk_segkm_meta *meta) {
  const int32_t nc = (int32_t)((n + HSGK_CHUNK - 1) / HSGK_CHUNK);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    t.img_row0[0] = 0;
    t.img_row0[1] = n;
    t.img_chunk0[0] = 0;
    t.img_chunk0[1] = nc;
    meta->n_rows = n;
    meta->n_chunks = nc;
    meta->error = 0;
  }
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < nc && c < max_chunks;
       c += gridDim.x * blockDim.x) {
    int64_t off = (int64_t)c * HSGK_CHUNK;
    int64_t left = n - off;
    t.chunk_row0[c] = off;
    t.chunk_rows[c] = (int32_t)(left < HSGK_CHUNK ? left : HSGK_CHUNK);
    t.chunk_img[c] = 0;
  }
}

int launch_flat_table(int64_t n, ChunkTable t, int max_chunks, hsgk_segkm_meta *meta,
                      hipStream_t s) {
  int blocks = (max_chunks + 255) / 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(flat_table_kernel, dim3(blocks), dim3(256), 0, s, n, t, max_chunks,
                     meta);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------
// prep kernel: one workgroup (256 threads) per 64-pixel tile of one image.
//
// LDS holds the tile pixel-major, tile[j][c] at j*S + c with S = C|1 (odd
// stride: 32 consecutive pixels hit 32 distinct banks both when lanes walk
// pixels (load, chains) and when lanes walk channels (store)).
//
//   phase 1  wave w loads channels w, w+4, ... : lanes = pixels, 256 B per
//            wave-instruction straight out of the NCHW plane.
//   phase 2a thread j<64 runs the canonical sum-of-squares chain of pixel j.
//   phase 2b all threads: e = x / norm in place.
//   phase 2c thread j<64: second chain over (e_0..e_{C-1}, ly, lx).
//   phase 3  wave w writes rows of pixels w, w+4, ... : lanes = channels.
__global__ __launch_bounds__(256) void prep_kernel(
    const float *__restrict__ in, int C, int64_t HW, int ntiles,
    const float *__restrict__ loc, int64_t loc_sb, const int64_t *__restrict__ labels,
    int has_ignore, int64_t ignore, const int32_t *__restrict__ tile_off,
    const int64_t *__restrict__ img_row0, const int32_t *__restrict__ seed_map, int64_t seed_sb,
    float eps, float *__restrict__ emb, float *__restrict__ emb_loc,
    int64_t *__restrict__ labels_out, int32_t *__restrict__ klab,
    float *__restrict__ norms_out, int64_t *__restrict__ rowmap_out,
    _Float16 *__restrict__ xh, uint2 *__restrict__ xt, PrepM0 m0) {
  extern __shared__ float lds[];
  const int S = C | 1;
  float *tile = lds;                       // [64][S]
  float *nrm1 = lds + 64 * S;              // [64]
  float *nrm2 = nrm1 + 64;                 // [64]
  float *locv = nrm2 + 64;                 // [64][2]
  int64_t *rowi = reinterpret_cast<int64_t *>(locv + 128);  // [64] (8B aligned: see launcher)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x, b = blockIdx.y;
  const int64_t p0 = (int64_t)t * kTilePix;
  const int D = C + 2;

  if (w == 0) {
    const int64_t pix = p0 + lane;
    bool keep = false;
    int64_t lab = 0;
    if (pix < HW) {
      lab = labels ? labels[(int64_t)b * HW + pix] : 0;
      keep = !(has_ignore && lab == ignore);
    }
    unsigned long long m = __ballot(keep);
    int rank = __popcll(m & ((1ull << lane) - 1ull));
    int64_t base = img_row0[b] + (tile_off ? (int64_t)tile_off[(int64_t)b * ntiles + t] : p0);
    int64_t row = keep ? base + rank : -1;
    rowi[lane] = row;
    if (rowmap_out && pix < HW) rowmap_out[(int64_t)b * HW + pix] = row;
    if (keep) {
      labels_out[row] = lab;
      klab[row] = seed_map[(int64_t)b * seed_sb + pix];
      locv[2 * lane + 0] = loc[(int64_t)b * loc_sb + pix * 2 + 0];
      locv[2 * lane + 1] = loc[(int64_t)b * loc_sb + pix * 2 + 1];
    }
    if (lane == 0) nrm1[0] = m ? 1.0f : 0.0f;   // "tile has work" flag, overwritten in 2a
  }
  __syncthreads();
  if (nrm1[0] == 0.0f) return;
  __syncthreads();

  // phase 1
  {
    const int64_t pix = p0 + lane;
    const bool ok = pix < HW;
    const float *src = in + (int64_t)b * C * HW + pix;
    for (int c = w; c < C; c += 4) {
      float v = ok ? src[(int64_t)c * HW] : 0.0f;
      tile[lane * S + c] = v;
    }
  }
  __syncthreads();
  // phase 2a
  if (w == 0) {
    const float *r = tile + lane * S;
    float ss = 0.0f;
    for (int c = 0; c < C; ++c) ss = fmaf(r[c], r[c], ss);
    float n1 = sqrtf(ss);
    if (!(n1 >= eps)) n1 = eps;
    nrm1[lane] = n1;
  }
  __syncthreads();
  // phase 2b
  for (int idx = tid; idx < 64 * C; idx += 256) {
    int c = idx >> 6, j = idx & 63;
    tile[j * S + c] = tile[j * S + c] / nrm1[j];
  }
  __syncthreads();
  // phase 2c
  if (w == 0) {
    const float *r = tile + lane * S;
    float ss = 0.0f;
    for (int c = 0; c < C; ++c) ss = fmaf(r[c], r[c], ss);
    float ly = locv[2 * lane], lx = locv[2 * lane + 1];
    ss = fmaf(ly, ly, ss);
    ss = fmaf(lx, lx, ss);
    float n2 = sqrtf(ss);
    if (!(n2 >= eps)) n2 = eps;
    nrm2[lane] = n2;
  }
  __syncthreads();
  // phase 3
  for (int j = w; j < 64; j += 4) {
    const int64_t row = rowi[j];
    if (row < 0) continue;
    const float n2 = nrm2[j];
    if (norms_out && lane == 0) { norms_out[2 * row] = nrm1[j]; norms_out[2 * row + 1] = n2; }
    const float *r = tile + j * S;
    float *eo = emb + row * C;
    float *lo = emb_loc + row * D;
    for (int c = lane; c < C; c += 64) {
      float e = r[c];
      eo[c] = e;
      lo[c] = e / n2;
    }
    if (lane < 2) lo[C + lane] = locv[2 * j + lane] / n2;
  }
}

// --------------------------------------------------------------------------
// Fast path of the prep kernel for C % 64 == 0 (C = 64, 128, 256, 384, 512):
// same phases and the same canonical arithmetic, but every LDS access moves
// 16 bytes.  tile[j][c] lives at j*C + (((c>>2) ^ (j&15)) << 2) + (c&3): the
// 16-byte quad index is XOR-swizzled with the pixel index, which makes
// ds_write_b128 (lanes = pixels, 8-lane groups), ds_read_b128 (lanes = pixels,
// 16-lane groups) and ds_read_b128 (lanes = quads of one pixel) conflict free.
// Global traffic: 256 B per wave-load on the NCHW side (plane rows), 1 KiB
// (float4) per wave-store for `embeddings`, 512 B (float2) for
// `embeddings_with_loc` whose rows are only 8-byte aligned (D = C+2).
__global__ __launch_bounds__(256) void prep_fast_kernel(
    const float *__restrict__ in, int C, int64_t HW, int ntiles,
    const float *__restrict__ loc, int64_t loc_sb, const int64_t *__restrict__ labels,
    int has_ignore, int64_t ignore, const int32_t *__restrict__ tile_off,
    const int64_t *__restrict__ img_row0, const int32_t *__restrict__ seed_map, int64_t seed_sb,
    float eps, float *__restrict__ emb, float *__restrict__ emb_loc,
    int64_t *__restrict__ labels_out, int32_t *__restrict__ klab,
    float *__restrict__ norms_out, int64_t *__restrict__ rowmap_out,
    _Float16 *__restrict__ xh, uint2 *__restrict__ xt, PrepM0 m0) {
  extern __shared__ float lds[];
  float *tile = lds;                       // [64][C] swizzled
  float *nrm1 = lds + 64 * C;              // [64]
  float *nrm2 = nrm1 + 64;                 // [64]
  float *locv = nrm2 + 64;                 // [64][2]
  int64_t *rowi = reinterpret_cast<int64_t *>(locv + 128);

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x, b = blockIdx.y;
  const int64_t p0 = (int64_t)t * kTilePix;
  const int D = C + 2;
  const int NQ = C >> 2;

  if (w == 0) {
    const int64_t pix = p0 + lane;
    bool keep = false;
    int64_t lab = 0;
    if (pix < HW) {
      lab = labels ? labels[(int64_t)b * HW + pix] : 0;
      keep = !(has_ignore && lab == ignore);
    }
    unsigned long long m = __ballot(keep);
    int rank = __popcll(m & ((1ull << lane) - 1ull));
    int64_t base = img_row0[b] + (tile_off ? (int64_t)tile_off[(int64_t)b * ntiles + t] : p0);
    int64_t row = keep ? base + rank : -1;
    rowi[lane] = row;
    if (rowmap_out && pix < HW) rowmap_out[(int64_t)b * HW + pix] = row;
    if (keep) {
      labels_out[row] = lab;
      klab[row] = seed_map[(int64_t)b * seed_sb + pix];
      locv[2 * lane + 0] = loc[(int64_t)b * loc_sb + pix * 2 + 0];
      locv[2 * lane + 1] = loc[(int64_t)b * loc_sb + pix * 2 + 1];
    }
    if (lane == 0) nrm1[0] = m ? 1.0f : 0.0f;
  }
  __syncthreads();
  if (nrm1[0] == 0.0f) return;
  __syncthreads();

  const int sw = lane & 15;
  // phase 1: 4 channel planes -> one 16-byte LDS write per pixel
  {
    const int64_t pix = p0 + lane;
    const bool ok = pix < HW;
    const float *src = in + (int64_t)b * C * HW + (ok ? pix : HW - 1);
    for (int q = w; q < NQ; q += 4) {
      float4 v;
      v.x = src[(int64_t)(4 * q + 0) * HW];
      v.y = src[(int64_t)(4 * q + 1) * HW];
      v.z = src[(int64_t)(4 * q + 2) * HW];
      v.w = src[(int64_t)(4 * q + 3) * HW];
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(tile + lane * C + ((q ^ sw) << 2)) = v;
    }
  }
  __syncthreads();
  // phase 2a
  if (w == 0) {
    const float *r = tile + lane * C;
    float ss = 0.0f;
    for (int q = 0; q < NQ; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
      ss = fmaf(v.x, v.x, ss);
      ss = fmaf(v.y, v.y, ss);
      ss = fmaf(v.z, v.z, ss);
      ss = fmaf(v.w, v.w, ss);
    }
    float n1 = sqrtf(ss);
    if (!(n1 >= eps)) n1 = eps;
    nrm1[lane] = n1;
  }
  __syncthreads();
  // phase 2b
  {
    const float n1 = nrm1[lane];
    float *r = tile + lane * C;
    for (int q = w; q < NQ; q += 4) {
      float4 *pv = reinterpret_cast<float4 *>(r + ((q ^ sw) << 2));
      float4 v = *pv;
      v.x = v.x / n1; v.y = v.y / n1; v.z = v.z / n1; v.w = v.w / n1;
      *pv = v;
    }
  }
  __syncthreads();
  // phase 2c
  if (w == 0) {
    const float *r = tile + lane * C;
    float ss = 0.0f;
    for (int q = 0; q < NQ; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
      ss = fmaf(v.x, v.x, ss);
      ss = fmaf(v.y, v.y, ss);
      ss = fmaf(v.z, v.z, ss);
      ss = fmaf(v.w, v.w, ss);
    }
    const float ly = locv[2 * lane], lx = locv[2 * lane + 1];
    ss = fmaf(ly, ly, ss);
    ss = fmaf(lx, lx, ss);
    float n2 = sqrtf(ss);
    if (!(n2 >= eps)) n2 = eps;
    nrm2[lane] = n2;
  }
  __syncthreads();
  // phase 3
  for (int j = w; j < 64; j += 4) {
    const int64_t row = rowi[j];
    if (row < 0) continue;
    const float n2 = nrm2[j];
    if (norms_out && lane == 0) { norms_out[2 * row] = nrm1[j]; norms_out[2 * row + 1] = n2; }
    const float *r = tile + j * C;
    float *eo = emb + row * C;
    float *lo = emb_loc + row * D;
    // fp16 copy of the emb_loc row for the first E-step filter level: the C main
    // columns (C % 64 == 0 here) in xh[row][C], the two location columns packed in
    // xt[row] (layout: score_tiles_f16.h)
    _Float16 *ho = xh ? xh + row * C : nullptr;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int sj = j & 15;
    float e2 = 0.0f;                         // |row - fp16(row)|^2, this lane's columns
    for (int q = lane; q < NQ; q += 64) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sj) << 2));
      *reinterpret_cast<float4 *>(eo + 4 * q) = v;
      float2 a, c2;
      a.x = v.x / n2; a.y = v.y / n2; c2.x = v.z / n2; c2.y = v.w / n2;
      *reinterpret_cast<float2 *>(lo + 4 * q) = a;
      *reinterpret_cast<float2 *>(lo + 4 * q + 2) = c2;
      if (ho) {
        const h4 hv = {(_Float16)a.x, (_Float16)a.y, (_Float16)c2.x, (_Float16)c2.y};
        *reinterpret_cast<h4 *>(ho + 4 * q) = hv;
        const float e0 = a.x - (float)hv[0], e1 = a.y - (float)hv[1];       // exact residuals
        const float e2b = c2.x - (float)hv[2], e3 = c2.y - (float)hv[3];
        e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2); e2 = fmaf(e2b, e2b, e2); e2 = fmaf(e3, e3, e2);
      }
    }
    if (ho)
      for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if (lane == 0) {
      float2 lv;
      lv.x = locv[2 * j] / n2;
      lv.y = locv[2 * j + 1] / n2;
      *reinterpret_cast<float2 *>(lo + C) = lv;
      if (ho) {
        const h2 hv = {(_Float16)lv.x, (_Float16)lv.y};
        const float e0 = lv.x - (float)hv[0], e1 = lv.y - (float)hv[1];
        e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2);
        // measured rounding error of the copy of this row, inflated against the rounding
        // of this very sum (bound: score_tiles_f16.h)
        xt[row] = make_uint2(__builtin_bit_cast(uint32_t, hv), __float_as_uint(sqrtf(e2) * 1.0001f));
      }
    }
  }
}

// --------------------------------------------------------------------------
// 32-pixel variant of the fast path: one workgroup handles HALF of a 64-pixel
// compaction tile (blockIdx.x = 2 * tile + half).  38 KiB of LDS per workgroup
// instead of 66 -> four workgroups per CU, so the serial norm chains of one
// workgroup (wave 0 only) overlap the load / divide / store phases of three
// others.  PMC on the 64-pixel kernel showed waves 69 % waiting (barriers +
// memory) with only two workgroups per CU.  Same arithmetic, same outputs.
// The kernel is latency-bound (two workgroups per CU: +45 %; 15 % less VALU work: no
// change), so the plane loads are issued before wave 0's bookkeeping, and it also
// produces the first M-step of the Lloyd loop: the exact fixed-point sums of its rows
// under their seed-grid labels (PrepM0, common.h).
// Phase timers for tools/probes/prep_timing.py: make EXTRA=-DHSGK_PREP_TIMING.
constexpr int kPrep32TilePad = 72;      // floats after the [32][C] tile: room for its rows as [32][C + 2] + a 16-byte phase
#ifdef HSGK_PREP_TIMING
__device__ unsigned long long g_prep_ts[8];
#define HSGK_TS(i) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
    atomicAdd(&g_prep_ts[i], now_ - ts_); ts_ = now_; } } while (0)
#else
#define HSGK_TS(i) do { } while (0)
#endif
// HSGK_PREP_NT (build-time A/B, tools/probes/ab_prep_nt.sh): the three row streams as nontemporal stores
#ifdef HSGK_PREP_NT
__device__ __forceinline__ void nt_store(float4 *p, float4 v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p));
}
__device__ __forceinline__ void nt_store(float2 *p, float2 v) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(f2v{v.x, v.y}, reinterpret_cast<f2v *>(p));
}
template <class T> __device__ __forceinline__ void nt_store(T *p, T v) { __builtin_nontemporal_store(v, p); }
#define HSGK_ROW_STORE(p, v) nt_store(p, v)
#else
#define HSGK_ROW_STORE(p, v) (*(p) = (v))
#endif
// HSGK_PREP_MARKSTEIN (build-time A/B): x / n as q0 = x * r, q = fma(fma(-q0, n, x), r, q0) with r = 1 / n correctly
// rounded once per row (Markstein's correction step) instead of the ~11-instruction IEEE division sequence
#ifdef HSGK_PREP_MARKSTEIN
#define HSGK_DIV_SETUP(n, r) const float r = 1.0f / (n);
__device__ __forceinline__ float div_markstein(float x, float n, float r) {
  const float q0 = x * r;
  return fmaf(fmaf(-q0, n, x), r, q0);
}
#define HSGK_DIV(x, n, r) div_markstein(x, n, r)
#else
#define HSGK_DIV_SETUP(n, r)
#define HSGK_DIV(x, n, r) ((x) / (n))
#endif
// ---- round 6: the kernel's vector work (static audit: tools/probes/isa_audit.py, profiles/r06_prep_isa_audit.txt)
// x / n, correctly rounded, for the quads of a row that share the divisor n.  hipcc lowers `x / n` to the IEEE
// sequence  s = v_div_scale(n), t = v_div_scale(x), r0 = v_rcp(s), r = fma(fma(-s, r0, 1), r0, r0), q0 = t * r,
// q1 = fma(fma(-s, q0, t), r, q0), q = v_div_fmas(fma(-s, q1, t), r, q1), v_div_fixup(q, n, x): twelve vector
// instructions per element, two of them (scale + reciprocal + its refinement) functions of n alone.  v_div_scale
// returns its operand unchanged, v_div_fmas is a plain fma and v_div_fixup passes q through (ISA manual: they act
// only on zeros, infinities, NaNs, denormals and exponent differences beyond +-96 / a dividend below 2^-103),
// whenever  2^-60 <= n <= 2^20  and  2^-100 <= |x| < 2^30.  In that range the five instructions below ARE the
// compiler's sequence -- same operations, same operands, same bits -- with r computed once per row.  |x| <= n (1 +
// 2^-20) holds for every dividend here (n is the row's own norm, or the eps clamp above it), so the guard is n's
// range, once per row, and min |x| >= 2^-100 per quad; a wave with any lane outside (exact zeros, tiny values,
// NaN / Inf rows) takes the compiler's sequence for that quad.  DivRow::make costs 3 instructions per row.
struct DivRow {
  float n, r;
  bool ok;
  static __device__ __forceinline__ DivRow make(float n) {
    const float r0 = __builtin_amdgcn_rcpf(n);
    const float e = fmaf(-n, r0, 1.0f);
    return DivRow{n, fmaf(e, r0, r0), n >= 0x1p-60f && n <= 0x1p20f};
  }
  __device__ __forceinline__ float one(float x) const {
    const float q0 = x * r;
    const float q1 = fmaf(fmaf(-n, q0, x), r, q0);
    return fmaf(fmaf(-n, q1, x), r, q1);
  }
  __device__ __forceinline__ float4 quad(float4 v) const {
    const float m = fminf(fminf(fabsf(v.x), fabsf(v.y)), fminf(fabsf(v.z), fabsf(v.w)));
    if (__builtin_amdgcn_ballot_w64(!(ok && m >= 0x1p-100f)) == 0ull)
      return make_float4(one(v.x), one(v.y), one(v.z), one(v.w));
    return make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
  }
};
// sum over the 64 lanes in the data-parallel-primitive network (six v_add_f32 with a DPP operand instead of six
// LDS-crossbar shuffles with their address arithmetic): pairs, quads, 8 (half-row mirror), 16 (row mirror), then
// lane 15 -> next row, lane 31 -> rows 2 and 3; the total sits in lane 63.  Fixed order, used for the error word only.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define HSGK_DPP_ADD(ctrl, rmask)                                                                         \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  HSGK_DPP_ADD(0xB1, 0xf);     // quad_perm [1, 0, 3, 2]
  HSGK_DPP_ADD(0x4E, 0xf);     // quad_perm [2, 3, 0, 1]
  HSGK_DPP_ADD(0x141, 0xf);    // row_half_mirror
  HSGK_DPP_ADD(0x140, 0xf);    // row_mirror
  HSGK_DPP_ADD(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
  HSGK_DPP_ADD(0x143, 0xc);    // row_bcast:31 into rows 2 and 3
#undef HSGK_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// to_fixed(x) + 2^52 * 1.5 as the raw bits of the double (common.h: to_fixed): x * 2^40 is exact in fp32 for |x| <= 1
// (a power-of-two scale), the double addition rounds it to the nearest-even integer exactly as to_fixed's fma does;
// the caller subtracts the magic constant's bits once per flushed run instead of once per element
__device__ __forceinline__ long long to_fixed_biased(float x) {
  return __builtin_bit_cast(long long, (double)(x * 0x1p40f) + 6755399441055744.0);
}
constexpr long long kFixedBias = 0x4338000000000000ll;      // bits of 6755399441055744.0

// FLAT (round 6; C <= 256, one 16-byte quad per lane): phase 3 keeps a wave's eight rows in registers, and the
// emb_loc rows -- 1 032 bytes each, so that a lane's quad is only 8-byte aligned and used to leave as two half-density
// 8-byte stores per lane plus a tail store per row -- go back to LDS in their final layout (the half tile's rows are
// one contiguous byte range of the output) and leave as full 16-byte pieces, 1 KiB contiguous per wave instruction;
// the per-row tail work (location columns, xt word, norms) is done once for all 32 rows with lane = row instead of on
// lane 0 of 32 wave-wide passes.  The kernel runs at the speed of its memory side (tools/probes/prep_mem.hip:
// the same access pattern without any arithmetic takes the kernel's time), and this is what the memory side
// gains: 6.38 -> 6.02 ms next to the XCD-contiguous order (profiles/r06_prep_mem.txt).  Same arithmetic, same bits.
template <bool FLAT>
__global__ __launch_bounds__(256) void prep_fast32_kernel(
    const float *__restrict__ in, int C, int64_t HW, int ntiles,
    const float *__restrict__ loc, int64_t loc_sb, const int64_t *__restrict__ labels,
    int has_ignore, int64_t ignore, const int32_t *__restrict__ tile_off,
    const int64_t *__restrict__ img_row0, const int32_t *__restrict__ seed_map, int64_t seed_sb,
    float eps, float *__restrict__ emb, float *__restrict__ emb_loc,
    int64_t *__restrict__ labels_out, int32_t *__restrict__ klab,
    float *__restrict__ norms_out, int64_t *__restrict__ rowmap_out,
    _Float16 *__restrict__ xh, uint2 *__restrict__ xt, PrepM0 m0) {
  const bool m0on = m0.part != nullptr;
  _Float16 *const xhT = m0.tiles;
  const bool tmode = xhT != nullptr;          // (host: no compaction, HW % 32 == 0 -> this half tile is one 32-row block)
#ifdef HSGK_PREP_TIMING
  unsigned long long ts_ = __builtin_readcyclecounter();
#endif
  extern __shared__ float lds[];
  float *tile = lds;                       // [32][C] swizzled; FLAT: later the rows of emb_loc, [<= 32][C + 2] + phase
  float *nrm1 = lds + 32 * C + kPrep32TilePad;   // [32]
  float *nrm2 = nrm1 + 32;                 // [32]
  float *locv = nrm2 + 32;                 // [32][2]
  int64_t *rowi = reinterpret_cast<int64_t *>(locv + 64);   // [32]
  int *seedl = reinterpret_cast<int *>(rowi + 32);          // [32] seed label of the kept pixels (-1: dropped)
  int *m0l = seedl + 32;                                    // [2] the (at most) two labels with an LDS slot, [2] rows kept in this half
  float *e2s = reinterpret_cast<float *>(m0l + 4);          // [32] squared fp16 rounding error of the rows' main columns
  int64_t *hrow0 = reinterpret_cast<int64_t *>(e2s + 32);   // [1] output row of this half's first kept pixel
  unsigned long long *mtab = reinterpret_cast<unsigned long long *>(hrow0 + 1);   // [2][D] exact sums (fused first M-step)

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // Half tile of this workgroup.  Workgroup ids are dealt round-robin over the 8 XCDs; with ids taken as they come,
  // neighbouring half tiles land on different XCDs, i.e. different L2s: the 128-byte lines that two neighbours
  // share (rows of 1 032 bytes, the 8-byte words per row) reach HBM as two partial writes, and no XCD streams a
  // contiguous range.  Round 6: id x + 8 i -> half tile start(x) + i, start(x) = x q + min(x, r) for gridDim.x =
  // 8 q + r -- one contiguous eighth of the image per XCD.  The kernel's memory side alone: 6.70 -> 6.38 ms,
  // tools/probes/prep_mem.hip, profiles/r06_prep_mem.txt.
  unsigned bx = blockIdx.x;
  if (m0.xcd_order) {
    const unsigned q8 = gridDim.x >> 3, r8 = gridDim.x & 7u, x8 = bx & 7u;
    bx = x8 * q8 + (x8 < r8 ? x8 : r8) + (bx >> 3);
  }
  const int t = (int)(bx >> 1), sh = (int)(bx & 1u), b = blockIdx.y;
  const int64_t p0 = (int64_t)t * kTilePix;        // start of the 64-pixel compaction tile
  const int64_t q0 = p0 + 32 * sh;                 // start of this half
  const int D = C + 2;
  const int NQ = C >> 2;
  const int jl = lane & 31, sub = lane >> 5;
  const int sw = jl & 15;

  // phase 1a: the first batch of plane loads (8 quads = 32 loads per thread; all of the tile
  // for C <= 256) is issued BEFORE the bookkeeping of wave 0, whose dependent loads would
  // otherwise put a second memory latency in front of them (18 % of a workgroup's lifetime
  // by the phase timers, tools/probes/prep_timing.py).  Indices are clamped, not branched on.
  const bool pix_ok = q0 + jl < HW;
  // plane bases on the scalar unit, ONE 32-bit byte offset per lane (pixel + which of the wave's two quads): a load
  // is one instruction with no vector address arithmetic (was ~6 per load, two of them quarter-rate multiplies).
  // The host guarantees 20 * HW < 2^32.
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const char *img = reinterpret_cast<const char *>(in + (int64_t)b * C * HW);
  const unsigned boff = ((unsigned)(pix_ok ? q0 + jl : HW - 1) + (unsigned)sub * 4u * (unsigned)HW) * 4u;
  const int64_t plane = HW * 4;
  float4 v0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int qs = min(2 * ws + 8 * u, NQ - 2);      // (scalar; a clamped quad is loaded twice and never stored)
    const char *pl = img + (int64_t)(4 * qs) * plane;
    v0[u].x = *reinterpret_cast<const float *>(pl + boff);
    v0[u].y = *reinterpret_cast<const float *>(pl + plane + boff);
    v0[u].z = *reinterpret_cast<const float *>(pl + 2 * plane + boff);
    v0[u].w = *reinterpret_cast<const float *>(pl + 3 * plane + boff);
  }

  if (w == 0) {
    const int64_t pix = p0 + lane;
    bool keep = false;
    int64_t lab = 0;
    if (pix < HW) {
      lab = labels ? labels[(int64_t)b * HW + pix] : 0;
      keep = !(has_ignore && lab == ignore);
    }
    const unsigned long long m = __ballot(keep);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    const int64_t base = img_row0[b] + (tile_off ? (int64_t)tile_off[(int64_t)b * ntiles + t] : p0);
    const int64_t row = keep ? base + rank : -1;
    if ((lane >> 5) == sh) {
      const int jl = lane & 31;
      rowi[jl] = row;
      if (rowmap_out && pix < HW) rowmap_out[(int64_t)b * HW + pix] = row;
      if (keep) {
        labels_out[row] = lab;
        klab[row] = seed_map[(int64_t)b * seed_sb + pix];
        locv[2 * jl + 0] = loc[(int64_t)b * loc_sb + pix * 2 + 0];
        locv[2 * jl + 1] = loc[(int64_t)b * loc_sb + pix * 2 + 1];
      }
    }
    const unsigned long long mh = sh ? (m >> 32) : (m & 0xffffffffull);
    if (lane == 0) {
      nrm1[0] = mh ? 1.0f : 0.0f;
      m0l[2] = __popcll(mh);
      *hrow0 = base + (sh ? __popcll(m & 0xffffffffull) : 0);
    }
    if (m0on && (lane >> 5) == sh) seedl[lane & 31] = keep ? seed_map[(int64_t)b * seed_sb + pix] : -1;   // (fused first M-step)
  }
  __syncthreads();
  HSGK_TS(0);
  if (nrm1[0] == 0.0f) return;
  __syncthreads();
  HSGK_TS(1);

  // phase 1b: 4 channel planes -> one 16-byte LDS write per (pixel, quad)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = 2 * w + sub + 8 * u;
      if (q < NQ)
        *reinterpret_cast<float4 *>(tile + jl * C + ((q ^ sw) << 2)) =
            pix_ok ? v0[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // further batches (C > 256): all 32 loads of a batch are issued before its first LDS
    // write (a load-use loop here exposes one memory latency per quad)
    for (int qb = 2 * ws + 64; qb < NQ; qb += 64) {      // (scalar loop: NQ is even, both quads of a wave agree)
      const int q0b = qb + sub;
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int qs = min(qb + 8 * u, NQ - 2);
        const char *pl = img + (int64_t)(4 * qs) * plane;
        v[u].x = *reinterpret_cast<const float *>(pl + boff);
        v[u].y = *reinterpret_cast<const float *>(pl + plane + boff);
        v[u].z = *reinterpret_cast<const float *>(pl + 2 * plane + boff);
        v[u].w = *reinterpret_cast<const float *>(pl + 3 * plane + boff);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int q = q0b + 8 * u;
        if (q < NQ)
          *reinterpret_cast<float4 *>(tile + jl * C + ((q ^ sw) << 2)) =
              pix_ok ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  HSGK_TS(2);
  // phase 2a: the C1 chain of pixel jl (wave 0, lanes 0..31).  The chain is the workgroup's critical path -- its
  // other three waves wait at the barrier below -- so the wave asks for issue priority over the other workgroups'
  // waves on its SIMD for as long as it runs (m0.xcd_order bit 1).
  if (w == 0 && sub == 0) {
    if (m0.xcd_order & 2) __builtin_amdgcn_s_setprio(3);
    const float *r = tile + jl * C;
    float ss = 0.0f;
    for (int q = 0; q < NQ; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
      ss = fmaf(v.x, v.x, ss);
      ss = fmaf(v.y, v.y, ss);
      ss = fmaf(v.z, v.z, ss);
      ss = fmaf(v.w, v.w, ss);
    }
    float n1 = sqrtf(ss);
    if (!(n1 >= eps)) n1 = eps;
    nrm1[jl] = n1;
    if (m0.xcd_order & 2) __builtin_amdgcn_s_setprio(0);
  } else if (m0on && w == 1) {
    // fused first M-step, off the critical path (wave 0 walks the chains meanwhile): the
    // first two distinct seed labels of this half tile get an LDS slot
    const int sl = lane < 32 ? seedl[lane] : -1;
    const unsigned long long mk = __ballot(sl >= 0);
    int L0 = -1, L1 = -1;
    if (mk) {
      L0 = __builtin_amdgcn_readlane(sl, __builtin_ctzll(mk));
      const unsigned long long m1 = __ballot(sl >= 0 && sl != L0);
      if (m1) L1 = __builtin_amdgcn_readlane(sl, __builtin_ctzll(m1));
    }
    if (lane == 0) { m0l[0] = L0; m0l[1] = L1; }
  } else if (m0on && w >= 2) {
    for (int i = tid - 128; i < 2 * (C + 2); i += 128) mtab[i] = 0ull;
  }
  __syncthreads();
  HSGK_TS(3);
  // phase 2b
  {
    const float n1 = nrm1[jl];
    HSGK_DIV_SETUP(n1, r1)
    const DivRow d1 = DivRow::make(n1);
    float *r = tile + jl * C;
    for (int q = 2 * w + sub; q < NQ; q += 8) {
      float4 *pv = reinterpret_cast<float4 *>(r + ((q ^ sw) << 2));
      float4 v = *pv;
      if constexpr (FLAT) {
        v = d1.quad(v);
      } else {
        v.x = HSGK_DIV(v.x, n1, r1); v.y = HSGK_DIV(v.y, n1, r1); v.z = HSGK_DIV(v.z, n1, r1); v.w = HSGK_DIV(v.w, n1, r1);
      }
      *pv = v;
    }
  }
  __syncthreads();
  HSGK_TS(4);
  // FLAT: this wave's eight rows -> registers, and the `embeddings` rows (final since phase 2b) leave NOW, so that
  // 28 % of the workgroup's store traffic flies during the second chain instead of after it (m0.xcd_order bit 2)
  [[maybe_unused]] float4 rv[8];
  [[maybe_unused]] const bool early = FLAT && (m0.xcd_order & 4);
  if constexpr (FLAT) {
    const bool act = lane < NQ;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = w + 4 * i;
      rv[i] = *reinterpret_cast<const float4 *>(tile + j * C + (((act ? lane : 0) ^ (j & 15)) << 2));
    }
    if (early) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long long rr = rowi[w + 4 * i];
        const int64_t row = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rr >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)rr));
        if (row >= 0 && act) HSGK_ROW_STORE(reinterpret_cast<float4 *>(emb + row * C + 4 * lane), rv[i]);
      }
    }
  }
  // phase 2c
  if (w == 0 && sub == 0) {
    if (m0.xcd_order & 2) __builtin_amdgcn_s_setprio(3);
    const float *r = tile + jl * C;
    float ss = 0.0f;
    for (int q = 0; q < NQ; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
      ss = fmaf(v.x, v.x, ss);
      ss = fmaf(v.y, v.y, ss);
      ss = fmaf(v.z, v.z, ss);
      ss = fmaf(v.w, v.w, ss);
    }
    const float ly = locv[2 * jl], lx = locv[2 * jl + 1];
    ss = fmaf(ly, ly, ss);
    ss = fmaf(lx, lx, ss);
    float n2 = sqrtf(ss);
    if (!(n2 >= eps)) n2 = eps;
    nrm2[jl] = n2;
    if (m0.xcd_order & 2) __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();
  HSGK_TS(5);
  // phase 3
  // fused first M-step: the exact sums of the wave's current run of rows with one seed label
  // stay in registers (columns 4*lane.., lane 0: the two location columns too) and go to their
  // place -- LDS slot 0 / 1, or the global table for a third label -- when the label changes
  long long cur[4] = {0, 0, 0, 0}, tcur[2] = {0, 0};
  int cslot = -1;                            // 0 / 1: LDS slot, 2: global (cg), -1: nowhere
  unsigned long long *cg = nullptr;
  auto m0_flush = [&]() {
    if (cslot == 0 || cslot == 1) {
      unsigned long long *t = mtab + cslot * D;
      if (lane < NQ)
        for (int i = 0; i < 4; ++i) atomicAdd(t + 4 * lane + i, (unsigned long long)cur[i]);   // ds_add_u64
      if (lane == 0) { atomicAdd(t + C, (unsigned long long)tcur[0]); atomicAdd(t + C + 1, (unsigned long long)tcur[1]); }
    } else if (cslot == 2) {
      if (lane < NQ)
        for (int i = 0; i < 4; ++i) atomicAdd(cg + 4 * lane + i, (unsigned long long)cur[i]);
      if (lane == 0) { atomicAdd(cg + C, (unsigned long long)tcur[0]); atomicAdd(cg + C + 1, (unsigned long long)tcur[1]); }
    }
    cur[0] = cur[1] = cur[2] = cur[3] = 0; tcur[0] = tcur[1] = 0;
  };
  if constexpr (FLAT) {
    // ---- (NQ <= 64) this wave's eight rows -> registers; emb, the fp16 copy and the first M-step straight from them
    const bool act = lane < NQ;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    int nrun = 0;                              // rows in the current run: their to_fixed bias leaves with the flush
    auto flat_flush = [&]() {
      const long long bias = (long long)nrun * kFixedBias;
      cur[0] -= bias; cur[1] -= bias; cur[2] -= bias; cur[3] -= bias;
      nrun = 0;
      m0_flush();
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = w + 4 * i;
      const long long rr = rowi[j];           // (wave-uniform: the row's addresses on the scalar unit)
      const int64_t row = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rr >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)rr));
      if (row < 0) continue;
      const DivRow d2 = DivRow::make(nrm2[j]);
      if (m0on) {
        const int L = __builtin_amdgcn_readfirstlane(seedl[j]);
        const int L0 = __builtin_amdgcn_readfirstlane(m0l[0]), L1 = __builtin_amdgcn_readfirstlane(m0l[1]);
        const int slot = L < 0 ? -1 : L == L0 ? 0 : L == L1 ? 1 : L < m0.K ? 2 : -1;
        unsigned long long *g = slot == 2 ? m0.sumq + ((int64_t)b * m0.K + L) * D : nullptr;
        if (slot != cslot || g != cg) { flat_flush(); cslot = slot; cg = g; }
      }
      float e2 = 0.0f;                         // |row - fp16(row)|^2, this lane's columns
      if (act) {
        const float4 v = rv[i];
        if (!early) HSGK_ROW_STORE(reinterpret_cast<float4 *>(emb + row * C + 4 * lane), v);
        const float4 a = d2.quad(v);
        rv[i] = a;                             // (kept: the emb_loc row leaves through LDS below)
        if (m0on) {                            // (uniform; the bias of to_fixed_biased leaves at the flush)
          cur[0] += to_fixed_biased(a.x); cur[1] += to_fixed_biased(a.y);
          cur[2] += to_fixed_biased(a.z); cur[3] += to_fixed_biased(a.w);
          ++nrun;
        }
        if (xh || tmode) {
          const h4 hv = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w};
          if (xh) HSGK_ROW_STORE(reinterpret_cast<h4 *>(xh + row * C + 4 * lane), hv);
          // tile order: the four halves wait in the quad's own LDS slot (read above by this lane, by nobody else)
          if (tmode) *reinterpret_cast<h4 *>(tile + j * C + ((lane ^ (j & 15)) << 2)) = hv;
          const float e0 = a.x - (float)hv[0], e1 = a.y - (float)hv[1];       // exact residuals
          const float e2b = a.z - (float)hv[2], e3 = a.w - (float)hv[3];
          e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2); e2 = fmaf(e2b, e2b, e2); e2 = fmaf(e3, e3, e2);
        }
      }
      if (xh || tmode) {
        e2 = wave_sum_dpp(e2);
        if (lane == 0) e2s[j] = e2;
      }
    }
    if (m0on) flat_flush();
    if (tmode) {
      // the block's 16-byte pieces in operand order: piece (kb, lane = jj + 32 g) = row jj, columns 16 kb + 8 g .. + 7
      // (two quads of four halves each, from their LDS slots); one KiB contiguous per wave instruction
      __syncthreads();
      const int jj = lane & 31, gg = lane >> 5;
      const int64_t blk = (img_row0[b] + q0) >> 5;
      uint2 *dst = reinterpret_cast<uint2 *>(xhT + blk * 32 * C);
      for (int kb = w; kb < (C >> 4); kb += 4) {
        const int qa = 4 * kb + 2 * gg;
        const uint2 lo2 = *reinterpret_cast<const uint2 *>(tile + jj * C + ((qa ^ (jj & 15)) << 2));
        const uint2 hi2 = *reinterpret_cast<const uint2 *>(tile + jj * C + (((qa + 1) ^ (jj & 15)) << 2));
        uint4 pc = {lo2.x, lo2.y, hi2.x, hi2.y};
        *reinterpret_cast<uint4 *>(dst + (kb * 64 + lane) * 2) = pc;
      }
    }
    __syncthreads();                           // nobody reads the tile any more: it becomes the rows of emb_loc
    HSGK_TS(6);
    const int64_t h0 = *hrow0;
    const int hc = m0l[2];
    float *const gl = emb_loc + h0 * D;        // this half's rows: [gl, gl + hc * D), 8-byte aligned
    // same 16-byte phase in LDS as in HBM, so that aligned pieces of one are aligned pieces of the other
    float *const flat = tile + ((reinterpret_cast<uintptr_t>(gl) >> 2) & 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = w + 4 * i;
      const long long rr = rowi[j];
      const int64_t row = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rr >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)rr));
      if (row < 0) continue;
      if (act) {
        float *fr = flat + (int)(row - h0) * D + 4 * lane;
        *reinterpret_cast<float2 *>(fr) = make_float2(rv[i].x, rv[i].y);
        *reinterpret_cast<float2 *>(fr + 2) = make_float2(rv[i].z, rv[i].w);
      }
    }
    // the per-row tail, lane = row: location columns, their share of the first M-step, norms, the xt word
    if (w == 3 && lane < 32) {
      const int64_t row = rowi[lane];
      if (row >= 0) {
        const float n2 = nrm2[lane];
        float2 lv;
        lv.x = locv[2 * lane] / n2;
        lv.y = locv[2 * lane + 1] / n2;
        *reinterpret_cast<float2 *>(flat + (int)(row - h0) * D + C) = lv;
        if (norms_out) { norms_out[2 * row] = nrm1[lane]; norms_out[2 * row + 1] = n2; }
        if (m0on) {
          const int L = seedl[lane];
          unsigned long long *t2 = L < 0 ? nullptr : L == m0l[0] ? mtab : L == m0l[1] ? mtab + D
                                   : L < m0.K ? m0.sumq + ((int64_t)b * m0.K + L) * D : nullptr;
          if (t2) {                            // (exact integer sums: any order; LDS or global atomic by address space)
            atomicAdd(t2 + C, (unsigned long long)to_fixed(lv.x));
            atomicAdd(t2 + C + 1, (unsigned long long)to_fixed(lv.y));
          }
        }
        if (xh || tmode) {
          const h2 hv = {(_Float16)lv.x, (_Float16)lv.y};
          const float e0 = lv.x - (float)hv[0], e1 = lv.y - (float)hv[1];
          float e2 = e2s[lane];
          e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2);
          // measured rounding error of the copy of this row, inflated against the rounding
          // of this very sum (bound: score_tiles_f16.h)
          xt[row] = make_uint2(__builtin_bit_cast(uint32_t, hv), __float_as_uint(sqrtf(e2) * 1.0001f));
        }
      }
    }
    __syncthreads();
    {
      // [gl, gl + hc * D) as 16-byte pieces (an 8-byte head / tail where the range starts / ends mid-piece)
      char *gb = reinterpret_cast<char *>(gl);
      const char *lb = reinterpret_cast<const char *>(flat);
      const int total = hc * D * 4;
      const int head = (int)((16 - (reinterpret_cast<uintptr_t>(gb) & 15)) & 15);     // 0 or 8
      const int n16 = (total - head) >> 4;
      if (head && tid == 64) *reinterpret_cast<float2 *>(gb) = *reinterpret_cast<const float2 *>(lb);
      float4 *gd = reinterpret_cast<float4 *>(gb + head);
      const float4 *ls = reinterpret_cast<const float4 *>(lb + head);
      for (int i = tid; i < n16; i += 256) HSGK_ROW_STORE(gd + i, ls[i]);
      if (total - head - 16 * n16 && tid == 128)
        *reinterpret_cast<float2 *>(gb + head + 16 * n16) = *reinterpret_cast<const float2 *>(lb + head + 16 * n16);
    }
    if (m0on) {                             // this workgroup's two partial sums
      const int64_t e0 = ((int64_t)b * gridDim.x + bx) * 2;
      for (int sidx = 0; sidx < 2; ++sidx) {
        if (m0l[sidx] < 0) continue;
        if (tid == 0) m0.lab[e0 + sidx] = m0l[sidx];
        unsigned long long *dst = m0.part + (e0 + sidx) * D;
        for (int i = tid; i < D; i += 256) dst[i] = mtab[sidx * D + i];
      }
    }
  } else {
  for (int j = w; j < 32; j += 4) {
    const int64_t row = rowi[j];
    if (row < 0) continue;
    const float n2 = nrm2[j];
    HSGK_DIV_SETUP(n2, r2)
    if (norms_out && lane == 0) { norms_out[2 * row] = nrm1[j]; norms_out[2 * row + 1] = n2; }
    const float *r = tile + j * C;
    float *eo = emb + row * C;
    float *lo = emb_loc + row * D;
    // fp16 copy of the emb_loc row for the first E-step filter level: the C main
    // columns (C % 64 == 0 here) in xh[row][C], the two location columns packed in
    // xt[row] (layout: score_tiles_f16.h)
    _Float16 *ho = xh ? xh + row * C : nullptr;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int sj = j & 15;
    float e2 = 0.0f;                         // |row - fp16(row)|^2, this lane's columns
    if (m0on) {
      const int L = __builtin_amdgcn_readfirstlane(seedl[j]);
      const int L0 = __builtin_amdgcn_readfirstlane(m0l[0]), L1 = __builtin_amdgcn_readfirstlane(m0l[1]);
      const int slot = L < 0 ? -1 : L == L0 ? 0 : L == L1 ? 1 : L < m0.K ? 2 : -1;
      unsigned long long *g = slot == 2 ? m0.sumq + ((int64_t)b * m0.K + L) * D : nullptr;
      if (slot != cslot || g != cg) { m0_flush(); cslot = slot; cg = g; }
    }
    for (int q = lane; q < NQ; q += 64) {
      const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sj) << 2));
      HSGK_ROW_STORE(reinterpret_cast<float4 *>(eo + 4 * q), v);
      float2 a, c2;
      a.x = HSGK_DIV(v.x, n2, r2); a.y = HSGK_DIV(v.y, n2, r2); c2.x = HSGK_DIV(v.z, n2, r2); c2.y = HSGK_DIV(v.w, n2, r2);
      HSGK_ROW_STORE(reinterpret_cast<float2 *>(lo + 4 * q), a);
      HSGK_ROW_STORE(reinterpret_cast<float2 *>(lo + 4 * q + 2), c2);
      if (m0on) {                            // (uniform)
        const long long f0 = to_fixed(a.x), f1 = to_fixed(a.y), f2 = to_fixed(c2.x), f3 = to_fixed(c2.y);
        cur[0] += f0; cur[1] += f1; cur[2] += f2; cur[3] += f3;     // (C <= 256 with the fusion on: one column pass)
      }
      if (ho || tmode) {
        const h4 hv = {(_Float16)a.x, (_Float16)a.y, (_Float16)c2.x, (_Float16)c2.y};
        if (ho) HSGK_ROW_STORE(reinterpret_cast<h4 *>(ho + 4 * q), hv);
        // tile order: the four halves wait in the quad's own LDS slot (read above by this lane, by nobody else)
        if (tmode) *reinterpret_cast<h4 *>(const_cast<float *>(r) + ((q ^ sj) << 2)) = hv;
        const float e0 = a.x - (float)hv[0], e1 = a.y - (float)hv[1];       // exact residuals
        const float e2b = c2.x - (float)hv[2], e3 = c2.y - (float)hv[3];
        e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2); e2 = fmaf(e2b, e2b, e2); e2 = fmaf(e3, e3, e2);
      }
    }
    if (ho || tmode)
      for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
    if (lane == 0) {
      float2 lv;
      lv.x = locv[2 * j] / n2;
      lv.y = locv[2 * j + 1] / n2;
      *reinterpret_cast<float2 *>(lo + C) = lv;
      if (m0on) { tcur[0] += to_fixed(lv.x); tcur[1] += to_fixed(lv.y); }
      if (ho || tmode) {
        const h2 hv = {(_Float16)lv.x, (_Float16)lv.y};
        const float e0 = lv.x - (float)hv[0], e1 = lv.y - (float)hv[1];
        e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2);
        // measured rounding error of the copy of this row, inflated against the rounding
        // of this very sum (bound: score_tiles_f16.h)
        xt[row] = make_uint2(__builtin_bit_cast(uint32_t, hv), __float_as_uint(sqrtf(e2) * 1.0001f));
      }
    }
  }
  if (m0on) {                             // this workgroup's two partial sums
    m0_flush();
    __syncthreads();
  HSGK_TS(6);
    const int64_t e0 = ((int64_t)b * gridDim.x + bx) * 2;
    for (int sidx = 0; sidx < 2; ++sidx) {
      if (m0l[sidx] < 0) continue;
      if (tid == 0) m0.lab[e0 + sidx] = m0l[sidx];
      unsigned long long *dst = m0.part + (e0 + sidx) * D;
      for (int i = tid; i < D; i += 256) dst[i] = mtab[sidx * D + i];
    }
  }
  if (tmode) {
    // the block's 16-byte pieces in operand order: piece (kb, lane = jj + 32 g) = row jj, columns 16 kb + 8 g .. + 7
    // (two quads of four halves each, from their LDS slots); one KiB contiguous per wave instruction
    __syncthreads();
    const int jj = lane & 31, gg = lane >> 5;
    const int64_t blk = (img_row0[b] + q0) >> 5;
    uint2 *dst = reinterpret_cast<uint2 *>(xhT + blk * 32 * C);
    for (int kb = w; kb < (C >> 4); kb += 4) {
      const int qa = 4 * kb + 2 * gg;
      const uint2 lo2 = *reinterpret_cast<const uint2 *>(tile + jj * C + ((qa ^ (jj & 15)) << 2));
      const uint2 hi2 = *reinterpret_cast<const uint2 *>(tile + jj * C + (((qa + 1) ^ (jj & 15)) << 2));
      uint4 pc = {lo2.x, lo2.y, hi2.x, hi2.y};
      *reinterpret_cast<uint4 *>(dst + (kb * 64 + lane) * 2) = pc;
    }
  }
  }
  HSGK_TS(7);
}

// --------------------------------------------------------------------------
// Persistent, software-pipelined form of prep_fast32_kernel (HSGK_PREP=pipe; C <= 256): a workgroup walks
// half tiles h = blockIdx.x, + gridDim.x, ... with TWO LDS tiles.  Per iteration: the plane loads and the
// bookkeeping loads of half tile i + 1 are issued, half tile i is computed from its LDS tile (chains, divide),
// then ONE full wait -- by then only those loads and the row stores of half tile i - 1 are outstanding, and
// both have had a whole compute phase to complete -- the prefetched registers go to the other LDS tile, and
// only then are the 85 KB of row stores of half tile i issued.  (The round-2 pipelined variant consumed the
// prefetch AFTER the stores: loads and stores share the vmcnt counter and complete out of order against each
// other, so that wait drained the stores it had just issued: 11.3 ms against 7.6.)  Same arithmetic, same
// outputs as prep_fast32_kernel.
__global__ __launch_bounds__(256) void prep_pipe32_kernel(
    const float *__restrict__ in, int C, int64_t HW, int ntiles, int B,
    const float *__restrict__ loc, int64_t loc_sb, const int64_t *__restrict__ labels,
    int has_ignore, int64_t ignore, const int32_t *__restrict__ tile_off,
    const int64_t *__restrict__ img_row0, const int32_t *__restrict__ seed_map, int64_t seed_sb,
    float eps, float *__restrict__ emb, float *__restrict__ emb_loc,
    int64_t *__restrict__ labels_out, int32_t *__restrict__ klab,
    float *__restrict__ norms_out, int64_t *__restrict__ rowmap_out,
    _Float16 *__restrict__ xh, uint2 *__restrict__ xt, PrepM0 m0) {
  const bool m0on = m0.part != nullptr;
  // workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for the global loads in flight
  // (the prefetch of the next half tile) and the stores of the previous one
  auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
  extern __shared__ float lds[];
  const int D = C + 2;
  const int NQ = C >> 2;
  // per buffer: tile [32][C], nrm1 [32], nrm2 [32], locv [64], rowi [32] i64, seedl [32], flags [4]
  const int buf_floats = 32 * C + 32 + 32 + 64 + 64 + 32 + 4;
  int *m0l = reinterpret_cast<int *>(lds + 2 * buf_floats);                        // [4]
  unsigned long long *mtab = reinterpret_cast<unsigned long long *>(m0l + 4);       // [2][D]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int jl = lane & 31, sub = lane >> 5;
  const int sw = jl & 15;
  const int64_t total = (int64_t)B * 2 * ntiles;

  // registers of the prefetched half tile
  float4 v1[8];
  int64_t pf_lab = 0, pf_base = 0;
  int pf_seed = -1;
  float pf_ly = 0.0f, pf_lx = 0.0f;
  auto issue = [&](int64_t h) {                     // h < total (uniform)
    const int b = (int)(h / (2 * ntiles));
    const int r2 = (int)(h - (int64_t)b * 2 * ntiles);
    const int t = r2 >> 1, sh = r2 & 1;
    const int64_t q0 = (int64_t)t * kTilePix + 32 * sh;
    const bool pix_ok = q0 + jl < HW;
    const float *src = in + (int64_t)b * C * HW + (pix_ok ? q0 + jl : HW - 1);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = min(2 * w + sub + 8 * u, NQ - 1);
      v1[u].x = src[(int64_t)(4 * q + 0) * HW];
      v1[u].y = src[(int64_t)(4 * q + 1) * HW];
      v1[u].z = src[(int64_t)(4 * q + 2) * HW];
      v1[u].w = src[(int64_t)(4 * q + 3) * HW];
    }
    if (w == 0) {                                   // bookkeeping loads: all independent of each other
      const int64_t pix = min((int64_t)t * kTilePix + lane, HW - 1);
      pf_lab = labels ? labels[(int64_t)b * HW + pix] : 0;
      pf_seed = seed_map[(int64_t)b * seed_sb + pix];
      pf_ly = loc[(int64_t)b * loc_sb + pix * 2 + 0];
      pf_lx = loc[(int64_t)b * loc_sb + pix * 2 + 1];
      pf_base = img_row0[b] + (tile_off ? (int64_t)tile_off[(int64_t)b * ntiles + t] : (int64_t)t * kTilePix);
    }
  };
  // prefetched registers -> LDS buffer `bf` (tile + bookkeeping); returns nothing, flags[0] = any kept pixel
  auto commit = [&](int64_t h, int bf) {
    float *tile = lds + bf * buf_floats;
    float *locv = tile + 32 * C + 64;
    int64_t *rowi = reinterpret_cast<int64_t *>(locv + 64);
    int *seedl = reinterpret_cast<int *>(rowi + 32);
    int *flags = seedl + 32;
    const int b = (int)(h / (2 * ntiles));
    const int r2 = (int)(h - (int64_t)b * 2 * ntiles);
    const int t = r2 >> 1, sh = r2 & 1;
    const int64_t p0 = (int64_t)t * kTilePix;
    const bool pix_ok = p0 + 32 * sh + jl < HW;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = 2 * w + sub + 8 * u;
      if (q < NQ)
        *reinterpret_cast<float4 *>(tile + jl * C + ((q ^ sw) << 2)) = pix_ok ? v1[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (w == 0) {
      const int64_t pix = p0 + lane;
      const bool keep = pix < HW && !(has_ignore && pf_lab == ignore);
      const unsigned long long m = __ballot(keep);
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      const int64_t row = keep ? pf_base + rank : -1;
      if ((lane >> 5) == sh) {
        rowi[lane & 31] = row;
        if (rowmap_out && pix < HW) rowmap_out[(int64_t)b * HW + pix] = row;
        if (keep) {
          labels_out[row] = pf_lab;
          klab[row] = pf_seed;
          locv[2 * (lane & 31) + 0] = pf_ly;
          locv[2 * (lane & 31) + 1] = pf_lx;
        }
        seedl[lane & 31] = keep ? pf_seed : -1;
      }
      const unsigned long long mh = sh ? (m >> 32) : (m & 0xffffffffull);
      if (lane == 0) flags[0] = mh ? 1 : 0;
    }
  };

  int64_t h = blockIdx.x;
  if (h >= total) return;
  issue(h);
  __builtin_amdgcn_s_waitcnt(0);
  commit(h, 0);
  __syncthreads();
  for (int it = 0; h < total; ++it, h += gridDim.x) {
    const int bf = it & 1;
    const int64_t hn = h + gridDim.x;
    const bool more = hn < total;
    if (more) issue(hn);                            // in flight during the phases below
    float *tile = lds + bf * buf_floats;
    float *nrm1 = tile + 32 * C;
    float *nrm2 = nrm1 + 32;
    float *locv = nrm2 + 32;
    int64_t *rowi = reinterpret_cast<int64_t *>(locv + 64);
    int *seedl = reinterpret_cast<int *>(rowi + 32);
    int *flags = seedl + 32;
    const int b = (int)(h / (2 * ntiles));
    const int r2 = (int)(h - (int64_t)b * 2 * ntiles);
    const bool live = flags[0] != 0;                // (uniform: written before the last barrier)
    if (live) {
      // phase 2a: C1 chain (wave 0, lanes 0..31); meanwhile the M-step slots
      if (w == 0 && sub == 0) {
        const float *r = tile + jl * C;
        float ss = 0.0f;
        for (int q = 0; q < NQ; ++q) {
          const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
          ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
        float n1 = sqrtf(ss);
        if (!(n1 >= eps)) n1 = eps;
        nrm1[jl] = n1;
      } else if (m0on && w == 1) {
        const int sl = lane < 32 ? seedl[lane] : -1;
        const unsigned long long mk = __ballot(sl >= 0);
        int L0 = -1, L1 = -1;
        if (mk) {
          L0 = __builtin_amdgcn_readlane(sl, __builtin_ctzll(mk));
          const unsigned long long m1 = __ballot(sl >= 0 && sl != L0);
          if (m1) L1 = __builtin_amdgcn_readlane(sl, __builtin_ctzll(m1));
        }
        if (lane == 0) { m0l[0] = L0; m0l[1] = L1; }
      } else if (m0on && w >= 2) {
        for (int i = tid - 128; i < 2 * D; i += 128) mtab[i] = 0ull;
      }
      lds_barrier();
      {                                             // phase 2b
        const float n1 = nrm1[jl];
        float *r = tile + jl * C;
        for (int q = 2 * w + sub; q < NQ; q += 8) {
          float4 *pv = reinterpret_cast<float4 *>(r + ((q ^ sw) << 2));
          float4 v = *pv;
          v.x = v.x / n1; v.y = v.y / n1; v.z = v.z / n1; v.w = v.w / n1;
          *pv = v;
        }
      }
      lds_barrier();
      if (w == 0 && sub == 0) {                     // phase 2c
        const float *r = tile + jl * C;
        float ss = 0.0f;
        for (int q = 0; q < NQ; ++q) {
          const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sw) << 2));
          ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
        }
        const float ly = locv[2 * jl], lx = locv[2 * jl + 1];
        ss = fmaf(ly, ly, ss);
        ss = fmaf(lx, lx, ss);
        float n2 = sqrtf(ss);
        if (!(n2 >= eps)) n2 = eps;
        nrm2[jl] = n2;
      }
    }
    // the ONE full wait of the iteration: loads of the next half tile (and the stores of the previous one),
    // then the prefetched registers go to the other LDS tile BEFORE this half tile's stores are issued
    __builtin_amdgcn_s_waitcnt(0);
    if (more) commit(hn, bf ^ 1);
    lds_barrier();
    if (live) {
      // phase 3 (as in prep_fast32_kernel)
      long long cur[4] = {0, 0, 0, 0}, tcur[2] = {0, 0};
      int cslot = -1;
      unsigned long long *cg = nullptr;
      auto m0_flush = [&]() {
        if (cslot == 0 || cslot == 1) {
          unsigned long long *t = mtab + cslot * D;
          if (lane < NQ)
            for (int i = 0; i < 4; ++i) atomicAdd(t + 4 * lane + i, (unsigned long long)cur[i]);
          if (lane == 0) { atomicAdd(t + C, (unsigned long long)tcur[0]); atomicAdd(t + C + 1, (unsigned long long)tcur[1]); }
        } else if (cslot == 2) {
          if (lane < NQ)
            for (int i = 0; i < 4; ++i) atomicAdd(cg + 4 * lane + i, (unsigned long long)cur[i]);
          if (lane == 0) { atomicAdd(cg + C, (unsigned long long)tcur[0]); atomicAdd(cg + C + 1, (unsigned long long)tcur[1]); }
        }
        cur[0] = cur[1] = cur[2] = cur[3] = 0; tcur[0] = tcur[1] = 0;
      };
      for (int j = w; j < 32; j += 4) {
        const int64_t row = rowi[j];
        if (row < 0) continue;
        const float n2 = nrm2[j];
        if (norms_out && lane == 0) { norms_out[2 * row] = nrm1[j]; norms_out[2 * row + 1] = n2; }
        const float *r = tile + j * C;
        float *eo = emb + row * C;
        float *lo = emb_loc + row * D;
        _Float16 *ho = xh ? xh + row * C : nullptr;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const int sj = j & 15;
        float e2 = 0.0f;
        if (m0on) {
          const int L = __builtin_amdgcn_readfirstlane(seedl[j]);
          const int L0 = __builtin_amdgcn_readfirstlane(m0l[0]), L1 = __builtin_amdgcn_readfirstlane(m0l[1]);
          const int slot = L < 0 ? -1 : L == L0 ? 0 : L == L1 ? 1 : L < m0.K ? 2 : -1;
          unsigned long long *g = slot == 2 ? m0.sumq + ((int64_t)b * m0.K + L) * D : nullptr;
          if (slot != cslot || g != cg) { m0_flush(); cslot = slot; cg = g; }
        }
        for (int q = lane; q < NQ; q += 64) {
          const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sj) << 2));
          *reinterpret_cast<float4 *>(eo + 4 * q) = v;
          float2 a, c2;
          a.x = v.x / n2; a.y = v.y / n2; c2.x = v.z / n2; c2.y = v.w / n2;
          *reinterpret_cast<float2 *>(lo + 4 * q) = a;
          *reinterpret_cast<float2 *>(lo + 4 * q + 2) = c2;
          if (m0on) {
            cur[0] += to_fixed(a.x); cur[1] += to_fixed(a.y); cur[2] += to_fixed(c2.x); cur[3] += to_fixed(c2.y);
          }
          if (ho) {
            const h4 hv = {(_Float16)a.x, (_Float16)a.y, (_Float16)c2.x, (_Float16)c2.y};
            *reinterpret_cast<h4 *>(ho + 4 * q) = hv;
            const float e0 = a.x - (float)hv[0], e1 = a.y - (float)hv[1];
            const float e2b = c2.x - (float)hv[2], e3 = c2.y - (float)hv[3];
            e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2); e2 = fmaf(e2b, e2b, e2); e2 = fmaf(e3, e3, e2);
          }
        }
        if (ho)
          for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off);
        if (lane == 0) {
          float2 lv;
          lv.x = locv[2 * j] / n2;
          lv.y = locv[2 * j + 1] / n2;
          *reinterpret_cast<float2 *>(lo + C) = lv;
          if (m0on) { tcur[0] += to_fixed(lv.x); tcur[1] += to_fixed(lv.y); }
          if (ho) {
            const h2 hv = {(_Float16)lv.x, (_Float16)lv.y};
            const float e0 = lv.x - (float)hv[0], e1 = lv.y - (float)hv[1];
            e2 = fmaf(e0, e0, e2); e2 = fmaf(e1, e1, e2);
            xt[row] = make_uint2(__builtin_bit_cast(uint32_t, hv), __float_as_uint(sqrtf(e2) * 1.0001f));
          }
        }
      }
      if (m0on) {
        m0_flush();
        lds_barrier();
        const int64_t e0 = ((int64_t)b * 2 * ntiles + r2) * 2;
        for (int sidx = 0; sidx < 2; ++sidx) {
          if (m0l[sidx] < 0) continue;
          if (tid == 0) m0.lab[e0 + sidx] = m0l[sidx];
          unsigned long long *dst = m0.part + (e0 + sidx) * D;
          for (int i = tid; i < D; i += 256) dst[i] = mtab[sidx * D + i];
        }
      }
    }
    lds_barrier();                                // mtab / m0l / this LDS tile free for the next rounds
  }
}

#ifdef HSGK_PREP_TIMING
extern "C" __attribute__((visibility("default"))) int hsgk_debug_prep_timing(unsigned long long *out) {
  unsigned long long h[8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prep_ts), sizeof(h)) != hipSuccess) return -1;
  for (int i = 0; i < 8; ++i) out[i] = h[i];
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prep_ts), z, sizeof(z));
  return 0;
}
#endif
int launch_prep(const hsgk_segkm_args &a, const int32_t *tile_off, const ChunkTable &t,
                int32_t *klab, hipStream_t s, _Float16 *xh, uint2 *xt, bool *wrote_half,
                const PrepM0 *m0, bool *wrote_m0, _Float16 *xmT, bool *wrote_tiles) {
  if (wrote_half) *wrote_half = false;
  if (wrote_m0) *wrote_m0 = false;
  if (wrote_tiles) *wrote_tiles = false;
  PrepM0 m0v{nullptr, nullptr, nullptr, 0};
  const int64_t HW = (int64_t)a.H * a.W;
  const int ntiles = (int)((HW + kTilePix - 1) / kTilePix);
  const bool fast = (a.C % 64) == 0;
  const int S = fast ? a.C : (a.C | 1);
  size_t floats = (size_t)64 * S + 64 + 64 + 128;
  floats = (floats + 1) & ~(size_t)1;     // keep the int64 row table 8-byte aligned
  size_t lds = floats * 4 + 64 * 8;
  HSGK_REQUIRE(lds <= 160 * 1024, "embedding dimension too large for the prep tile");
  static const int tile32 = [] {
    const char *e = getenv("HSGK_PREP_TILE");        // "64" selects the 64-pixel fast kernel
    return (e && e[0] == '6') ? 0 : 1;
  }();
  auto kern = fast ? prep_fast_kernel : prep_kernel;
  dim3 grid(ntiles, a.B);
  if (fast && tile32) {
    HSGK_REQUIRE(HW * 20 < ((int64_t)1 << 32), "image too large for the 32-bit lane offsets of the prep kernel");
    const char *fe = getenv("HSGK_PREP_FLAT");        // "0": phase 3 as in rounds 2-5 (A/B; read per call)
    kern = (a.C <= 256 && !(fe && fe[0] == '0')) ? prep_fast32_kernel<true> : prep_fast32_kernel<false>;
    lds = ((size_t)32 * a.C + kPrep32TilePad + 32 + 32 + 64) * 4 + 32 * 8 + (32 + 4 + 32 + 2) * 4;
#ifdef HSGK_PREP_PAD_LDS
    lds += HSGK_PREP_PAD_LDS;
#endif
    grid.x = 2 * ntiles;
    if (m0 && m0->part && a.C <= 256) {       // fused first M-step (this kernel only; one column pass per row)
      m0v = *m0;
      lds += (size_t)2 * (a.C + 2) * 8;       // its two LDS slots
      HSGK_CHECK_HIP(hipMemsetAsync(m0v.lab, 0xFF, sizeof(int32_t) * (size_t)a.B * grid.x * 2, s));
      if (wrote_m0) *wrote_m0 = true;
    }
  }
  const char *pipe_env = getenv("HSGK_PREP");
  if (fast && tile32 && xmT && tile_off == nullptr && HW % 32 == 0 && xt != nullptr && !(pipe_env && pipe_env[0] == 'p')) {
    // the fp16 copy in tile order straight from this kernel (no labels to compact by: rows = pixels, a half tile =
    // one 32-row block); the caller passes xh = null when nothing reads the row-major copy
    m0v.tiles = xmT;
    if (wrote_tiles) *wrote_tiles = true;
  }
  {
    const char *oe = getenv("HSGK_PREP_ORDER");      // "0": workgroup ids as they come (A/B; read per call)
    m0v.xcd_order = !(oe && oe[0] == '0');
    // two scheduling experiments, measured neutral and off by default (profiles/r06_prep_ab.txt): bit 1 = issue
    // priority for the wave that walks a norm chain, bit 2 = the `embeddings` rows stored before the second chain
    const char *xe = getenv("HSGK_PREP_X");
    m0v.xcd_order |= xe ? (atoi(xe) & 6) : 0;
  }
  if (fast && wrote_half) *wrote_half = xh != nullptr || m0v.tiles != nullptr;      // both fast kernels write the fp16 copy
  {
    const char *pe = getenv("HSGK_PREP");            // "pipe": the persistent two-tile kernel (A/B; read per call)
    if (fast && tile32 && a.C <= 256 && pe && pe[0] == 'p') {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      (void)hipGetLastError();
      const size_t bufb = ((size_t)32 * a.C + 32 + 32 + 64 + 64 + 32 + 4) * 4;
      const size_t lds2 = 2 * bufb + 16 + (size_t)2 * (a.C + 2) * 8;
      const int per_cu = lds2 <= 78 * 1024 ? 2 : 1;
      const char *ge = getenv("HSGK_PREP_WGS");      // workgroups per CU override (tuning)
      const int wpc = ge ? atoi(ge) : per_cu;
      const int64_t total = (int64_t)a.B * 2 * ntiles;
      const int g = (int)std::min<int64_t>(total, (int64_t)cus * (wpc > 0 ? wpc : per_cu));
      HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(prep_pipe32_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      hipLaunchKernelGGL(prep_pipe32_kernel, dim3(g), dim3(256), lds2, s, a.embeddings, a.C, HW, ntiles, a.B,
                         a.loc, a.loc_batch_stride, a.labels, a.has_ignore, a.ignore_index,
                         tile_off, t.img_row0, a.seed_map, a.seed_batch_stride, HSGK_EPS, a.out_embeddings,
                         a.out_embeddings_loc, a.out_labels, klab, a.out_norms, a.out_rowmap, xh, xt, m0v);
      HSGK_LAUNCH_CHECK();
      return 0;
    }
  }
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a.embeddings, a.C, HW, ntiles,
                     a.loc, a.loc_batch_stride, a.labels, a.has_ignore, a.ignore_index,
                     tile_off, t.img_row0, a.seed_map, a.seed_batch_stride, HSGK_EPS, a.out_embeddings,
                     a.out_embeddings_loc, a.out_labels, klab, a.out_norms, a.out_rowmap, xh, xt, m0v);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// --------------------------------------------------------------------------
// Backward of the prep stage.  Same 64-pixel tiling as the forward pass.
//   el = v / n2, v = (e, loc)      g_v = (g_el - el <el, g_el>) / n2
//   e  = x / n1                    g_x = (g   - e  <e , g   >) / n1,  g = g_e + g_v[:C]
// (the eps-clamped branches are linear: g / eps).  Wave per row for the two
// dot products (shuffle reduction; gradients are tolerance quantities), LDS
// transpose, then coalesced NCHW plane writes with lanes = pixels.
__global__ __launch_bounds__(256) void prep_bwd_kernel(
    const float *__restrict__ g_emb, const float *__restrict__ g_emb_loc,
    const float *__restrict__ emb, const float *__restrict__ emb_loc,
    const float *__restrict__ norms, const int64_t *__restrict__ rowmap, int C, int64_t HW,
    float eps, float *__restrict__ gx) {
  extern __shared__ float lds[];
  const int S = C | 1;
  float *tile = lds;                        // [64][S]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int t = blockIdx.x, b = blockIdx.y;
  const int64_t p0 = (int64_t)t * kTilePix;
  const int D = C + 2;
  for (int j = w; j < 64; j += 4) {
    const int64_t pix = p0 + j;
    int64_t row = -1;
    if (pix < HW) row = rowmap ? rowmap[(int64_t)b * HW + pix] : (int64_t)b * HW + pix;
    float *dst = tile + j * S;
    if (row < 0) {
      for (int c = lane; c < C; c += 64) dst[c] = 0.0f;
      continue;
    }
    const float n1 = norms[2 * row], n2 = norms[2 * row + 1];
    const float *e = emb + row * C, *el = emb_loc + row * D;
    const float *ge = g_emb ? g_emb + row * C : nullptr;
    const float *gl = g_emb_loc ? g_emb_loc + row * D : nullptr;
    float dot2 = 0.0f;
    if (gl && n2 > eps) {
      for (int c = lane; c < D; c += 64) dot2 += el[c] * gl[c];
      for (int off = 32; off > 0; off >>= 1) dot2 += __shfl_xor(dot2, off);
    }
    float dot1 = 0.0f;
    for (int c = lane; c < C; c += 64) {
      float g = ge ? ge[c] : 0.0f;
      if (gl) g += (gl[c] - el[c] * dot2) / n2;
      dst[c] = g;                            // g = g_e + g_v[c]
      dot1 += e[c] * g;
    }
    for (int off = 32; off > 0; off >>= 1) dot1 += __shfl_xor(dot1, off);
    if (!(n1 > eps)) dot1 = 0.0f;
    for (int c = lane; c < C; c += 64) dst[c] = (dst[c] - e[c] * dot1) / n1;
  }
  __syncthreads();
  const int64_t pix = p0 + lane;
  if (pix < HW) {
    float *out = gx + (int64_t)b * C * HW + pix;
    for (int c = w; c < C; c += 4) out[(int64_t)c * HW] = tile[lane * S + c];
  }
}

// 32-pixel variant for C % 64 == 0, C <= 512 (NV 16-byte quads per lane and row): a wave keeps
// four rows of all four streams in flight (g_emb, g_emb_loc, emb, emb_loc: vector loads, nothing
// is read twice), both dot products come from registers, the result goes through an
// XOR-swizzled LDS tile (33 KiB -> four workgroups per CU) and leaves as 128-byte plane segments.
// (The 64-pixel kernel above -- scalar loads, one dependent load-reduce chain per row, LDS
// read-modify-write -- ran at 31 ms for 48 x 256 x 448 x 448; gradients are tolerance quantities.)
template <int NV>
__global__ __launch_bounds__(256) void prep_bwd32_kernel(
    const float *__restrict__ g_emb, const float *__restrict__ g_emb_loc,
    const float *__restrict__ emb, const float *__restrict__ emb_loc,
    const float *__restrict__ norms, const int64_t *__restrict__ rowmap, int C, int64_t HW,
    float eps, float *__restrict__ gx) {
  extern __shared__ float lds[];
  float *tile = lds;                        // [32][C], quads swizzled by (pixel & 15)
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y;
  const int64_t q0 = (int64_t)blockIdx.x * 32;      // first pixel of this half tile
  const int D = C + 2, NQ = C >> 2;
  bool qon[NV];
  int qc[NV];
#pragma unroll
  for (int h = 0; h < NV; ++h) { qon[h] = lane + 64 * h < NQ; qc[h] = min(lane + 64 * h, NQ - 1); }
  for (int j0 = w; j0 < 32; j0 += 16) {             // rows j0, j0 + 4, j0 + 8, j0 + 12 of this wave
    int64_t row[4];
    f4u ge[4][NV], gl[4][NV], e[4][NV], el[4][NV];
    float2 tl[4], tg[4];
    float n1[4], n2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t pix = q0 + j0 + 4 * u;
      row[u] = -1;
      if (pix < HW) row[u] = rowmap ? rowmap[(int64_t)b * HW + pix] : (int64_t)b * HW + pix;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = row[u] < 0 ? 0 : row[u];    // (clamped: loads stay unconditional)
      n1[u] = norms[2 * r];
      n2[u] = norms[2 * r + 1];
      const f4u z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        ge[u][h] = g_emb ? *reinterpret_cast<const f4u *>(g_emb + r * C + 4 * qc[h]) : z;
        gl[u][h] = g_emb_loc ? *reinterpret_cast<const f4u *>(g_emb_loc + r * D + 4 * qc[h]) : z;
        e[u][h] = *reinterpret_cast<const f4u *>(emb + r * C + 4 * qc[h]);
        el[u][h] = *reinterpret_cast<const f4u *>(emb_loc + r * D + 4 * qc[h]);
      }
      tl[u] = make_float2(emb_loc[r * D + C], emb_loc[r * D + C + 1]);
      tg[u] = g_emb_loc ? make_float2(g_emb_loc[r * D + C], g_emb_loc[r * D + C + 1]) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + 4 * u;
      float4 o[NV];
#pragma unroll
      for (int h = 0; h < NV; ++h) o[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row[u] >= 0) {                              // (wave-uniform)
        float dot2 = 0.0f;
        if (g_emb_loc && n2[u] > eps) {
#pragma unroll
          for (int h = 0; h < NV; ++h)
            if (qon[h]) dot2 += el[u][h].x * gl[u][h].x + el[u][h].y * gl[u][h].y + el[u][h].z * gl[u][h].z + el[u][h].w * gl[u][h].w;
          if (lane == 0) dot2 += tl[u].x * tg[u].x + tl[u].y * tg[u].y;
          for (int off = 32; off > 0; off >>= 1) dot2 += __shfl_xor(dot2, off);
        }
        float4 g[NV];
        float dot1 = 0.0f;
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          g[h] = make_float4(ge[u][h].x, ge[u][h].y, ge[u][h].z, ge[u][h].w);
          if (g_emb_loc) {
            g[h].x += (gl[u][h].x - el[u][h].x * dot2) / n2[u];
            g[h].y += (gl[u][h].y - el[u][h].y * dot2) / n2[u];
            g[h].z += (gl[u][h].z - el[u][h].z * dot2) / n2[u];
            g[h].w += (gl[u][h].w - el[u][h].w * dot2) / n2[u];
          }
          if (qon[h]) dot1 += e[u][h].x * g[h].x + e[u][h].y * g[h].y + e[u][h].z * g[h].z + e[u][h].w * g[h].w;
        }
        for (int off = 32; off > 0; off >>= 1) dot1 += __shfl_xor(dot1, off);
        if (!(n1[u] > eps)) dot1 = 0.0f;
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          o[h].x = (g[h].x - e[u][h].x * dot1) / n1[u];
          o[h].y = (g[h].y - e[u][h].y * dot1) / n1[u];
          o[h].z = (g[h].z - e[u][h].z * dot1) / n1[u];
          o[h].w = (g[h].w - e[u][h].w * dot1) / n1[u];
        }
      }
#pragma unroll
      for (int h = 0; h < NV; ++h)
        if (qon[h]) *reinterpret_cast<float4 *>(tile + j * C + (((lane + 64 * h) ^ (j & 15)) << 2)) = o[h];
    }
  }
  __syncthreads();
  // plane writes: lane = (pixel, quad parity); every store instruction covers two 128-byte segments
  const int jl = lane & 31, sub = lane >> 5;
  const int64_t pix = q0 + jl;
  if (pix < HW) {
    float *out = gx + (int64_t)b * C * HW + pix;
    for (int q = 2 * w + sub; q < NQ; q += 8) {
      const float4 v = *reinterpret_cast<const float4 *>(tile + jl * C + ((q ^ (jl & 15)) << 2));
      out[(int64_t)(4 * q + 0) * HW] = v.x;
      out[(int64_t)(4 * q + 1) * HW] = v.y;
      out[(int64_t)(4 * q + 2) * HW] = v.z;
      out[(int64_t)(4 * q + 3) * HW] = v.w;
    }
  }
}

int launch_prep_bwd(const float *g_emb, const float *g_emb_loc, const float *emb,
                    const float *emb_loc, const float *norms, const int64_t *rowmap, int B, int C,
                    int H, int W, float eps, float *gx, hipStream_t s) {
  const int64_t HW = (int64_t)H * W;
  const int ntiles = (int)((HW + kTilePix - 1) / kTilePix);
  if ((C % 64) == 0 && C <= 512) {
    const size_t lds32 = (size_t)32 * C * 4;
    auto kern = C <= 256 ? prep_bwd32_kernel<1> : prep_bwd32_kernel<2>;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
    hipLaunchKernelGGL(kern, dim3(2 * ntiles, B), dim3(256), lds32, s, g_emb, g_emb_loc, emb,
                       emb_loc, norms, rowmap, C, HW, eps, gx);
    HSGK_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)64 * (C | 1) * 4;
  HSGK_REQUIRE(lds <= 160 * 1024, "embedding dimension too large for the prep tile");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(prep_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(prep_bwd_kernel, dim3(ntiles, B), dim3(256), lds, s, g_emb, g_emb_loc, emb,
                     emb_loc, norms, rowmap, C, HW, eps, gx);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // namespace hsgk
