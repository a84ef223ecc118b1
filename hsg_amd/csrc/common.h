// Internal helpers shared by the libhsgk translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/hsgk.h"

namespace hsgk {

void set_error(const char *fmt, ...);

#define HSGK_CHECK_HIP(expr)                                                   \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) {                                                   \
      hsgk::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,            \
                      hipGetErrorString(e__));                                 \
      return -2;                                                               \
    }                                                                          \
  } while (0)

#define HSGK_REQUIRE(cond, msg)                                                \
  do {                                                                         \
    if (!(cond)) {                                                             \
      hsgk::set_error("%s:%d requirement failed: %s (%s)", __FILE__, __LINE__, \
                      #cond, msg);                                             \
      return -1;                                                               \
    }                                                                          \
  } while (0)

#define HSGK_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      hsgk::set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__,        \
                      hipGetErrorString(e__));                                 \
      return -3;                                                               \
    }                                                                          \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carves consecutive 256-byte aligned regions out of one workspace buffer.
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *p) : base(static_cast<char *>(p)) {}
  template <typename T> T *take(size_t count) {
    T *p = reinterpret_cast<T *>(base + off);
    off += align_up(count * sizeof(T), 256);
    return p;
  }
};

// Chunk table of one batch (device resident, built by build_tables_kernel).
// A chunk is <= HSGK_CHUNK consecutive rows of ONE image.
struct ChunkTable {
  int64_t *img_row0;   // [B+1] first output row of each image
  int32_t *img_chunk0; // [B+1] first chunk of each image
  int64_t *chunk_row0; // [max_chunks]
  int32_t *chunk_rows; // [max_chunks]
  int32_t *chunk_img;  // [max_chunks]
};

constexpr int kTilePix = 64;      // prep kernel tile (pixels)
constexpr int kAssignTile = 256;  // assign kernel tile (rows)

// fixed-point conversion of the exact sums (canonical order C2x):
// q = rint(x * 2^40), round to nearest even
#if defined(__HIPCC__)
__device__ inline long long to_fixed(float x) {
  // (the oracle computes q as an exactly rounded hi / lo split in fp32; this is the same
  // integer in three instructions) x * 2^40 is exact in fp64, and adding 1.5 * 2^52 (ulp 1
  // there) rounds it to the nearest-even integer, which then sits in the low mantissa bits
  // (|x| * 2^40 < 2^51)
  const double magic = 6755399441055744.0;
  const double t = __builtin_fma((double)x, 1099511627776.0, magic);
  return __builtin_bit_cast(long long, t) - __builtin_bit_cast(long long, magic);
}
#endif

// Working labels of the Lloyd loop.  A label buffer is int32 per row -- or, when bit 0 of the pointer is set
// (labels_as_u8), ONE BYTE per row (K <= 256).  Why: every E-step rewrites all labels, 128 bytes per 32-row wave
// tile, and such isolated line writes into a read stream cost a DRAM row activation and a bus turn-around each
// (~66 ns of a channel: 38.5 MB of int32 labels cost the first-level kernel 78 us per launch at 48 x 448 x 448,
// 21 us as bytes, tools/probes/ab_estore.sh) -- the price follows the number of 128-byte lines, not the number of
// store instructions or their cache policy (nt / sc1 / staged bursts: no change).  The tag travels inside the
// pointer so that the kernels that read or write labels (every E-step level, the sums update, the verify pass)
// keep their signatures; buffers are 256-byte aligned, so the bit is free.
inline int32_t *labels_as_u8(void *bytes) {
  return reinterpret_cast<int32_t *>(reinterpret_cast<uintptr_t>(bytes) | (uintptr_t)1);
}
inline bool labels_are_u8(const int32_t *lab) { return (reinterpret_cast<uintptr_t>(lab) & (uintptr_t)1) != 0; }
inline int32_t *labels_base(const int32_t *lab) {
  return reinterpret_cast<int32_t *>(reinterpret_cast<uintptr_t>(lab) & ~(uintptr_t)1);
}
#if defined(__HIPCC__)
__device__ __forceinline__ void put_label(int32_t *lab, int64_t i, int v) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(lab);
  if (a & 1) reinterpret_cast<uint8_t *>(a ^ 1)[i] = (uint8_t)v;
  else lab[i] = v;
}
__device__ __forceinline__ int get_label(const int32_t *lab, int64_t i) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(lab);
  return (a & 1) ? (int)reinterpret_cast<const uint8_t *>(a ^ 1)[i] : lab[i];
}
#endif

// First M-step fused into the prep kernel (the seed-grid labels are known there): every
// 32-pixel prep workgroup leaves the exact sums of its rows, split over at most two seed
// labels, in part [B][WT][2][D] / lab [B][WT][2] (WT = workgroups per image; lab = -1: unused);
// pixels of a third label go to sumq with global atomics.  m0_reduce then folds the partials
// into sumq.  part == nullptr: off.
struct PrepM0 {
  unsigned long long *part;
  int32_t *lab;
  unsigned long long *sumq;
  int K;
  // (rides along: the 32-pixel prep kernel writes the fp16 copy of its rows in TILE order here -- score_tiles_f16t.h
  //  -- when every image keeps all its pixels in place and H * W % 32 == 0: a half tile is one 32-row block)
  _Float16 *tiles = nullptr;
  // (rides along too) 1: workgroup ids of an image -- which the dispatcher deals round-robin over the 8 XCDs -- are
  // mapped to half tiles so that each XCD walks one CONTIGUOUS eighth of the image (prep_fast32_kernel)
  int xcd_order = 0;
};
int launch_m0_reduce(const PrepM0 &m0, int B, int WT, int d, hipStream_t s);

// ---- kernels implemented across the .hip files (host launchers) -----------
__global__ void init_meta_kernel(hsgk_segkm_meta *meta, int has_labels);
int launch_count_valid(const int64_t *labels, int B, int64_t HW, int has_ignore,
                       int64_t ignore, int32_t *tile_cnt, hsgk_segkm_meta *meta,
                       hipStream_t s);
int launch_build_tables(const int32_t *tile_cnt, int B, int64_t HW, int ntiles,
                        int32_t *tile_off, ChunkTable t, int max_chunks,
                        hsgk_segkm_meta *meta, hipStream_t s);
int launch_prep(const hsgk_segkm_args &a, const int32_t *tile_off,
                const ChunkTable &t, int32_t *klab, hipStream_t s, _Float16 *xm = nullptr,
                uint2 *xt = nullptr, bool *wrote_half = nullptr, const PrepM0 *m0 = nullptr,
                bool *wrote_m0 = nullptr, _Float16 *xmT = nullptr, bool *wrote_tiles = nullptr);
// whether launch_assign_half_wide2 will take the tile-order kernel for this shape (then nothing reads the row-major copy)
bool assign_half_wide2_tiles(int d, int K, int max_chunks);
int launch_prep_bwd(const float *g_emb, const float *g_emb_loc, const float *emb,
                    const float *emb_loc, const float *norms, const int64_t *rowmap, int B, int C,
                    int H, int W, float eps, float *gx, hipStream_t s);
int launch_normalize_rows(const float *x, int64_t n, int d, float eps, float *out,
                          float *norms, hipStream_t s);

// partial [chunks][K][d] holds only the rows flagged in pmask [chunks][K]
int launch_accumulate(const float *x, int d, const int32_t *klab,
                      const ChunkTable &t, int max_chunks, int K, float *partial,
                      unsigned char *pmask, const hsgk_segkm_meta *meta, hipStream_t s);
int launch_finalize(const float *partial, const unsigned char *pmask, int d, int K, int B,
                    const ChunkTable &t, int max_chunks_per_image, float eps, float *cent,
                    hipStream_t s);
int launch_assign(const float *x, int d, const float *cent, int K,
                  const ChunkTable &t, int max_chunks, int32_t *klab, float *best,
                  const hsgk_segkm_meta *meta, hipStream_t s);

// unit-norm fast path (bf16 split filter + exact re-score); falls back to
// launch_assign when the shape is not eligible or HSGK_ASSIGN=fp32
int launch_assign_fast(const float *x, int d, const float *cent, int K, int B, const ChunkTable &t,
                       int max_chunks, int32_t *klab, float *best, void *qrows,
                       int32_t *qcount, const hsgk_segkm_meta *meta, hipStream_t s);

// M-step with exact (fixed-point) segment sums, updated from the rows whose label changed
// (sums_fx.hip): prev / cur int32 labels [rows], sumq [B][K][d] int64
bool sums_fx_eligible(int d);
int launch_update_sums(const float *x, int d, const int32_t *prev, const int32_t *cur,
                       const ChunkTable &t, int max_chunks, int K, long long *sumq,
                       const hsgk_segkm_meta *meta, hipStream_t s, int unit_cols = 0);
// (unit_cols: leading columns the caller guarantees within [-1, 1] -- the normalised embedding columns; they may
//  take the matrix-core route of the update, sums_fx.hip)
// zero_a[0 .. na) and zero_b[0] (optional): queue counters of the E-step that follows, reset
// here instead of by two memsets per iteration; errc (optional) [B * K]: the fp16 rounding error of every centroid
// row it writes (what the hi-plane filters of K > 64 otherwise measure with a launch of their own)
int launch_finalize_fx(const long long *sumq, int d, int K, int B, float eps, float *cent,
                       hipStream_t s, int32_t *zero_a = nullptr, int na = 0, int32_t *zero_b = nullptr,
                       float *errc = nullptr, int32_t *keep_a = nullptr,
                       bool keep_valid = false);   // keep_a[i] = keep_valid ? zero_a[i] : 0 before the zeroing

// three-level E-step (fp16 copy -> bf16x3 on the undecided rows -> exact chains)
inline int half_main_cols_host(int d) { return d & ~63; }
constexpr int kHalfSlackRowsHost = 2 * 8 * 32 + 64;     // == kHalfSlackRows (score_tiles_f16.h)
bool assign_split_eligible(int d, int K);
bool assign_half_eligible(int d, int K);
int launch_to_half_rows(const float *x, const ChunkTable &t, int max_chunks, int d, _Float16 *xm,
                        uint2 *xt, const hsgk_segkm_meta *meta, hipStream_t s);
bool assign_half_wide_eligible(int d, int K);          // 64 < K <= 128: hi-plane fp16 filter -> exact chains
int launch_assign_half_wide(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent,
                            float *errc, int K, int B, const ChunkTable &t, int max_chunks, int32_t *klab,
                            void *qrows, int32_t *qcount, const hsgk_segkm_meta *meta, hipStream_t s,
                            bool table_ready = false,    // errc measured and the counters zeroed by launch_finalize_fx
                            int32_t *hard_rows = nullptr, int32_t *hard_count = nullptr, int64_t hard_cap = 0);
// hard_rows [B][hard_cap] / hard_count [B]: the rows whose exact-queue entry asks for all K centroids are listed
// per image and scored by the dense fp32 matrix-pipe pass (kmeans.hip: assign_hard_rows_kernel); null: one entry
// at a time inside the exact pass
bool assign_half_wide2_eligible(int d, int K);         // 128 < K <= 256: two table halves per pass
int launch_assign_half_wide2(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent,
                             float *errc, int K, int B, const ChunkTable &t, int max_chunks, int32_t *klab,
                             void *state, void *qrows, int32_t *qcount, const hsgk_segkm_meta *meta,
                             hipStream_t s, const _Float16 *xmT = nullptr, bool table_ready = false,
                             int32_t *hard_rows = nullptr, int32_t *hard_count = nullptr, int64_t hard_cap = 0);
int launch_assign_half(const float *x, const _Float16 *xm, const uint2 *xt, int d, const float *cent, int K, int B,
                       const ChunkTable &t, int max_chunks, int32_t *klab, int32_t *q1,
                       int32_t *q1count, int64_t q1cap, void *qrows, int32_t *qcount,
                       const hsgk_segkm_meta *meta, hipStream_t s, bool counters_zeroed = false,
                       const _Float16 *xmT = nullptr);      // xmT: the rows in tile order (score_tiles_f16t.h)
int launch_rows_to_tiles(const _Float16 *xm, int d, int64_t rows, _Float16 *xmT, hipStream_t s);

// small feature maps: the whole Lloyd loop of an image in one workgroup, one launch per call
bool lloyd_small_eligible(int d, int K, int B, int64_t rows_per_image);
int launch_lloyd_small(const float *x, const _Float16 *xm, const uint2 *xt, int d, int K, int B, int iterations,
                       const ChunkTable &t, int32_t *lab_a, int32_t *lab_b, long long *sumq, float *cent,
                       void *qrows, int32_t *counters, bool first_sums_ready, hsgk_segkm_meta *meta,
                       int64_t rows_per_image, bool single_group, float *cent_multi, hipStream_t s);
int lloyd_small_groups(int B, int64_t rows_per_image);
int lloyd_small_rows_max();        // rows per image ONE workgroup of the fused kernel accepts
size_t lloyd_small_cent_floats(int d, int K, int B);

size_t relabel_scan_bytes(int64_t table_cap);      // scratch of the chained scans (scan_tmp of launch_relabel)
int launch_relabel(const hsgk_segkm_args &a, const ChunkTable &t, int max_chunks,
                   const int32_t *klab, int32_t *table, int32_t *scan_tmp,
                   hipStream_t s);
// exchange.hip: segment sums of one or two row sets by ids in [0, P), order C2 (sorted runs per chunk, column slices)
int launch_sorted_sums(const float *xa, int da, const float *xb, int db, const int64_t *ids, int64_t n, int64_t P,
                       void *chunks, int32_t *seg_range, int64_t *pool_ids, float *pool, int64_t pool_rows,
                       int32_t *pool_used, float *table, int32_t *status, hipStream_t s);
size_t sorted_sums_chunk_bytes();
int launch_i64_to_i32(const int64_t *in, int64_t n, int32_t *out, hipStream_t s);
int launch_i32_to_i64(const int32_t *in, int64_t n, int64_t *out, hipStream_t s);
int launch_flat_table(int64_t n, ChunkTable t, int max_chunks, hsgk_segkm_meta *meta,
                      hipStream_t s);

}  // namespace hsgk
