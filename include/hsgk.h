/*
 * hsgk.h -- C ABI of libhsgk.so: MI355X (gfx950) kernels for the HSG
 * dense-pixel clustering / pixel-contrast hot path.
 *
 * The reference (twke18/HSG) is pure Python on torch tensors, so "the FFI a
 * maintainer would bind" is a ctypes stub.  Every entry
 * point below names the reference callable it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - float = IEEE binary32, labels/indices at the boundary are int64 (torch
 *     `long`), internal working labels are int32;
 *   - every function enqueues work on `stream` (a hipStream_t passed as
 *     void*) and returns immediately: no allocation, no synchronisation;
 *     scratch comes from the caller (`*_workspace_bytes`);
 *   - return value 0 = ok, negative = error; hsgk_last_error() gives the
 *     thread-local message.  Errors detected on the device (label range too
 *     large for the relabel table, negative labels) are reported through the
 *     `meta` block, see hsgk_segkm_meta.
 *   - floating-point summation orders are fixed ("canonical orders" C1, C2, C2x,
 *     DESIGN.md section 4; C2x = exact fixed-point segment sums in the Lloyd loop of
 *     hsgk_segment_by_kmeans) so results are run-to-run deterministic and
 *     bit-identical to oracle/hsg_oracle.c.
 */
#ifndef HSGK_H_
#define HSGK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSGK_VERSION 401
#define HSGK_CHUNK 2048          /* rows per segment-sum chunk (order C2)      */
#define HSGK_EPS 1e-12f          /* normalize_embedding eps (general/common.py:101) */

typedef void *hsgk_stream_t;     /* hipStream_t */

#if defined(__GNUC__)
#define HSGK_API __attribute__((visibility("default")))
#else
#define HSGK_API
#endif

HSGK_API int hsgk_version(void);
HSGK_API const char *hsgk_last_error(void);

/* ---- optional HIP-event profiling of the kernel groups --------------------
 * While enabled, every group launch is bracketed by hipEvents on the caller's
 * stream; hsgk_profile_collect waits for them, returns summed milliseconds and
 * launch counts per kind, and clears the log.                                 */
#define HSGK_PROF_PREP 0        /* count + tables + prep kernels              */
#define HSGK_PROF_ACCUMULATE 1  /* M-step sums (exact update or streaming pass)  */
#define HSGK_PROF_FINALIZE 2    /* M-step: sums -> normalised centroids         */
#define HSGK_PROF_ASSIGN 3      /* E-step (the roofline kernel)               */
#define HSGK_PROF_RELABEL 4
#define HSGK_PROF_KINDS 5
HSGK_API void hsgk_profile_enable(int on);
HSGK_API int hsgk_profile_collect(double *ms_sum, int64_t *count);

/* ---- optional verification of the filtered E-steps ------------------------
 * While enabled, every filtered E-step inside hsgk_segment_by_kmeans /
 * hsgk_kmeans_with_initial_labels is re-done by the exact fp32 E-step and the two label
 * vectors are compared on the device.  hsgk_verify_collect synchronises the current
 * device, returns and clears the counters (rows compared, rows whose labels differ: the
 * filters only label provably unique maxima, so the second must be 0).  Per device.    */
HSGK_API void hsgk_verify_enable(int on);
HSGK_API int hsgk_verify_collect(uint64_t *rows_compared, uint64_t *rows_differing);

/* ---- device-side result block of hsgk_segment_by_kmeans ------------------ */
typedef struct hsgk_segkm_meta {
  int64_t n_rows;        /* N: kept pixels over the whole batch                */
  int64_t n_segments;    /* number of distinct (image, cluster, label) ids     */
  int64_t label_min;     /* min / max of kept labels (0 / 0 when no labels)    */
  int64_t label_max;
  int64_t n_chunks;      /* chunks actually used                               */
  int64_t error;         /* 0 ok; 1 negative label; 2 relabel table too small; 3 the co-operating
                            workgroups of a small-map call waited 10 s for each other (the labels
                            are then not valid: repeat the call with HSGK_SEGKM_ONE_GROUP)  */
  int64_t relabel_mode;  /* 0 direct table, 1 label-ranked table               */
  int64_t relabel_L;     /* effective label extent used by the table           */
} hsgk_segkm_meta;

/* ---- hsg/utils/general/common.py:101-120 normalize_embedding ------------- */
/* rows [n,d] -> out [n,d] (may alias x); norms [n] (nullable) receives the
 * clamped norms.  Backward = hsgk_segment_reduce_bwd(mode 0, n = 0) on the
 * rows (same formula as a prototype row).                                     */
HSGK_API int hsgk_normalize_rows(const float *x, int64_t n, int d, float eps, float *out,
                                 float *norms, hsgk_stream_t stream);

/* ---- host helpers (no GPU work): the two small tables hsgk_segment_by_kmeans reads, with the
 * float32 bit patterns torch.linspace gives them in the reference ------------------------------
 * hsgk_host_grid_seed_map: initialize_cluster_labels (common.py:129-153) made dense (:341-342):
 *   seed_map[H*W] (HOST memory) and the number of seed clusters K.
 * hsgk_host_location_features: generate_location_features(.., 'float') - 0.5 (common.py:156-189,
 *   :313-316): loc[H*W*2] (HOST memory), (y, x) per pixel.  Copy both to the device.          */
HSGK_API int hsgk_host_grid_seed_map(int ky, int kx, int H, int W, int32_t *seed_map_host,
                                     int32_t *num_clusters);
HSGK_API int hsgk_host_location_features(int H, int W, float *loc_host);

/* ---- hsg/utils/segsort/common.py:270-408 segment_by_kmeans ---------------- */
typedef struct hsgk_segkm_args {
  /* inputs */
  const float *embeddings;     /* [B,C,H,W] f32 (NCHW, contiguous)            */
  const int64_t *labels;       /* [B,H,W] i64 or NULL (-> all zero)           */
  const float *loc;            /* location features, element (b,p,i) at
                                  loc[b*loc_batch_stride + p*2 + i]           */
  int64_t loc_batch_stride;    /* 0 when one [H,W,2] map is shared            */
  const int32_t *seed_map;     /* [H*W] i32: dense grid-seed label per pixel
                                  (common.py:320-323,341-345)                 */
  int32_t B, C, H, W;
  int32_t K;                   /* number of seed clusters (max seed + 1)      */
  int32_t iterations;
  int32_t has_ignore;          /* ignore_index is not None                    */
  int64_t ignore_index;
  int64_t batch_offset;        /* B * gpu_id (common.py:375-377)              */
  int64_t table_cap;           /* entries available for the relabel table     */
  /* outputs (sized for the worst case B*H*W rows; first meta->n_rows valid)  */
  float *out_embeddings;       /* [N,C]                                       */
  float *out_embeddings_loc;   /* [N,C+2]                                     */
  int64_t *out_labels;         /* [N]                                         */
  int64_t *out_cluster;        /* [N] dense segment ids                       */
  int64_t *out_batch;          /* [N]                                         */
  hsgk_segkm_meta *meta;       /* device                                      */
  /* optional state for the backward pass (NULL = not needed)                 */
  float *out_norms;            /* [N,2]: clamped ||x|| and ||(e,loc)|| per row */
  int64_t *out_rowmap;         /* [B*H*W]: output row of each pixel, -1 = dropped */
  /* scratch */
  void *workspace;
  size_t workspace_bytes;
  /* per-image seed maps (the `cluster_indices=` argument of the reference, common.py:320-323):
   * seed label of pixel p of image b = seed_map[b * seed_batch_stride + p]; 0 = one map for all   */
  int64_t seed_batch_stride;
  int32_t flags;               /* HSGK_SEGKM_* */
} hsgk_segkm_args;
/* small feature maps (training resolution) run the whole Lloyd loop of an image in one launch; maps of
 * more than 512 rows are shared by several workgroups that wait for each other: a plain launch, admitted only
 * within half of the device, at most two such grids in flight per device and process (a third waits in-stream);
 * HSGK_SMALL_COOP=1 uses hipLaunchCooperativeKernel instead.  A wait that still times out (10 s: another process
 * on the GPU) sets meta error 3.  ONE_GROUP keeps one workgroup per image (no inter-workgroup wait); maps beyond
 * that kernel's 1024 rows per image then take the per-kernel route.  hsgk_small_map_groups: the workgroups per image the
 * call would use on the current device (0: the shape does not take the one-launch route).          */
#define HSGK_SEGKM_ONE_GROUP 1
HSGK_API int hsgk_small_map_groups(int B, int C, int H, int W, int K);

HSGK_API size_t hsgk_segment_by_kmeans_workspace_bytes(int B, int C, int H, int W, int K,
                                              int64_t table_cap);
HSGK_API int hsgk_segment_by_kmeans(const hsgk_segkm_args *args, hsgk_stream_t stream);
/* Backward of the two float outputs w.r.t. the NCHW input (common.py:306-365:
 * normalise -> concat loc -> normalise -> index_select).  g_emb [N,C] and/or
 * g_emb_loc [N,C+2] may be NULL (= zero).  gx [B,C,H,W] is fully written
 * (dropped pixels get 0).  rowmap may be NULL when nothing was dropped.       */
HSGK_API int hsgk_segment_by_kmeans_bwd(const float *g_emb, const float *g_emb_loc,
                                        const float *emb, const float *emb_loc,
                                        const float *norms, const int64_t *rowmap, int B,
                                        int C, int H, int W, float eps, float *gx,
                                        hsgk_stream_t stream);

/* ---- hsg/utils/segsort/common.py:67-97 kmeans_with_initial_labels --------- */
/* One row set x[n,d]; labels_io holds the initial labels on entry (int64,
 * values in [0,K)) and the final labels on return.                            */
HSGK_API size_t hsgk_kmeans_workspace_bytes(int64_t n, int d, int K);
HSGK_API int hsgk_kmeans_with_initial_labels(const float *x, int64_t n, int d,
                                    int64_t *labels_io, int K, int iterations,
                                    void *workspace, size_t workspace_bytes,
                                    hsgk_stream_t stream);

/* ---- batch-level Lloyd half-steps (common.py:92 and :95) -------------------
 * x [B*rows_per_image, d]; labels int32 [B*rows_per_image] in [0,K);
 * centroids [B,K,d].  mstep = calculate_prototypes_from_labels per image,
 * estep = find_nearest_prototypes per image.                                  */
HSGK_API size_t hsgk_lloyd_workspace_bytes(int B, int64_t rows_per_image, int d, int K);
HSGK_API int hsgk_lloyd_mstep(const float *x, int B, int64_t rows_per_image, int d, int K,
                              const int32_t *labels, float *centroids, void *workspace,
                              size_t workspace_bytes, hsgk_stream_t stream);
/* M-step with EXACT segment sums (canonical order C2x: fixed point, quantum 2^-40, 64-bit
 * integer sums, one rounding to fp32), the arithmetic of the Lloyd loop inside
 * hsgk_segment_by_kmeans.  PRECONDITION: every element satisfies |x| <= 1 (true for the
 * L2-normalised rows this path is built for; 2^22 rows of such elements per segment stay
 * inside int64).  Larger magnitudes are not checked: the default kernels then only risk
 * int64 overflow of a sum beyond 2^23 in magnitude, but the opt-in matrix-core variant
 * (environment HSGK_MSTEP=mfma, an experiment kept for A/B) cuts the fixed
 * point value into digits that ASSUME |x| <= 1 and returns wrong sums otherwise.  sums [B,K,d] int64 is the
 * caller-owned state: labels_prev == NULL computes it from scratch for `labels`; otherwise
 * sums must hold the exact sums of labels_prev and is UPDATED from the rows whose label
 * differs (identical to a from-scratch pass over `labels`).  centroids [B,K,d].          */
HSGK_API int hsgk_lloyd_mstep_exact(const float *x, int B, int64_t rows_per_image, int d, int K,
                                    const int32_t *labels_prev, const int32_t *labels,
                                    int64_t *sums, float *centroids, void *workspace,
                                    size_t workspace_bytes, hsgk_stream_t stream);
/* unit_rows != 0 promises L2-normalised rows and centroids and enables the
 * filtered E-step (same labels, faster): 1 = bf16-split filter + exact re-score;
 * 2 = fp16 copy of the rows first (made inside the call; the composite makes it
 * once per segment_by_kmeans), then bf16-split on the undecided rows, then exact. */
HSGK_API int hsgk_lloyd_estep(const float *x, int B, int64_t rows_per_image, int d, int K,
                              const float *centroids, int32_t *labels_out, int unit_rows,
                              void *workspace, size_t workspace_bytes, hsgk_stream_t stream);

/* diagnostics of the last unit_rows E-step on this workspace: out[0] = rows re-scored
 * exactly, out[1] = rows the fp16 level (unit_rows = 2) handed to the bf16-split level
 * (only meaningful right after such a step).  out: int64[2] on the device.          */
HSGK_API int hsgk_lloyd_requeued_rows(int B, int64_t rows_per_image, int d, int K,
                                      void *workspace, size_t workspace_bytes, int64_t *out,
                                      hsgk_stream_t stream);

/* ---- hsg/utils/segsort/common.py:44-64 find_nearest_prototypes ------------ */
/* labels_out[n] = argmax_k <x_r, proto_k>, first index on ties.               */
HSGK_API size_t hsgk_assign_workspace_bytes(int64_t n, int d, int K);
HSGK_API int hsgk_find_nearest_prototypes(const float *x, int64_t n, int d,
                                 const float *prototypes, int K,
                                 int64_t *labels_out, void *workspace,
                                 size_t workspace_bytes, hsgk_stream_t stream);

/* ---- hsg/utils/segsort/common.py:11-41 calculate_prototypes_from_labels and
 *      hsg/utils/general/common.py:123-147 segment_mean -----------------------
 * x [n,d], labels int64 [n] (rows with labels outside [0,P) are skipped).
 * mode 0: L2-normalised segment sums; mode 1: means (count 0 -> 1); mode 2: raw
 * sums.  out [P,d]; aux [P] (nullable) receives the clamped norm (mode 0) or
 * the count (mode 1) for the backward pass.  Labels may be arbitrary: the
 * per-chunk partial sums are stored by the rank of an id among the chunk's distinct ids;
 * chunks with more than 512 distinct ids (or an id range beyond 32768) are summed by the
 * per-segment kernel instead (same order C2, slower).  *status (device int32): bit 1 is
 * set if a label lay outside [0,P) (such rows are skipped; the reference's scatter_add_
 * raises).                                                                      */
HSGK_API size_t hsgk_segment_reduce_workspace_bytes(int64_t n, int d, int64_t P);
HSGK_API int hsgk_segment_reduce(const float *x, int64_t n, int d, const int64_t *labels,
                                 int64_t P, int mode, float eps, float *out, float *aux,
                                 int32_t *status, void *workspace, size_t workspace_bytes,
                                 hsgk_stream_t stream);
/* gx [n,d] = d(loss)/d(x) given gout [P,d]; gseg [P,d] is scratch.             */
HSGK_API int hsgk_segment_reduce_bwd(const float *gout, const float *out, const float *aux,
                                     const int64_t *labels, int64_t n, int d, int64_t P,
                                     int mode, float eps, float *gseg, float *gx,
                                     hsgk_stream_t stream);

/* ---- hsg/utils/segsort/loss.py:15-82,149-190 SegSortLoss (and :85-130,193-251
 *      SetSegSortLoss), for up to three label sets in one pass -----------------------
 * hsg/models/predictions/hsg.py:78-155 evaluates the loss three times on the SAME
 * embeddings, instance labels and prototype table with different semantic labels
 * (image similarity, fine and coarse hierarchy); here E P^T is formed once.
 * emb [n,c] f32, inst int64 [n] (index of the pixel's own prototype), proto [P,c] f32.
 * Label set l (host array `sets`, L <= HSGK_LOSS_MAX_SETS): sem int64 [n], psem int64
 * [P], kappa = concentration, mode bit 0 = 'segsort+' (else 'segsort'), bit 1 = set mode:
 * sem / psem carry one bit per class of the multi-hot labels, 63 classes per int64 word,
 * W = mode >> 8 words per row (0 means 1; sem [n][W], psem [P][W]; W > 1 only with L = 1),
 * and "same semantic label" means a non-zero label affinity, i.e. the masks meet.
 * fwd writes, per set, the per-pixel negative log likelihood nll[L][n] and the backward
 * state num[L][n], den[L][n], use_same[L][n]; no [n,P] matrix exists.
 * bwd takes gscale[L][n] = dLoss/dnll and writes g_emb [n,c] and / or g_proto [P,c]
 * (either may be null): the score tiles are recomputed and contracted in place, memory
 * stays O(n c + P c).  c <= 384 for bwd.
 * pixel_group int64 [n] / proto_group int64 [P] (both null: none): prototype p takes part in pixel
 * i's sums only when proto_group[p] == pixel_group[i] -- the per-image prototype tables of
 * hsg/models/predictions/segsort.py:224-244 in ONE launch, and prototypes without a valid class
 * masked out instead of compacted (:181-196).  A pixel whose gscale is 0 contributes nothing to
 * either gradient (even if its own prototype lies outside its group).                          */
#define HSGK_LOSS_MAX_SETS 3
#define HSGK_LOSS_MASK_WORDS 4   /* set mode: <= 4 x 63 = 252 classes */
typedef struct hsgk_loss_set {
  const int64_t *sem;    /* [n] semantic label (or class mask) of every pixel       */
  const int64_t *psem;   /* [P] semantic label (or class mask) of every prototype   */
  float kappa;
  int32_t mode;
} hsgk_loss_set;
HSGK_API size_t hsgk_segsort_loss_workspace_bytes(int64_t n, int c, int64_t P, int L);
HSGK_API int hsgk_segsort_loss_fwd(const float *emb, int64_t n, int c, const int64_t *inst,
                                   const float *proto, int64_t P, int L,
                                   const hsgk_loss_set *sets, const int64_t *pixel_group,
                                   const int64_t *proto_group, float *nll, float *num, float *den,
                                   int32_t *use_same, void *workspace, size_t workspace_bytes,
                                   hsgk_stream_t stream);
HSGK_API size_t hsgk_segsort_loss_bwd_workspace_bytes(int64_t n, int c, int64_t P, int L);
HSGK_API int hsgk_segsort_loss_bwd(const float *emb, int64_t n, int c, const int64_t *inst,
                                   const float *proto, int64_t P, int L,
                                   const hsgk_loss_set *sets, const int64_t *pixel_group,
                                   const int64_t *proto_group, const float *num, const float *den,
                                   const int32_t *use_same, const float *gscale, float *g_emb,
                                   float *g_proto, void *workspace, size_t workspace_bytes,
                                   hsgk_stream_t stream);

/* ---- hsg/models/utils.py:127-217 gather_clustering_and_update_prototypes: the cross-GPU
 *      prototype exchange (called three times per training iteration, pyscripts/train/train.py:190,219)
 * Per-rank pixel rows in, batch-wide prototype tables out.  The reference ships every pixel
 * embedding to one GPU; here a rank reduces its pixels to per-segment sums and only tables cross
 * xGMI: ONE all_gather of the ranks' sorted distinct (batch, cluster, semantic, instance) tuples
 * (fixed capacity, row count in the block header) and ONE all_reduce(sum) of the zero-padded
 * [rows, C + D] sums, both RCCL calls on `stream` with the caller's ncclComm_t (`comm`, void*).
 * No host read anywhere: counts and data-dependent errors come back in the device `meta` block.
 *   embeddings [n,C], embeddings_loc [n,D] f32; cluster / batch / semantic / instance int64 [n], >= 0.
 *   Dense id of a pixel = rank of its tuple among the distinct tuples of ALL ranks in lexicographic
 *   order (the reference's two nested sorted `unique`s, utils.py:181-193) -> updated_cluster [n].
 *   prototypes [rows,C], prototypes_loc [rows,D] = L2-normalised segment sums (order C2 per rank,
 *   eps-clamped norm); norms [rows][2] (nullable) keeps the clamped norms for the backward pass;
 *   proto_semantic / proto_instance / proto_batch [cap_total] decoded from the tuples (utils.py:193-197).
 *   table [cap_total][C+D]: the raw sums (caller-owned so that it outlives the workspace between
 *   begin and finish).
 *   cap_local = tuple rows a rank may contribute, cap_total = rows of the tables; pool_rows = partial
 *   rows of the chunk sums (sum over the chunks of 2048 rows of their distinct ids; chunks that find
 *   the pool exhausted are summed by the slower per-segment scan, same order).
 *   meta (device, int64[8]): [0] distinct local tuples, [1] distinct tuples over all ranks (= valid
 *   table rows), [2] error bits: 1 negative component, 2 more tuples than cap_local (on any rank),
 *   4 packed key overflows 2^62, 8 more rows than cap_total; [3] the largest tuple count of any rank
 *   (what cap_local has to hold); [4] / [5] (one rank, tables up to 64 K rows; else -1): the number of distinct
 *   batch values among the table rows and the most rows one of them has -- the shape of the per-image padded
 *   prototype tables of resnet_fcn_hsg.py:499-502.  [0..5] are final once the merge stage has run (keys and merge
 *   set every error bit), so a host may fetch the block BEFORE the sums stage and let the sums run behind its
 *   read.  With an error the outputs are undefined.
 * hsgk_exchange_prototypes = hsgk_exchange_begin + hsgk_exchange_finish(rows = cap_total): fully
 * asynchronous.  A caller that needs the row count on the host anyway (to shape tensors) calls
 * begin, reads meta, then finish with rows = meta[1] (the all_reduce then moves only the used rows).
 * The phases are also exported for ONE process driving several GPUs (the reference's
 * DataParallel mode: lists of per-GPU tensors): hsgk_exchange_keys on every device, the send
 * blocks copied into the recv blocks of the anchor, hsgk_exchange_merge(my_rank = -1) there (slots
 * for every source), hsgk_exchange_sums on every device, the tables added up on the anchor,
 * hsgk_exchange_finish(comm = NULL).                                                            */
typedef struct hsgk_exchange_args {
  const float *embeddings;       /* [n,C]  */
  const float *embeddings_loc;   /* [n,D]  */
  const int64_t *cluster, *batch, *semantic, *instance;   /* [n] */
  int64_t n;
  int32_t C, D;
  int64_t cap_local, cap_total, pool_rows;
  float eps;
  float *table;                  /* [cap_total][C+D] */
  float *prototypes;             /* [rows][C] */
  float *prototypes_loc;         /* [rows][D] */
  float *norms;                  /* [rows][2] or NULL */
  int64_t *proto_semantic, *proto_instance, *proto_batch;   /* [cap_total] */
  int64_t *updated_cluster;      /* [n] */
  int64_t *meta;                 /* device int64[8] */
  void *workspace;
  size_t workspace_bytes;
} hsgk_exchange_args;
HSGK_API size_t hsgk_exchange_workspace_bytes(int64_t n, int C, int D, int64_t cap_local,
                                              int64_t cap_total, int world, int64_t pool_rows);
HSGK_API int hsgk_exchange_prototypes(const hsgk_exchange_args *args, void *comm, int rank, int world,
                                      hsgk_stream_t stream);
HSGK_API int hsgk_exchange_begin(const hsgk_exchange_args *args, void *comm, int rank, int world,
                                 hsgk_stream_t stream);
HSGK_API int hsgk_exchange_finish(const hsgk_exchange_args *args, int64_t rows, void *comm, int world,
                                  hsgk_stream_t stream);
/* phases (see above).  send block = int64[8 + 4 cap_local]: [0] row count, [1] error bits, then the
 * sorted tuples; recv blocks = `world` such blocks in rank order (both inside the workspace).
 * merge: slots_out int32 [world][cap_local] (NULL: kept in the workspace) = table row of every
 * source's tuple; sums: `slots` = this rank's row of that array (NULL: the workspace's).         */
HSGK_API int hsgk_exchange_keys(const hsgk_exchange_args *args, int world, hsgk_stream_t stream);
HSGK_API const int64_t *hsgk_exchange_send_block(const hsgk_exchange_args *args, int world, size_t *bytes);
HSGK_API int64_t *hsgk_exchange_recv_blocks(const hsgk_exchange_args *args, int world);
HSGK_API int hsgk_exchange_merge(const hsgk_exchange_args *args, int my_rank, int world,
                                 int32_t *slots_out, hsgk_stream_t stream);
HSGK_API int hsgk_exchange_sums(const hsgk_exchange_args *args, int my_rank, int world,
                                const int32_t *slots, hsgk_stream_t stream);
/* Communicator helpers (host): librccl is bound at run time -- the instance the process already
 * loaded (torch's), else the system one -- so libhsgk.so has no link-time dependency on it.
 * unique_id on rank 0, hand the bytes to every rank (any transport), init_rank on every rank with
 * its device current.  all_reduce_f32 (in place) is the gradient's way back through the exchange. */
#define HSGK_COMM_ID_BYTES 128
HSGK_API int hsgk_comm_unique_id(void *id_out, size_t bytes);
HSGK_API int hsgk_comm_init_rank(void **comm_out, int world, int rank, const void *id, size_t bytes);
HSGK_API int hsgk_comm_destroy(void *comm);
HSGK_API int hsgk_comm_all_reduce_f32(float *buf, int64_t count, void *comm, hsgk_stream_t stream);
HSGK_API int hsgk_comm_all_gather_bytes(const void *send, void *recv, size_t bytes_per_rank, void *comm,
                                        hsgk_stream_t stream);

/* ---- hsg/models/embeddings/resnet_fcn_hsg.py:638-672 _hierarchical_grouping tail
 * fine_logits [B,KF,N], coarse_logits [B,KC,KF] (nullable).  fine_prob = softmax
 * over KF, fine_lab = argmax; coarse_prob [B,KC,N] = softmax_KC(coarse) x fine_prob,
 * coarse_lab = argmax.                                                          */
HSGK_API int hsgk_hier_assign(const float *fine_logits, const float *coarse_logits, int B, int KF,
                              int KC, int N, float *fine_prob, int64_t *fine_lab,
                              float *coarse_prob, int64_t *coarse_lab, hsgk_stream_t stream);
/* backward of the two probability outputs: g_fine_prob [B,KF,N] (nullable = zero), g_coarse_prob [B,KC,N]
 * (nullable: no coarse branch in the gradient) -> g_fine_logits [B,KF,N], g_coarse_logits [B,KC,KF]          */
HSGK_API int hsgk_hier_assign_bwd(const float *fine_logits, const float *coarse_logits, int B, int KF,
                                  int KC, int N, const float *g_fine_prob, const float *g_coarse_prob,
                                  float *g_fine_logits, float *g_coarse_logits, hsgk_stream_t stream);
/* ---- resnet_fcn_hsg.py:683-748 _collect_nd_coarser_prototype ------------------
 * protos [B,C,N], labels int64 [B,N], masks uint8 [B,N] (nullable) -> out [B,C,G]:
 * mean of the unpadded nodes of every group, optionally L2-normalised over C.   */
HSGK_API int hsgk_group_mean(const float *protos, const int64_t *labels, const uint8_t *masks,
                             int B, int C, int N, int G, int normalized, float eps, float *out,
                             hsgk_stream_t stream);
/* backward of hsgk_group_mean (what autograd derives from :706-746): g_out [B,C,G] -> g_protos [B,C,N]; padded
 * nodes and nodes whose label lies outside [0, G) receive zero.                                              */
HSGK_API int hsgk_group_mean_bwd(const float *protos, const int64_t *labels, const uint8_t *masks,
                                 int B, int C, int N, int G, int normalized, float eps,
                                 const float *g_out, float *g_protos, hsgk_stream_t stream);
/* ---- resnet_fcn_hsg.py:751-780 pixel -> segment -> group label ---------------
 * out[i] = table[img[i] * M + seg[i]]                                           */
HSGK_API int hsgk_gather_labels(const int64_t *table, int M, const int64_t *img,
                                const int64_t *seg, int64_t n, int64_t *out, hsgk_stream_t stream);

/* ---- resnet_fcn_hsg.py:499-577 / :1061-1136 the padded per-image tables of _calculate_kmeans_prototypes
 * The P segments arrive sorted by image (seg_image int64 [P], ascending: the order of the tuple kernels of the
 * exchange); segment s takes the slot (dense image number, rank inside its image) of the [B, M] tables:
 * table [B*M, C] <- protos [P, C], pos_table [B*M, Cp] <- pos [P, Cp] (both nullable together), masks uint8 [B*M]
 * (1 = padding), plabs / pbatch int64 [B*M] <- seg_lab / seg_batch [P] (-1 = padding); per pixel (pixel_seg int64 [n]
 * = its segment): by_image = the segment's rank inside its image, pixel_image = its dense image number.
 * B = distinct images and M >= the most segments of one are the caller's (hsgk_exchange meta [4], [5]).
 * seg_slot int64 [P] (nullable): every segment's table row (what a backward pass gathers).  work: int32 [2 P + B + 1]. */
HSGK_API int hsgk_pad_prototype_tables(const int64_t *seg_image, int64_t P, const float *protos, int C,
                                       const float *pos, int Cp, const int64_t *seg_lab, const int64_t *seg_batch,
                                       const int64_t *pixel_seg, int64_t n, int B, int M, float *table,
                                       float *pos_table, uint8_t *masks, int64_t *plabs, int64_t *pbatch,
                                       int64_t *by_image, int64_t *pixel_image, int64_t *seg_slot, int32_t *work,
                                       hsgk_stream_t stream);

/* ---- hsg/models/embeddings/transformer_clusters.py:99-114 TransformerClustering tail
 * centroids / centroid_feats [B,C,tl], node_features [B,C,sl] (the reference's
 * channel-major layouts).  logits_all [B,tl,sl] = cent^T feat / sqrt(C); order [B,k] =
 * the k query rows with the largest row maximum, descending (lower index first on
 * ties); logits_sel [B,k,sl], centroids_sel / centroid_feats_sel [B,C,k] gathered in
 * that order.  1 <= k <= tl.                                                       */
HSGK_API int hsgk_cluster_topk(const float *centroids, const float *centroid_feats,
                               const float *node_features, int B, int C, int tl, int sl, int k,
                               float *logits_all, int64_t *order, float *logits_sel,
                               float *centroids_sel, float *centroid_feats_sel,
                               hsgk_stream_t stream);

/* ---- hsg/utils/segsort/eval.py:9-52 top_k_ranking (retrieval contraction) ----
 * queries [n,c], proto [P,c] -> out_idx [n,topk] (int64 prototype indices by
 * descending <q,p>, lower index first on exact ties) and out_val [n,topk].
 * 1 <= topk <= min(32, P).                                                      */
HSGK_API size_t hsgk_topk_workspace_bytes(int64_t n, int c, int64_t P, int topk);
HSGK_API int hsgk_topk_prototypes(const float *queries, int64_t n, int c, const float *proto,
                                  int64_t P, int topk, int64_t *out_idx, float *out_val,
                                  void *workspace, size_t workspace_bytes, hsgk_stream_t stream);
/* The same with groups (hsg/models/utils.py:243-309, nearest labelled segments of the SAME image):
 * only prototypes with proto_group[p] == query_group[i] compete for query i; slots that no such
 * prototype fills get index 0 and value -inf.  Both vectors int64, both null = no grouping.      */
HSGK_API int hsgk_topk_prototypes_grouped(const float *queries, int64_t n, int c, const float *proto,
                                          int64_t P, int topk, const int64_t *query_group,
                                          const int64_t *proto_group, int64_t *out_idx,
                                          float *out_val, void *workspace, size_t workspace_bytes,
                                          hsgk_stream_t stream);

/* ---- full-resolution inference around k-means ---------------------------------
 * pyscripts/inference/prototype.py:141-177 (inference.py:165-196): crop [C,h,w] (NCHW
 * plane of one crop) is L2-normalised per pixel (general/common.py:101-120, eps 1e-12)
 * and added into canvas [C,H,W] at (sh, sw); counts [H,W] += 1.  finish: canvas /=
 * counts.  Crops are accumulated in call order (same float sums as the reference).    */
HSGK_API int hsgk_overlap_accumulate(const float *crop, int C, int h, int w, float *canvas,
                                     float *counts, int H, int W, int sh, int sw, float eps,
                                     hsgk_stream_t stream);
HSGK_API int hsgk_overlap_finish(float *canvas, const float *counts, int C, int H, int W,
                                 hsgk_stream_t stream);
/* ---- hsg/utils/segsort/common.py:221-268 find_majority_label_index -------------
 * hist [num_clusters,num_classes] int32 (workspace, overwritten), majority [num_clusters]
 * = first maximal class per cluster, select [n] = 1 where the pixel's class is its
 * cluster's majority class.  Labels must lie in [0,num_classes) / [0,num_clusters).  */
HSGK_API int hsgk_majority_labels(const int64_t *semantic, const int64_t *cluster, int64_t n,
                                  int64_t num_clusters, int num_classes, int32_t *hist,
                                  int64_t *majority, uint8_t *select, hsgk_stream_t stream);

/* ---- hsg/utils/graph/common.py:39-125 affinity_matrix_as_attention -------------
 * x [B,C,N]: A = exp(concentration * x^T x) (common.py:23-36) is computed into
 * affinity_tmp [B,N,N] unless affinity_in [B,N,N] is given (a caller-evaluated kernel
 * function).  Then, per image: entries of padded nodes (padding_mask [B,N], nullable)
 * -> 0; the diagonal -> 0 when remove_self_loop and the image has more than one valid
 * node; knn > 0: within every segment (segment_labels [B,N], nullable = one segment)
 * each row keeps the entries that are not below its k-th largest one of that segment,
 * k = min(valid nodes of the segment, knn); binarize: out = (A > 0).  out [B,N,N].    */
HSGK_API int hsgk_knn_affinity(const float *x, const float *affinity_in, int B, int C, int N,
                               float concentration, const uint8_t *padding_mask,
                               const int64_t *segment_labels, int knn, int remove_self_loop,
                               int binarize, float *affinity_tmp, float *out,
                               hsgk_stream_t stream);

/* ---- hsg/utils/graph/loss.py:27-96 dmon_pool_loss, for an adjacency that carries no gradient (the binary
 * k-NN graph of DMonLoss, loss.py:99-145) -------------------------------------------------------------------
 * adj [B,N,N], s [B,N,K] cluster assignments (K <= 32), valid [B,N] (nullable; rows with 0 count as zero rows).
 * Per image: t = (Tr(S^T A S) - |S^T d|^2 / 2m) / 2m with d = A 1, 2m = 2 sum(d), and c = |sum_i S_i|_2 -- so
 * that dmon_loss = mean(1 - t), collapse_loss = mean(c) * sqrt(K) / N as the reference defines them.  `saved`
 * (hsgk_dmon_pool_workspace_bytes) keeps A S, d and the per-image sums for the backward:
 * grad_s = g_t[b] * dt/dS + g_c[b] * dc/dS.                                                                   */
HSGK_API size_t hsgk_dmon_pool_workspace_bytes(int B, int N, int K);
HSGK_API int hsgk_dmon_pool_fwd(const float *adj, const float *s, const uint8_t *valid, int B, int N, int K,
                                float *t_out, float *c_out, void *saved, size_t saved_bytes,
                                hsgk_stream_t stream);
HSGK_API int hsgk_dmon_pool_bwd(const float *adj, const float *s, const uint8_t *valid, int B, int N, int K,
                                const void *saved, const float *g_t, const float *g_c, float *grad_s,
                                hsgk_stream_t stream);

/* ---- synthetic inputs of the benchmark (hsg_amd/utils/synth.py; no reference counterpart:
 * the reference ships no benchmark, BASELINE.md section 2 defines the generator) ------------
 * out[i] = gaussish(hash(key, offset + i)); key = synth.stream_key(seed).  Bit-identical to
 * the numpy generator that made the golden fixtures.                                        */
HSGK_API int hsgk_synth_gaussish(uint64_t key, uint64_t offset, int64_t n, float *out,
                                 hsgk_stream_t stream);
/* 'mixture' flavour: out[b][c][y][x] = centres[b][blob(first_image + b, y, x)][c] + 0.05f * noise;
 * images first_image .. first_image + B of the stream, centres [B][ncentres][C] of those images  */
HSGK_API int hsgk_synth_mixture(uint64_t noise_key, uint64_t seed, const float *centres,
                                int ncentres, int first_image, int B, int C, int H, int W,
                                float *out, hsgk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HSGK_H_ */
