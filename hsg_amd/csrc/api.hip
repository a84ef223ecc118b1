// api.hip -- extern "C" surface of libhsgk.so (include/hsgk.h) and the host
// orchestration of segment_by_kmeans.  Host code only enqueues kernels on the
// caller's stream; it never allocates or synchronises.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace hsgk {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- optional event profiling of the kernel groups ------------------------
// Records reference events by index: consecutive scopes of one Lloyd loop share their boundary event (ProfChain), so a
// ten-iteration call records 31 loop events instead of 60 -- every event is a packet the queue drains before the next
// kernel starts, ~2.5 us each (cfg3: 0.25 ms of a 2.6 ms call with the scopes on, measured).
struct ProfRec { int kind; int ia, ib; };
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_events;
static std::atomic<int> g_prof_on{0};
static std::atomic<uint64_t> g_prof_epoch{0};          // bumped by every collect (which destroys the events)

struct ProfChain { int last = -1; hipStream_t s = nullptr; uint64_t epoch = 0; };

static int prof_new_event(hipStream_t s) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return -1;
  (void)hipEventRecord(e, s);
  std::lock_guard<std::mutex> g(g_prof_mu);
  g_prof_events.push_back(e);
  return (int)g_prof_events.size() - 1;
}

struct ProfScope {
  int kind; hipStream_t s; int ia = -1; bool on; ProfChain *chain; uint64_t epoch;
  // chain: nothing is launched on `s` between the previous scope of the chain and this one (the caller's promise)
  ProfScope(int k, hipStream_t st, ProfChain *ch = nullptr)
      : kind(k), s(st), on(g_prof_on.load() != 0), chain(ch), epoch(g_prof_epoch.load()) {
    if (!on) return;
    if (chain && chain->last >= 0 && chain->s == s && chain->epoch == epoch) ia = chain->last;
    else ia = prof_new_event(s);
    if (ia < 0) on = false;
  }
  ~ProfScope() {
    if (!on) return;
    const int ib = epoch == g_prof_epoch.load() ? prof_new_event(s) : -1;
    if (chain) { chain->last = ib; chain->s = s; chain->epoch = epoch; }
    if (ib < 0) return;
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (epoch == g_prof_epoch.load()) g_prof.push_back({kind, ia, ib});
  }
};

// ---- optional verification of the filtered E-steps --------------------------
// While enabled, every filtered E-step of lloyd() is followed by the exact fp32-MFMA
// E-step on the same rows and centroids (into a scratch label buffer) and a compare
// pass: g_verify[0] counts the rows compared, g_verify[1] the rows whose labels differ.
// The filters only ever label rows whose exact argmax is provably unique, so the second
// counter must stay 0 (tests run every BASELINE config under this switch).
static std::atomic<int> g_verify_on{0};
__device__ unsigned long long g_verify[2];

__global__ void verify_compare_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b,
                                      const hsgk_segkm_meta *__restrict__ meta, int64_t cap) {
  const int64_t n = meta->n_rows < cap ? meta->n_rows : cap;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    bad += get_label(a, i) != get_label(b, i);
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&g_verify[1], bad);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_verify[0], (unsigned long long)n);
}

// byte labels (common.h: labels_as_u8) -> int32, for the consumers outside the Lloyd loop
__global__ void labels_u8_to_i32_kernel(const uint8_t *__restrict__ in, int64_t n, int32_t *__restrict__ out) {
  const int64_t nq = n >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t v = reinterpret_cast<const uint32_t *>(in)[q];
    reinterpret_cast<int4 *>(out)[q] = make_int4((int)(v & 255u), (int)((v >> 8) & 255u), (int)((v >> 16) & 255u), (int)(v >> 24));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(nq << 2) + threadIdx.x] = in[(nq << 2) + threadIdx.x];
}

// HSGK_LABELS=i32 keeps int32 working labels inside the Lloyd loop (read per call, for the tests)
static bool label_u8_enabled() {
  const char *e = getenv("HSGK_LABELS");
  return !(e && e[0] == 'i');
}

struct KmeansScratch {
  ChunkTable t;
  int32_t *klab;
  float *best;
  float *partial;
  unsigned char *pmask;    // [max_chunks][K] which partial rows exist
  float *cent;
  void *qrows;         // [max_chunks * HSGK_CHUNK] 12-byte entries: exact re-score queue (split E-step)
  int32_t *qcount;     // [1] queue length
  int32_t *klab_prev;  // [rows] labels the exact sums currently hold (-1 = row not added yet), or null
  long long *sumq;     // [B][K][d] exact fixed-point segment sums (sums_fx.hip)
  _Float16 *xhT;       // the same copy in tile order (score_tiles_f16t.h), or null
  _Float16 *xh;        // [rows][half_main_cols(d)] fp16 copy of the rows' main columns (first filter level), or null
  void *state;         // [rows] 16-byte records of the two-half filter (128 < K <= 256)
  float *errc;         // [B][K] measured fp16 rounding error of the centroid rows (wide filter)
  uint2 *xt;           // [rows] {packed fp16 tail columns of the copy, measured rounding error of the row}
  int32_t *q1;         // [B][q1cap] rows the first level left undecided
  int32_t *q1count;    // [B]
  float *cent_multi;   // per-workgroup centroid copies of the multi-workgroup small-map route (or null)
  int64_t q1cap;
  size_t rows_cap;     // B * rows_per_image
  int max_chunks;
  PrepM0 m0;           // partial sums of the first M-step, written by the prep kernel (or part == null)
  int m0_wt;           // prep workgroups per image
};

// HSGK_TLAYOUT=1: the first E-step level reads a tile-ordered fp16 copy wherever an engine for it exists (K <= 64
// and 128 < K <= 256 at 256 main columns), converted from the row-major copy when the prep kernel cannot write it
// itself (+0.5 ms per call at 4 x 768^2).  Default (unset): only 128 < K <= 256, and only when the prep kernel
// writes the tile order directly (no label compaction, H * W % 32 == 0) -- filter 0.49 -> 0.42 ms per launch with
// no conversion.  HSGK_TLAYOUT=0: never.
static int tlayout_mode() {
  const char *e = getenv("HSGK_TLAYOUT");
  return !e ? -1 : e[0] == '1' ? 1 : 0;
}
static bool tlayout_env() { return tlayout_mode() == 1; }

static int max_chunks_for(int B, int64_t rows_per_img) {
  return (int)(B * ((rows_per_img + HSGK_CHUNK - 1) / HSGK_CHUNK));
}

static void carve_kmeans(Carver &cv, int B, int64_t rows_per_img, int d, int K,
                         KmeansScratch *k) {
  const int mc = max_chunks_for(B, rows_per_img);
  const size_t mcs = mc > 0 ? mc : 1;
  k->max_chunks = mc;
  k->rows_cap = (size_t)B * rows_per_img;
  k->t.img_row0 = cv.take<int64_t>(B + 1);
  k->t.img_chunk0 = cv.take<int32_t>(B + 1);
  k->t.chunk_row0 = cv.take<int64_t>(mcs);
  k->t.chunk_rows = cv.take<int32_t>(mcs);
  k->t.chunk_img = cv.take<int32_t>(mcs);
  k->klab = cv.take<int32_t>((size_t)B * rows_per_img + 1);
  k->best = cv.take<float>((size_t)B * rows_per_img + 1);
  k->partial = cv.take<float>(mcs * K * d);
  k->pmask = cv.take<unsigned char>(mcs * K + 4);
  k->cent = cv.take<float>((size_t)B * K * d + 1);
  k->qrows = cv.take<char>(mcs * HSGK_CHUNK * 12);
  k->qcount = cv.take<int32_t>(4);
  k->klab_prev = nullptr;
  k->sumq = nullptr;
  // exact sums: the update kernels address clusters with 10-bit fields (K <= 1023) and a 2^40
  // fixed-point sum of unit rows stays inside int64 for <= 2^22 rows per segment; beyond either
  // bound the Lloyd loop keeps the streaming M-step of order C2
  if (sums_fx_eligible(d) && K <= 1023 && rows_per_img <= (int64_t)1 << 22) {
    k->klab_prev = cv.take<int32_t>((size_t)B * rows_per_img + 1);
    k->sumq = cv.take<long long>((size_t)B * K * d + 1);
  }
  k->m0 = PrepM0{nullptr, nullptr, nullptr, 0};
  k->m0_wt = 0;
  k->xh = nullptr;
  k->xhT = nullptr;
  k->xt = nullptr;
  k->q1 = k->q1count = nullptr;
  k->q1cap = rows_per_img;
  k->errc = nullptr;
  k->state = nullptr;
  if (assign_half_wide2_eligible(d, K)) k->state = cv.take<char>(((size_t)B * rows_per_img + kHalfSlackRowsHost) * 16);
  if (assign_half_eligible(d, K) || assign_half_wide_eligible(d, K) || assign_half_wide2_eligible(d, K)) {
    k->errc = cv.take<float>((size_t)B * K + 1);
    // + slack rows: the fp16 engine reads past the end of a pass instead of clamping
    k->xh = cv.take<_Float16>(((size_t)B * rows_per_img + kHalfSlackRowsHost) * half_main_cols_host(d) + 8);
    k->xt = cv.take<uint2>((size_t)B * rows_per_img + kHalfSlackRowsHost);
    if (((tlayout_env() && assign_half_eligible(d, K)) || (tlayout_mode() != 0 && assign_half_wide2_eligible(d, K))) &&
        d / 64 == 4 && rows_per_img % 32 == 0)
      k->xhT = cv.take<_Float16>(((size_t)B * rows_per_img + kHalfSlackRowsHost) * half_main_cols_host(d) + 8);
    k->q1 = cv.take<int32_t>((size_t)B * rows_per_img + 1);
    // K <= 64: [B] lengths of the fp16 level's row queues; wider filters: the all-K row lists' lengths, one 128-byte
    // line per image (kmeans.hip: kHardStride), followed by the previous iteration's
    k->q1count = cv.take<int32_t>(2 * ((size_t)B + 1) * 32);
  }
  k->cent_multi = nullptr;
  if (k->sumq && assign_half_eligible(d, K) && rows_per_img <= 16 * 1024)       // small maps only (lloyd_small_groups)
    k->cent_multi = cv.take<float>(lloyd_small_cent_floats(d, K, B) + 1);
}

// segment_by_kmeans only: room for the first M-step's partial sums (prep.hip), when the
// 32-pixel prep kernel and the exact sums both apply
static void carve_m0(Carver &cv, int B, int C, int ntiles, int K, KmeansScratch *k) {
  // C <= 256 only: wider rows need further column passes with per-row atomics and an LDS footprint
  // that costs the prep kernel a workgroup per CU (C = 384: prep 3.4 ms with, 1.3 ms without)
  if (k->sumq == nullptr || (C % 64) != 0 || C > 256) return;
  k->m0_wt = 2 * ntiles;
  k->m0.part = cv.take<unsigned long long>((size_t)B * k->m0_wt * 2 * (C + 2));
  k->m0.lab = cv.take<int32_t>((size_t)B * k->m0_wt * 2);
  k->m0.sumq = reinterpret_cast<unsigned long long *>(k->sumq);
  k->m0.K = K;
}

static bool fx_enabled() {            // HSGK_MSTEP=stream keeps the streaming C2 M-step
  static const bool on = [] {
    const char *e = getenv("HSGK_MSTEP");
    return !(e && e[0] == 's');
  }();
  return on;
}

// HSGK_ASSIGN = "fp32": exact kernel only; "split": bf16x3 filter + exact; default: fp16 filter first
static int assign_mode() {
  static const int mode = [] {
    const char *e = getenv("HSGK_ASSIGN");
    return !e ? 2 : e[0] == 'f' ? 0 : e[0] == 's' ? 1 : 2;
  }();
  return mode;
}

// unit_rows: rows of x and centroids are L2-normalised (enables the filtered E-step:
// fp16 copy -> bf16x3 -> exact; results are identical either way).
// half_ready: the producer of x (the prep kernel) already wrote the fp16 copy k.xh;
// otherwise a separate pass makes it when the iteration count pays for it.
static int lloyd(const float *x, int d, int K, int B, int iterations,
                 const KmeansScratch &k, const hsgk_segkm_meta *meta, hipStream_t s,
                 bool unit_rows = false, bool half_ready = false, bool m0_ready = false,
                 bool single_group = false, const int32_t **final_labels = nullptr, bool tiles_ready = false) {
  const bool half_any = unit_rows && assign_mode() == 2 && k.xh && (half_ready || iterations >= 3);
  const bool half = half_any && assign_half_eligible(d, K);
  const bool wide = half_any && !half && assign_half_wide_eligible(d, K);
  const bool wide2 = half_any && !half && !wide && assign_half_wide2_eligible(d, K);
  if ((half || wide || wide2) && !half_ready) {
    ProfScope p(HSGK_PROF_PREP, s);
    if (int rc = launch_to_half_rows(x, k.t, k.max_chunks, d, k.xh, k.xt, meta, s)) return rc;
  }
  // exact (fixed-point) segment sums, updated from the rows whose label changed: unit rows
  // only (|x| <= 1 bounds the integer sums).  m0_ready: the prep kernel already summed the
  // rows under their seed labels (partials in k.m0, sumq zeroed before it ran).
  const bool fx = unit_rows && fx_enabled() && k.sumq != nullptr && k.max_chunks > 0;
  // Small feature maps (training resolution): one workgroup per image runs the whole loop in one
  // launch (kmeans.hip: lloyd_small_kernel).  HSGK_SMALL = 0 / 1 forces the per-kernel / the fused
  // route (read per call, so that the tests cover both); the verify switch keeps the per-kernel route.
  {
    const char *se = getenv("HSGK_SMALL");
    const int64_t rows_per_image = B > 0 ? (int64_t)(k.rows_cap / (size_t)B) : 0;
    // (single_group: one workgroup holds at most kSmallRowsMax rows -- larger maps take the per-kernel route)
    const bool can = fx && half && iterations >= 1 && k.q1 && lloyd_small_eligible(d, K, B, rows_per_image) &&
                     (!single_group || rows_per_image <= lloyd_small_rows_max()) && !g_verify_on.load();
    const bool want = se ? se[0] == '1' : true;
    if (can && want) {
      if (m0_ready)
        if (int rc = launch_m0_reduce(k.m0, B, k.m0_wt, d, s)) return rc;
      ProfScope p(HSGK_PROF_ASSIGN, s);
      if (final_labels) *final_labels = k.klab;
      // (k.q1, the first level's row queue of the per-kernel route, holds the fused kernel's counters)
      return launch_lloyd_small(x, k.xh, k.xt, d, K, B, iterations, k.t, k.klab, k.klab_prev, k.sumq, k.cent,
                                k.qrows, k.q1, m0_ready, const_cast<hsgk_segkm_meta *>(meta), rows_per_image, single_group, k.cent_multi, s);
    }
  }
  if (fx && !m0_ready) {     // (the fused kernel above needs neither)
    HSGK_CHECK_HIP(hipMemsetAsync(k.klab_prev, 0xFF, sizeof(int32_t) * k.rows_cap, s));
    HSGK_CHECK_HIP(hipMemsetAsync(k.sumq, 0, sizeof(long long) * (size_t)B * K * d, s));
  }
  // The working labels ping-pong between k.klab and k.klab_prev (every E-step rewrites all
  // rows): after the sums are brought up to date with `cur`, that buffer becomes `prev` and the
  // E-step writes the other one -- no label copy per iteration.
  int32_t *cur = k.klab, *prev = k.klab_prev;
  // Inside the loop the E-step levels write ONE BYTE per label (common.h: labels_as_u8; the buffers keep their
  // int32 size, the bytes use the front quarter): a quarter of the isolated line writes.  The seed labels arrive
  // as int32 (prep / the caller); the sums update reads either format, tagged per buffer.
  const bool u8 = label_u8_enabled() && fx && K <= 256 && (half || wide || wide2);
  const _Float16 *xhT = nullptr;
  if ((half || wide2) && k.xhT && !single_group && (tiles_ready || tlayout_env())) {
    // (experiment) every image holds rows_cap / B rows only when nothing was compacted away: the caller of the
    // experiment guarantees it
    if (!tiles_ready)
      if (int rc = launch_rows_to_tiles(k.xh, d, (int64_t)k.rows_cap, k.xhT, s)) return rc;
    xhT = k.xhT;
  }
  ProfChain chain;           // the loop's scopes follow one another with nothing launched in between
  for (int it = 0; it < iterations; ++it) {
    bool counters_zeroed = false;
    if (fx) {
      { ProfScope p(HSGK_PROF_ACCUMULATE, s, &chain);
        if (it == 0 && m0_ready) {
          if (int rc = launch_m0_reduce(k.m0, B, k.m0_wt, d, s)) return rc;
        } else if (int rc = launch_update_sums(x, d, prev, cur, k.t, k.max_chunks, K, k.sumq, meta, s, unit_rows ? d - 2 : 0))
          return rc;
        std::swap(cur, prev); }
      { ProfScope p(HSGK_PROF_FINALIZE, s, &chain);
        // (the queue counters of the E-step that follows, and for the hi-plane filters of K > 64 the table's fp16
        //  rounding errors, come out of this launch instead of memsets / a kernel of their own per iteration)
        counters_zeroed = half || ((wide || wide2) && k.errc);
        // (k.q1 / k.q1count: the fp16 level's row queues when K <= 64, the all-K row lists of the wider filters)
        if (int rc = launch_finalize_fx(k.sumq, d, K, B, HSGK_EPS, k.cent, s,
                                        (half || counters_zeroed) ? k.q1count : nullptr, half ? B : B * 32,
                                        counters_zeroed ? k.qcount : nullptr,
                                        (wide || wide2) ? k.errc : nullptr,
                                        (!half && counters_zeroed) ? k.q1count + (size_t)(B + 1) * 32 : nullptr, it > 0)) return rc; }
    } else {
      { ProfScope p(HSGK_PROF_ACCUMULATE, s);
        if (int rc = launch_accumulate(x, d, cur, k.t, k.max_chunks, K, k.partial, k.pmask, meta, s)) return rc; }
      { ProfScope p(HSGK_PROF_FINALIZE, s);
        if (int rc = launch_finalize(k.partial, k.pmask, d, K, B, k.t, k.max_chunks / B, HSGK_EPS, k.cent, s)) return rc; }
    }
    if (u8) cur = labels_as_u8(labels_base(cur));          // this iteration's labels: bytes in this buffer
    { ProfScope p(HSGK_PROF_ASSIGN, s, &chain);
      if (int rc = half ? launch_assign_half(x, k.xh, k.xt, d, k.cent, K, B, k.t, k.max_chunks, cur, k.q1,
                                             k.q1count, k.q1cap, k.qrows, k.qcount, meta, s, counters_zeroed, xhT)
               : wide ? launch_assign_half_wide(x, k.xh, k.xt, d, k.cent, k.errc, K, B, k.t, k.max_chunks, cur,
                                                k.qrows, k.qcount, meta, s, counters_zeroed, k.q1, k.q1count, k.q1cap)
               : wide2 ? launch_assign_half_wide2(x, k.xh, k.xt, d, k.cent, k.errc, K, B, k.t, k.max_chunks,
                                                  cur, k.state, k.qrows, k.qcount, meta, s, xhT, counters_zeroed,
                                                  k.q1, k.q1count, k.q1cap)
               : unit_rows && assign_mode() >= 1
                   ? launch_assign_fast(x, d, k.cent, K, B, k.t, k.max_chunks, cur, k.best,
                                        k.qrows, k.qcount, meta, s)
                   : launch_assign(x, d, k.cent, K, k.t, k.max_chunks, cur, k.best, meta, s))
        return rc; }
    if (g_verify_on.load() && k.q1 && (half || wide || wide2 || (unit_rows && assign_mode() >= 1))) {
      // k.q1 (the first level's row queue, consumed inside the launch group above) doubles as
      // the scratch label buffer; k.best is only used by the exact kernel
      if (int rc = launch_assign(x, d, k.cent, K, k.t, k.max_chunks, k.q1, k.best, meta, s)) return rc;
      hipLaunchKernelGGL(verify_compare_kernel, dim3(1024), dim3(256), 0, s, cur, k.q1, meta, (int64_t)k.rows_cap);
      HSGK_LAUNCH_CHECK();
      chain.last = -1;       // (these launches belong to no scope: the next one starts with its own event)
    }
  }
  if (final_labels) {        // the caller reads the labels where (and in the format) the loop left them
    *final_labels = cur;
    return 0;
  }
  if (labels_are_u8(cur)) {
    // bytes -> int32 in k.klab (through the other buffer when the bytes sit in k.klab itself)
    int32_t *base = labels_base(cur);
    int32_t *dst = base == k.klab ? k.klab_prev : k.klab;
    hipLaunchKernelGGL(labels_u8_to_i32_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(base),
                       (int64_t)k.rows_cap, dst);
    HSGK_LAUNCH_CHECK();
    cur = dst;
  }
  if (cur != k.klab)      // odd number of swaps: the final labels sit in the other buffer
    HSGK_CHECK_HIP(hipMemcpyAsync(k.klab, cur, sizeof(int32_t) * k.rows_cap, hipMemcpyDeviceToDevice, s));
  return 0;
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

int hsgk_version(void) { return HSGK_VERSION; }

int hsgk_small_map_groups(int B, int C, int H, int W, int K) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 0;
  const char *se = getenv("HSGK_SMALL");
  if (se && se[0] == '0') return 0;
  const int d = C + 2;
  const int64_t rows = (int64_t)H * W;
  if (!(sums_fx_eligible(d) && K <= 1023 && assign_half_eligible(d, K) && rows <= 16 * 1024 &&
        lloyd_small_eligible(d, K, B, rows)))
    return 0;
  return lloyd_small_groups(B, rows);
}
const char *hsgk_last_error(void) { return g_err; }

void hsgk_profile_enable(int on) { g_prof_on.store(on ? 1 : 0); }

int hsgk_profile_collect(double *ms_sum, int64_t *count) {
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> events;
  {
    std::lock_guard<std::mutex> g(g_prof_mu);
    recs.swap(g_prof);
    events.swap(g_prof_events);
    g_prof_epoch.fetch_add(1);
  }
  for (int i = 0; i < HSGK_PROF_KINDS; ++i) { ms_sum[i] = 0.0; count[i] = 0; }
  for (auto &r : recs) {
    float ms = 0.0f;
    if (r.ia < 0 || r.ib < 0 || r.ia >= (int)events.size() || r.ib >= (int)events.size()) continue;
    if (hipEventSynchronize(events[r.ib]) == hipSuccess &&
        hipEventElapsedTime(&ms, events[r.ia], events[r.ib]) == hipSuccess && r.kind >= 0 && r.kind < HSGK_PROF_KINDS) {
      ms_sum[r.kind] += ms;
      count[r.kind] += 1;
    }
  }
  for (hipEvent_t e : events) (void)hipEventDestroy(e);
  return 0;
}

void hsgk_verify_enable(int on) { g_verify_on.store(on ? 1 : 0); }

int hsgk_verify_collect(uint64_t *rows_compared, uint64_t *rows_differing) {
  unsigned long long v[2] = {0, 0}, zero[2] = {0, 0};
  HSGK_CHECK_HIP(hipDeviceSynchronize());
  HSGK_CHECK_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_verify), sizeof(v)));
  HSGK_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_verify), zero, sizeof(zero)));
  if (rows_compared) *rows_compared = v[0];
  if (rows_differing) *rows_differing = v[1];
  return 0;
}

// ---- host helpers: the two small tables of hsgk_segment_by_kmeans -----------------------------
// torch.linspace(start, end, n) in float32 as ATen's CPU kernel evaluates it (step in float32; element
// i = start + step * i below n / 2, end - step * (n - 1 - i) above, one rounding each): its bit pattern
// is part of the reference's behaviour (hsg/utils/segsort/common.py:145-148, 175-187).
static void linspace_f32(float start, float end, int n, std::vector<float> *out) {
  out->resize((size_t)n);
  if (n == 1) { (*out)[0] = start; return; }
  const float step = (end - start) / (float)(n - 1);
  for (int i = 0; i < n; ++i)
    (*out)[i] = i < n / 2 ? (float)((double)start + (double)step * (double)i)
                          : (float)((double)end - (double)step * (double)(n - 1 - i));
}

int hsgk_host_grid_seed_map(int ky, int kx, int H, int W, int32_t *seed_map, int32_t *num_clusters) {
  HSGK_REQUIRE(ky >= 1 && kx >= 1 && H >= 1 && W >= 1 && seed_map && num_clusters, "bad arguments");
  std::vector<float> fy, fx;
  linspace_f32(0.0f, (float)(ky - 1), H, &fy);
  linspace_f32(0.0f, (float)(kx - 1), W, &fx);
  std::vector<long long> y((size_t)H), x((size_t)W);
  long long ymax = 0;
  for (int i = 0; i < H; ++i) { y[i] = (long long)rintf(fy[i]); ymax = y[i] > ymax ? y[i] : ymax; }   // round half to even
  for (int i = 0; i < W; ++i) x[i] = (long long)rintf(fx[i]);
  // y + (y.max() + 1) * x (common.py:150-151), then the rank among the distinct values (:341-342)
  std::vector<long long> vals;
  vals.reserve((size_t)H * W);
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) vals.push_back(y[i] + (ymax + 1) * x[j]);
  std::vector<long long> uniq(vals);
  std::sort(uniq.begin(), uniq.end());
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  for (size_t p = 0; p < vals.size(); ++p)
    seed_map[p] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), vals[p]) - uniq.begin());
  *num_clusters = (int32_t)uniq.size();
  return 0;
}

int hsgk_host_location_features(int H, int W, float *loc) {
  HSGK_REQUIRE(H >= 1 && W >= 1 && loc, "bad arguments");
  std::vector<float> fy, fx;
  linspace_f32(0.0f, 1.0f, H, &fy);
  linspace_f32(0.0f, 1.0f, W, &fx);
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) {
      loc[((size_t)i * W + j) * 2 + 0] = fy[i] - 0.5f;       // (:316 / local_model.py:88-93)
      loc[((size_t)i * W + j) * 2 + 1] = fx[j] - 0.5f;
    }
  return 0;
}

int hsgk_normalize_rows(const float *x, int64_t n, int d, float eps, float *out, float *norms,
                        hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && d >= 1, "bad shape");
  (void)hipGetLastError();
  return launch_normalize_rows(x, n, d, eps, out, norms, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------
size_t hsgk_segment_by_kmeans_workspace_bytes(int B, int C, int H, int W, int K,
                                              int64_t table_cap) {
  const int64_t HW = (int64_t)H * W;
  const int ntiles = (int)((HW + kTilePix - 1) / kTilePix);
  Carver cv(nullptr);
  cv.take<int32_t>((size_t)B * ntiles);                                  // tile_cnt
  cv.take<char>(align_up((size_t)B * ntiles, 64) * 4 + (size_t)B * 8);   // tile_off + img_cnt
  KmeansScratch k;
  carve_kmeans(cv, B, HW, C + 2, K, &k);
  carve_m0(cv, B, C, ntiles, K, &k);
  cv.take<int32_t>((size_t)table_cap * 2);
  cv.take<char>(relabel_scan_bytes(table_cap));
  return cv.off + 256;
}

int hsgk_segment_by_kmeans(const hsgk_segkm_args *a, hsgk_stream_t stream) {
  HSGK_REQUIRE(a != nullptr, "null args");
  HSGK_REQUIRE(a->B >= 1 && a->C >= 1 && a->H >= 1 && a->W >= 1, "bad shape");
  HSGK_REQUIRE(a->K >= 1 && a->iterations >= 0, "bad K / iterations");
  HSGK_REQUIRE(a->embeddings && a->loc && a->seed_map && a->meta, "null input");
  HSGK_REQUIRE(a->out_embeddings && a->out_embeddings_loc && a->out_labels &&
                   a->out_cluster && a->out_batch, "null output");
  HSGK_REQUIRE(a->workspace_bytes >=
                   hsgk_segment_by_kmeans_workspace_bytes(a->B, a->C, a->H, a->W, a->K,
                                                          a->table_cap),
               "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t HW = (int64_t)a->H * a->W;
  const int ntiles = (int)((HW + kTilePix - 1) / kTilePix);
  const int D = a->C + 2;

  Carver cv(a->workspace);
  int32_t *tile_cnt = cv.take<int32_t>((size_t)a->B * ntiles);
  int32_t *tile_off = reinterpret_cast<int32_t *>(
      cv.take<char>(align_up((size_t)a->B * ntiles, 64) * 4 + (size_t)a->B * 8));
  KmeansScratch k;
  carve_kmeans(cv, a->B, HW, D, a->K, &k);
  carve_m0(cv, a->B, a->C, ntiles, a->K, &k);
  int32_t *table = cv.take<int32_t>((size_t)a->table_cap * 2);
  int32_t *scan_tmp = reinterpret_cast<int32_t *>(cv.take<char>(relabel_scan_bytes(a->table_cap)));

  const bool compact = a->labels != nullptr && a->has_ignore;
  const bool want_half = assign_mode() == 2 && k.xh && a->iterations >= 1;
  bool half_ready = false, m0_ready = false, tiles_ready = false;
  // HSGK_M0 = 0 / 1 (read per call, for the tests): never / whenever the shape allows
  const char *m0e = getenv("HSGK_M0");
  const bool m0_env = !(m0e && m0e[0] == '0'), m0_force = m0e && m0e[0] == '1';
  const bool wide_cells = (double)a->H * (double)a->W >= 400.0 * (double)a->K;
  const bool want_m0 = k.m0.part && fx_enabled() && a->iterations >= 1 && k.max_chunks > 0 && m0_env && (wide_cells || m0_force);
  (void)hipGetLastError();   // drop stale errors left by other users of the runtime
  {
    ProfScope p(HSGK_PROF_PREP, s);
    if (want_m0)
      HSGK_CHECK_HIP(hipMemsetAsync(k.sumq, 0, sizeof(long long) * (size_t)a->B * a->K * D, s));
    if (int rc = launch_count_valid(a->labels, a->B, HW, a->has_ignore, a->ignore_index,
                                    tile_cnt, a->meta, s)) return rc;
    if (int rc = launch_build_tables(compact ? tile_cnt : nullptr, a->B, HW, ntiles, tile_off,
                                     k.t, k.max_chunks, a->meta, s)) return rc;
    // the fp16 copy in tile order straight from the prep kernel where the E-step has an engine for it (launch_prep
    // declines when rows are compacted or H * W % 32 != 0); the row-major copy is left out when nothing would read it
    const bool one_group = (a->flags & HSGK_SEGKM_ONE_GROUP) != 0;
    const bool t_shape = k.xhT != nullptr && tlayout_mode() != 0 && !one_group &&
                         (assign_half_wide2_eligible(D, a->K) || tlayout_env());
    const bool tiles_only = t_shape && !compact && HW % 32 == 0 && assign_half_wide2_tiles(D, a->K, k.max_chunks) &&
                            !assign_half_eligible(D, a->K);
    if (int rc = launch_prep(*a, compact ? tile_off : nullptr, k.t, k.klab, s,
                             want_half && !tiles_only ? k.xh : nullptr, k.xt, &half_ready,
                             want_m0 ? &k.m0 : nullptr, &m0_ready, want_half && t_shape ? k.xhT : nullptr,
                             &tiles_ready)) return rc;
    if (tiles_only && !tiles_ready) half_ready = false;       // (declined: lloyd() makes the row-major copy itself)
  }
  const int32_t *final_labels = k.klab;      // int32 or byte labels (common.h), in either label buffer
  if (int rc = lloyd(a->out_embeddings_loc, D, a->K, a->B, a->iterations, k, a->meta, s,
                     /*unit_rows=*/true, half_ready, m0_ready, /*single_group=*/(a->flags & HSGK_SEGKM_ONE_GROUP) != 0,
                     &final_labels, tiles_ready)) return rc;
  {
    ProfScope p(HSGK_PROF_RELABEL, s);
    if (int rc = launch_relabel(*a, k.t, k.max_chunks, final_labels, table, scan_tmp, s)) return rc;
  }
  return 0;
}

int hsgk_segment_by_kmeans_bwd(const float *g_emb, const float *g_emb_loc, const float *emb,
                               const float *emb_loc, const float *norms, const int64_t *rowmap,
                               int B, int C, int H, int W, float eps, float *gx,
                               hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1, "bad shape");
  HSGK_REQUIRE(emb && emb_loc && norms && gx, "null input");
  (void)hipGetLastError();
  return launch_prep_bwd(g_emb, g_emb_loc, emb, emb_loc, norms, rowmap, B, C, H, W, eps, gx,
                         static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------
size_t hsgk_kmeans_workspace_bytes(int64_t n, int d, int K) {
  Carver cv(nullptr);
  cv.take<hsgk_segkm_meta>(1);
  KmeansScratch k;
  carve_kmeans(cv, 1, n, d, K, &k);
  return cv.off + 256;
}

int hsgk_kmeans_with_initial_labels(const float *x, int64_t n, int d, int64_t *labels_io,
                                    int K, int iterations, void *workspace,
                                    size_t workspace_bytes, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && d >= 1 && K >= 1 && iterations >= 0, "bad shape");
  HSGK_REQUIRE(workspace_bytes >= hsgk_kmeans_workspace_bytes(n, d, K), "workspace too small");
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver cv(workspace);
  hsgk_segkm_meta *meta = cv.take<hsgk_segkm_meta>(1);
  KmeansScratch k;
  carve_kmeans(cv, 1, n, d, K, &k);
  if (int rc = launch_flat_table(n, k.t, k.max_chunks, meta, s)) return rc;
  if (int rc = launch_i64_to_i32(labels_io, n, k.klab, s)) return rc;
  if (int rc = lloyd(x, d, K, 1, iterations, k, meta, s)) return rc;
  return launch_i32_to_i64(k.klab, n, labels_io, s);
}

// ---------------------------------------------------------------------------
// Batch-level Lloyd half-steps over B images of `rows_per_image` rows each.
size_t hsgk_lloyd_workspace_bytes(int B, int64_t rows_per_image, int d, int K) {
  Carver cv(nullptr);
  cv.take<hsgk_segkm_meta>(1);
  KmeansScratch k;
  carve_kmeans(cv, B, rows_per_image, d, K, &k);
  return cv.off + 256;
}

static int lloyd_setup(int B, int64_t rows, int d, int K, void *workspace, size_t bytes,
                       KmeansScratch *k, hsgk_segkm_meta **meta, hipStream_t s) {
  HSGK_REQUIRE(B >= 1 && rows >= 1 && d >= 1 && K >= 1, "bad shape");
  HSGK_REQUIRE(bytes >= hsgk_lloyd_workspace_bytes(B, rows, d, K), "workspace too small");
  Carver cv(workspace);
  *meta = cv.take<hsgk_segkm_meta>(1);
  carve_kmeans(cv, B, rows, d, K, k);
  (void)hipGetLastError();
  hipLaunchKernelGGL(init_meta_kernel, dim3(1), dim3(1), 0, s, *meta, 0);
  HSGK_LAUNCH_CHECK();
  return launch_build_tables(nullptr, B, rows, 0, nullptr, k->t, k->max_chunks, *meta, s);
}

int hsgk_lloyd_mstep(const float *x, int B, int64_t rows_per_image, int d, int K,
                     const int32_t *labels, float *centroids, void *workspace,
                     size_t workspace_bytes, hsgk_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  KmeansScratch k; hsgk_segkm_meta *meta;
  if (int rc = lloyd_setup(B, rows_per_image, d, K, workspace, workspace_bytes, &k, &meta, s)) return rc;
  { ProfScope p(HSGK_PROF_ACCUMULATE, s);
    if (int rc = launch_accumulate(x, d, labels, k.t, k.max_chunks, K, k.partial, k.pmask, meta, s)) return rc; }
  ProfScope p(HSGK_PROF_FINALIZE, s);
  return launch_finalize(k.partial, k.pmask, d, K, B, k.t, k.max_chunks / B, HSGK_EPS, centroids, s);
}

int hsgk_lloyd_mstep_exact(const float *x, int B, int64_t rows_per_image, int d, int K,
                           const int32_t *labels_prev, const int32_t *labels, int64_t *sums,
                           float *centroids, void *workspace, size_t workspace_bytes,
                           hsgk_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  HSGK_REQUIRE(x && labels && sums && centroids, "null argument");
  HSGK_REQUIRE(sums_fx_eligible(d), "row length not supported by the exact-sum M-step");
  KmeansScratch k; hsgk_segkm_meta *meta;
  if (int rc = lloyd_setup(B, rows_per_image, d, K, workspace, workspace_bytes, &k, &meta, s)) return rc;
  const int32_t *prev = labels_prev;
  if (!prev) {                                  // from scratch: nothing added yet
    HSGK_CHECK_HIP(hipMemsetAsync(k.klab_prev, 0xFF, sizeof(int32_t) * k.rows_cap, s));
    HSGK_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * (size_t)B * K * d, s));
    prev = k.klab_prev;
  }
  { ProfScope p(HSGK_PROF_ACCUMULATE, s);
    if (int rc = launch_update_sums(x, d, prev, labels, k.t, k.max_chunks, K,
                                    reinterpret_cast<long long *>(sums), meta, s, d)) return rc; }   // (|x| <= 1: hsgk.h)
  ProfScope p(HSGK_PROF_FINALIZE, s);
  return launch_finalize_fx(reinterpret_cast<const long long *>(sums), d, K, B, HSGK_EPS, centroids, s);
}

int hsgk_lloyd_estep(const float *x, int B, int64_t rows_per_image, int d, int K,
                     const float *centroids, int32_t *labels_out, int unit_rows, void *workspace,
                     size_t workspace_bytes, hsgk_stream_t stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  KmeansScratch k; hsgk_segkm_meta *meta;
  if (int rc = lloyd_setup(B, rows_per_image, d, K, workspace, workspace_bytes, &k, &meta, s)) return rc;
  ProfScope p(HSGK_PROF_ASSIGN, s);
  if (unit_rows == 2 && k.xh) {              // fp16 filter first (the copy is made here)
    if (int rc = launch_to_half_rows(x, k.t, k.max_chunks, d, k.xh, k.xt, meta, s)) return rc;
    const _Float16 *xhT = nullptr;
    // (tile-ordered copy: every image here holds rows_per_image rows.  A single E-step would pay the
    //  conversion for one filter pass, so only the explicit switch takes it; the Lloyd loop amortises it.)
    if (k.xhT && tlayout_env()) {
      if (int rc = launch_rows_to_tiles(k.xh, d, (int64_t)k.rows_cap, k.xhT, s)) return rc;
      xhT = k.xhT;
    }
    if (assign_half_eligible(d, K))
      return launch_assign_half(x, k.xh, k.xt, d, centroids, K, B, k.t, k.max_chunks, labels_out, k.q1,
                                k.q1count, k.q1cap, k.qrows, k.qcount, meta, s, false, xhT);
    if (assign_half_wide_eligible(d, K))
      return launch_assign_half_wide(x, k.xh, k.xt, d, centroids, k.errc, K, B, k.t, k.max_chunks, labels_out,
                                     k.qrows, k.qcount, meta, s, false, k.q1, k.q1count, k.q1cap);
    return launch_assign_half_wide2(x, k.xh, k.xt, d, centroids, k.errc, K, B, k.t, k.max_chunks, labels_out,
                                    k.state, k.qrows, k.qcount, meta, s, xhT, false, k.q1, k.q1count, k.q1cap);
  }
  if (unit_rows)
    return launch_assign_fast(x, d, centroids, K, B, k.t, k.max_chunks, labels_out, k.best, k.qrows,
                              k.qcount, meta, s);
  return launch_assign(x, d, centroids, K, k.t, k.max_chunks, labels_out, k.best, meta, s);
}

// Number of rows the last hsgk_lloyd_estep(unit_rows=1) on this workspace sent
// to the exact re-score pass (diagnostics for the bf16 split filter).
__global__ void sum_qcount_kernel(const int32_t *__restrict__ qcount, int n,
                                  int64_t *__restrict__ out) {
  __shared__ long long ws[4];
  long long s = 0;
  for (int i = threadIdx.x; i < n; i += 256) s += qcount[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = ws[0] + ws[1] + ws[2] + ws[3];
}

int hsgk_lloyd_requeued_rows(int B, int64_t rows_per_image, int d, int K, void *workspace,
                             size_t workspace_bytes, int64_t *out, hsgk_stream_t stream) {
  HSGK_REQUIRE(workspace_bytes >= hsgk_lloyd_workspace_bytes(B, rows_per_image, d, K), "workspace too small");
  Carver cv(workspace);
  cv.take<hsgk_segkm_meta>(1);
  KmeansScratch k;
  carve_kmeans(cv, B, rows_per_image, d, K, &k);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sum_qcount_kernel, dim3(1), dim3(256), 0, s, k.qcount, 1, out);
  HSGK_LAUNCH_CHECK();
  if (k.q1count) {                 // rows the fp16 level left to the bf16x3 level (unit_rows = 2)
    // (K > 64: the all-K row lists' lengths, one 128-byte line per image)
    hipLaunchKernelGGL(sum_qcount_kernel, dim3(1), dim3(256), 0, s, k.q1count, assign_half_eligible(d, K) ? B : B * 32,
                       out + 1);
    HSGK_LAUNCH_CHECK();
  } else {
    HSGK_CHECK_HIP(hipMemsetAsync(out + 1, 0, sizeof(int64_t), s));
  }
  return 0;
}

size_t hsgk_assign_workspace_bytes(int64_t n, int d, int K) {
  (void)K;                       // the prototype table is read in place
  return hsgk_kmeans_workspace_bytes(n, d, 1);
}

int hsgk_find_nearest_prototypes(const float *x, int64_t n, int d, const float *prototypes,
                                 int K, int64_t *labels_out, void *workspace,
                                 size_t workspace_bytes, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && d >= 1 && K >= 1, "bad shape");
  HSGK_REQUIRE(workspace_bytes >= hsgk_assign_workspace_bytes(n, d, K), "workspace too small");
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  Carver cv(workspace);
  hsgk_segkm_meta *meta = cv.take<hsgk_segkm_meta>(1);
  KmeansScratch k;
  carve_kmeans(cv, 1, n, d, 1, &k);
  if (int rc = launch_flat_table(n, k.t, k.max_chunks, meta, s)) return rc;
  if (int rc = launch_assign(x, d, prototypes, K, k.t, k.max_chunks, k.klab, k.best, meta, s)) return rc;
  return launch_i32_to_i64(k.klab, n, labels_out, s);
}

}  // extern "C"
