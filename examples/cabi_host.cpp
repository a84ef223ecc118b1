// A torch-free host of libhsgk's C ABI (include/hsgk.h): what a non-Python caller of the reference's
// `segment_by_kmeans` (hsg/utils/segsort/common.py:270-408) links against.
//
//   hipcc -O2 --offload-arch=gfx950 -Iinclude examples/cabi_host.cpp -Lhsg_amd/csrc -lhsgk \
//         -Wl,-rpath,'$ORIGIN/../hsg_amd/csrc' -o examples/cabi_host
//   examples/cabi_host [B C H W ky kx iterations]          (default: 8 256 224 224 8 8 10)
//
// Generates a synthetic batch in HBM (the portable generator, hsgk_synth_gaussish), builds the seed map
// and the location features with the library's host helpers, calls hsgk_segment_by_kmeans twice on the
// caller's stream, checks that both calls agree bit for bit (run-to-run determinism), that the rows are
// unit vectors and the ids dense, and prints the time per call and a checksum of the cluster ids.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "hsgk.h"

#define CHECK_HIP(e) do { hipError_t err_ = (e); if (err_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); return 2; } } while (0)
#define CHECK_HSGK(e) do { int rc_ = (e); if (rc_ != 0) { \
  fprintf(stderr, "%s:%d libhsgk error %d: %s\n", __FILE__, __LINE__, rc_, hsgk_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
  int v[7] = {8, 256, 224, 224, 8, 8, 10};
  for (int i = 0; i < 7 && i + 1 < argc; ++i) v[i] = atoi(argv[i + 1]);
  const int B = v[0], C = v[1], H = v[2], W = v[3], ky = v[4], kx = v[5], iters = v[6];
  const int64_t HW = (int64_t)H * W, N = B * HW;
  const int D = C + 2;

  // host-side tables (float32 linspace bits of the reference), then to the device
  std::vector<int32_t> seed((size_t)HW);
  std::vector<float> loc((size_t)HW * 2);
  int32_t K = 0;
  CHECK_HSGK(hsgk_host_grid_seed_map(ky, kx, H, W, seed.data(), &K));
  CHECK_HSGK(hsgk_host_location_features(H, W, loc.data()));

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  float *x, *d_loc, *emb, *eloc;
  int32_t *d_seed;
  int64_t *labels, *cluster[2], *batch;
  hsgk_segkm_meta *meta;
  CHECK_HIP(hipMalloc(&x, sizeof(float) * N * C));
  CHECK_HIP(hipMalloc(&d_loc, sizeof(float) * HW * 2));
  CHECK_HIP(hipMalloc(&d_seed, sizeof(int32_t) * HW));
  CHECK_HIP(hipMalloc(&emb, sizeof(float) * N * C));
  CHECK_HIP(hipMalloc(&eloc, sizeof(float) * N * D));
  CHECK_HIP(hipMalloc(&labels, sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc(&cluster[0], sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc(&cluster[1], sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc(&batch, sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc(&meta, sizeof(hsgk_segkm_meta)));
  CHECK_HIP(hipMemcpyAsync(d_loc, loc.data(), sizeof(float) * HW * 2, hipMemcpyHostToDevice, stream));
  CHECK_HIP(hipMemcpyAsync(d_seed, seed.data(), sizeof(int32_t) * HW, hipMemcpyHostToDevice, stream));
  CHECK_HSGK(hsgk_synth_gaussish(0x9E3779B97F4A7C15ull, 0, N * C, x, stream));

  const int64_t table_cap = (int64_t)B * K;                     // no label map: one entry per (image, cluster)
  const size_t ws_bytes = hsgk_segment_by_kmeans_workspace_bytes(B, C, H, W, K, table_cap);
  void *ws;
  CHECK_HIP(hipMalloc(&ws, ws_bytes));

  hsgk_segkm_args a = {};
  a.embeddings = x; a.labels = nullptr; a.loc = d_loc; a.loc_batch_stride = 0; a.seed_map = d_seed;
  a.B = B; a.C = C; a.H = H; a.W = W; a.K = K; a.iterations = iters; a.has_ignore = 0; a.ignore_index = 0;
  a.batch_offset = 0; a.table_cap = table_cap;
  a.out_embeddings = emb; a.out_embeddings_loc = eloc; a.out_labels = labels; a.out_batch = batch;
  a.meta = meta; a.out_norms = nullptr; a.out_rowmap = nullptr; a.workspace = ws; a.workspace_bytes = ws_bytes;

  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  float ms[2] = {0.f, 0.f};
  for (int run = 0; run < 2; ++run) {
    a.out_cluster = cluster[run];
    CHECK_HIP(hipEventRecord(e0, stream));
    CHECK_HSGK(hsgk_segment_by_kmeans(&a, stream));
    CHECK_HIP(hipEventRecord(e1, stream));
    CHECK_HIP(hipEventSynchronize(e1));
    CHECK_HIP(hipEventElapsedTime(&ms[run], e0, e1));
  }
  hsgk_segkm_meta m;
  CHECK_HIP(hipMemcpy(&m, meta, sizeof(m), hipMemcpyDeviceToHost));
  std::vector<int64_t> c0((size_t)N), c1((size_t)N);
  CHECK_HIP(hipMemcpy(c0.data(), cluster[0], sizeof(int64_t) * N, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(c1.data(), cluster[1], sizeof(int64_t) * N, hipMemcpyDeviceToHost));
  std::vector<float> row((size_t)D);
  CHECK_HIP(hipMemcpy(row.data(), eloc + (N / 2) * D, sizeof(float) * D, hipMemcpyDeviceToHost));
  double nrm = 0.0;
  for (int i = 0; i < D; ++i) nrm += (double)row[i] * row[i];
  uint64_t sum = 0;
  int64_t cmax = 0, diff = 0;
  for (int64_t i = 0; i < N; ++i) {
    sum = sum * 1099511628211ull + (uint64_t)c0[i];
    cmax = c0[i] > cmax ? c0[i] : cmax;
    diff += c0[i] != c1[i];
  }
  // ---- the batch-wide prototype table of that output through the C ABI's exchange (one process, no
  //      communicator: hsg/models/utils.py:127-217; with an ncclComm_t the same call spans the ranks)
  const int64_t cap = (int64_t)B * K + 64, pool_rows = 256 * ((N + HSGK_CHUNK - 1) / HSGK_CHUNK);
  hsgk_exchange_args xa = {};
  float *table, *protos, *protos_loc, *pnorms;
  int64_t *plab, *ids, *xmeta;
  CHECK_HIP(hipMalloc(&table, sizeof(float) * cap * (C + D)));
  CHECK_HIP(hipMalloc(&protos, sizeof(float) * cap * C));
  CHECK_HIP(hipMalloc(&protos_loc, sizeof(float) * cap * D));
  CHECK_HIP(hipMalloc(&pnorms, sizeof(float) * cap * 2));
  CHECK_HIP(hipMalloc(&plab, sizeof(int64_t) * cap * 3));
  CHECK_HIP(hipMalloc(&ids, sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc(&xmeta, sizeof(int64_t) * 8));
  xa.embeddings = emb; xa.embeddings_loc = eloc; xa.cluster = cluster[0]; xa.batch = batch;
  xa.semantic = labels; xa.instance = labels;                       // (no label map: all zero)
  xa.n = N; xa.C = C; xa.D = D; xa.cap_local = cap; xa.cap_total = cap; xa.pool_rows = pool_rows; xa.eps = HSGK_EPS;
  xa.table = table; xa.prototypes = protos; xa.prototypes_loc = protos_loc; xa.norms = pnorms;
  xa.proto_semantic = plab; xa.proto_instance = plab + cap; xa.proto_batch = plab + 2 * cap;
  xa.updated_cluster = ids; xa.meta = xmeta;
  xa.workspace_bytes = hsgk_exchange_workspace_bytes(N, C, D, cap, cap, 1, pool_rows);
  CHECK_HIP(hipMalloc(&xa.workspace, xa.workspace_bytes));
  float xms = 0.f;
  for (int run = 0; run < 2; ++run) {
    CHECK_HIP(hipEventRecord(e0, stream));
    CHECK_HSGK(hsgk_exchange_prototypes(&xa, nullptr, 0, 1, stream));
    CHECK_HIP(hipEventRecord(e1, stream));
    CHECK_HIP(hipEventSynchronize(e1));
    CHECK_HIP(hipEventElapsedTime(&xms, e0, e1));
  }
  int64_t xm[8];
  CHECK_HIP(hipMemcpy(xm, xmeta, sizeof(xm), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(c1.data(), ids, sizeof(int64_t) * N, hipMemcpyDeviceToHost));
  int64_t idiff = 0;
  for (int64_t i = 0; i < N; ++i) idiff += c0[i] != c1[i];          // the ids were dense and sorted already
  std::vector<float> prow((size_t)C);
  CHECK_HIP(hipMemcpy(prow.data(), protos + (size_t)(xm[1] / 2) * C, sizeof(float) * C, hipMemcpyDeviceToHost));
  double pn = 0.0;
  for (int i = 0; i < C; ++i) pn += (double)prow[i] * prow[i];
  const bool xok = xm[2] == 0 && xm[1] == m.n_segments && idiff == 0 && fabs(pn - 1.0) < 1e-5;
  printf("exchange: %lld prototypes, %.3f ms, ids unchanged: %s, |prototype| = %.7f -> %s\n", (long long)xm[1], xms,
         idiff == 0 ? "yes" : "NO", sqrt(pn), xok ? "ok" : "FAILED");
  const bool ok = m.error == 0 && m.n_rows == N && diff == 0 && fabs(nrm - 1.0) < 1e-5 && cmax < (int64_t)B * K &&
                  m.n_segments == cmax + 1 && xok;
  printf("%dx%dx%dx%d grid %dx%d (K=%d) it=%d: %.3f ms per call (first %.3f), %.1f Mpixel/s, rows %lld, segments %lld, "
         "cluster-id checksum %016llx, second call identical: %s, |row| = %.7f  -> %s\n",
         B, C, H, W, ky, kx, K, iters, ms[1], ms[0], N / (ms[1] * 1e3), (long long)m.n_rows, (long long)m.n_segments,
         (unsigned long long)sum, diff == 0 ? "yes" : "NO", sqrt(nrm), ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
