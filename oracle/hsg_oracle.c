/*
 * hsg_oracle.c -- CPU restatement of the HSG dense-pixel clustering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hsg_amd/ may link, import or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * Parity status: PINNED.  tools/gen_golden.py imports the reference
 * (/root/reference, torch CPU) in the build container and writes
 * tests/golden/ *.npz; tests/test_oracle_golden.py checks this file against
 * those vectors (labels bit-exact, floats <= 2e-6 abs).
 *
 * Every function cites the reference lines (relative to /root/reference) it
 * restates.  The reference leaves floating-point summation order to ATen/MKL;
 * this restatement FIXES one order (the "canonical order")
 * and the HIP kernels implement exactly the same order, so GPU-vs-oracle
 * comparisons are bit-exact for floats as well as labels:
 *
 *   C1  sum of squares / dot products: a single fmaf chain in ascending
 *       element index starting from +0.0f.
 *   C2  segment sums: rows are cut into chunks of `chunk` consecutive rows;
 *       inside a chunk each (segment, column) is summed sequentially in row
 *       order starting from +0.0f; chunk partials are then summed
 *       sequentially in chunk order starting from +0.0f.
 *   C3  sqrtf and division are IEEE correctly rounded; no fused contraction
 *       other than the explicit fmaf of C1 (build with -ffp-contract=off).
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* hsg/utils/general/common.py:101-120 normalize_embedding:
 *   norm = ||x||_2 ; norm = where(norm >= eps, norm, eps) ; x / norm        */
static inline float orc_row_norm(const float *x, int d, float eps) {
  float ss = 0.0f;
  for (int i = 0; i < d; ++i) ss = fmaf(x[i], x[i], ss);           /* C1 */
  float nrm = sqrtf(ss);
  if (!(nrm >= eps)) nrm = eps;                /* torch.ge is false on NaN */
  return nrm;
}

ORC_API void orc_normalize_rows(const float *x, int64_t n, int d, float eps,
                                float *out) {
  for (int64_t r = 0; r < n; ++r) {
    const float *xr = x + r * (int64_t)d;
    float *yr = out + r * (int64_t)d;
    float nrm = orc_row_norm(xr, d, eps);
    for (int i = 0; i < d; ++i) yr[i] = xr[i] / nrm;
  }
}

/* hsg/utils/segsort/common.py:306-310 (NCHW->NHWC, normalise), :349-352
 * (append the 2 location channels, re-normalise), :355-365 (drop pixels whose
 * label equals ignore_index, keeping row-major order).
 *
 * loc is addressed as loc[b*loc_sb + (h*W+w)*2 + {0,1}] (loc_sb = 0 when the
 * same [H,W,2] map is shared by all images, common.py:317).
 * Outputs are compacted, image-major.  counts[b] receives the kept pixels of
 * image b.  Returns the total kept-pixel count N.                          */
ORC_API int64_t orc_prep(const float *nchw, int B, int C, int H, int W,
                         const float *loc, int64_t loc_sb,
                         const int64_t *labels, int has_ignore, int64_t ignore,
                         float eps, float *emb, float *emb_loc,
                         int64_t *labels_out, int64_t *counts) {
  const int64_t HW = (int64_t)H * W;
  const int D = C + 2;
  float *row = (float *)malloc(sizeof(float) * (size_t)D);
  int64_t n = 0;
  for (int b = 0; b < B; ++b) {
    int64_t kept = 0;
    for (int64_t p = 0; p < HW; ++p) {
      int64_t lab = labels ? labels[b * HW + p] : 0;
      if (has_ignore && lab == ignore) continue;
      for (int c = 0; c < C; ++c) row[c] = nchw[((int64_t)b * C + c) * HW + p];
      float nrm = orc_row_norm(row, C, eps);
      float *e = emb + n * (int64_t)C;
      for (int c = 0; c < C; ++c) e[c] = row[c] / nrm;
      for (int c = 0; c < C; ++c) row[c] = e[c];
      row[C] = loc[b * loc_sb + p * 2 + 0];
      row[C + 1] = loc[b * loc_sb + p * 2 + 1];
      float nrm2 = orc_row_norm(row, D, eps);
      float *el = emb_loc + n * (int64_t)D;
      for (int c = 0; c < D; ++c) el[c] = row[c] / nrm2;
      labels_out[n] = lab;
      ++n;
      ++kept;
    }
    counts[b] = kept;
  }
  free(row);
  return n;
}

/* hsg/utils/segsort/common.py:11-41 calculate_prototypes_from_labels
 * (zeros -> scatter_add_ -> normalize_embedding), with summation order C2.
 * labels[r] < 0 or >= P rows are skipped (the reference would raise).      */
ORC_API void orc_segment_sums(const float *x, int64_t n, int d,
                              const int64_t *labels, int64_t P, int chunk,
                              float *sums /* [P,d] */) {
  size_t tot = (size_t)P * (size_t)d;
  float *part = (float *)malloc(sizeof(float) * tot);
  for (size_t i = 0; i < tot; ++i) sums[i] = 0.0f;
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    int64_t r1 = r0 + chunk < n ? r0 + chunk : n;
    for (size_t i = 0; i < tot; ++i) part[i] = 0.0f;
    for (int64_t r = r0; r < r1; ++r) {
      int64_t l = labels[r];
      if (l < 0 || l >= P) continue;
      float *pr = part + (size_t)l * d;
      const float *xr = x + r * (int64_t)d;
      for (int i = 0; i < d; ++i) pr[i] = pr[i] + xr[i];
    }
    for (size_t i = 0; i < tot; ++i) sums[i] = sums[i] + part[i];
  }
  free(part);
}

/* Canonical order C2x (the Lloyd loop of segment_by_kmeans, unit-norm rows): EXACT segment
 * sums.  Every element is converted to the fixed-point integer q = rint(x * 2^40) (exact for
 * |x| >= 2^-16), the q are summed as 64-bit integers (any order: integer addition is
 * associative), and the sum is converted to fp32 with one rounding.  Closer to the real sum
 * than any fp32 summation order, and independent of it -- which is what lets the GPU M-step
 * update the sums from the rows whose label changed (csrc/sums_fx.hip).                    */
static void orc_segment_sums_fx(const float *x, int64_t n, int d, const int64_t *labels, int64_t P,
                                float *sums) {
  size_t tot = (size_t)P * (size_t)d;
  int64_t *acc = (int64_t *)calloc(tot, sizeof(int64_t));
  for (int64_t r = 0; r < n; ++r) {
    int64_t l = labels[r];
    if (l < 0 || l >= P) continue;
    for (int i = 0; i < d; ++i) {
      const float v = x[r * (int64_t)d + i] * 65536.0f;            /* exact */
      const float hi = rintf(v);
      const float lo = rintf((v - hi) * 16777216.0f);               /* exact remainder, exact scaling */
      acc[(size_t)l * d + i] += (int64_t)(int32_t)hi * 16777216 + (int64_t)(int32_t)lo;
    }
  }
  for (size_t i = 0; i < tot; ++i) sums[i] = (float)acc[i] * 9.094947017729282e-13f;   /* 2^-40 */
  free(acc);
}

/* prototypes with exact sums (C2x): the M-step of the Lloyd loop of segment_by_kmeans */
ORC_API void orc_prototypes_exact(const float *x, int64_t n, int d, const int64_t *labels, int64_t P,
                                  float eps, float *out);

static void orc_normalize_table(float *out, int64_t P, int d, float eps) {
  for (int64_t k = 0; k < P; ++k) {
    float *row = out + k * (int64_t)d;
    float nrm = orc_row_norm(row, d, eps);
    for (int i = 0; i < d; ++i) row[i] = row[i] / nrm;
  }
}

ORC_API void orc_prototypes_exact(const float *x, int64_t n, int d, const int64_t *labels, int64_t P,
                                  float eps, float *out) {
  orc_segment_sums_fx(x, n, d, labels, P, out);
  orc_normalize_table(out, P, d, eps);
}

ORC_API void orc_prototypes(const float *x, int64_t n, int d,
                            const int64_t *labels, int64_t P, int chunk,
                            float eps, float *out /* [P,d] */) {
  orc_segment_sums(x, n, d, labels, P, chunk, out);
  for (int64_t k = 0; k < P; ++k) {
    float *row = out + k * (int64_t)d;
    float nrm = orc_row_norm(row, d, eps);
    for (int i = 0; i < d; ++i) row[i] = row[i] / nrm;
  }
}

/* hsg/utils/segsort/common.py:44-64 find_nearest_prototypes:
 * argmax_k <x, c_k>, first index on ties; dot product in order C1.
 * ct is the centroid table transposed to [d][K] so the k loop vectorises
 * (each k keeps its own sequential chain -> same bits as the scalar loop). */
static void orc_assign_rows(const float *x, int64_t n, int d, const float *ct,
                            int K, int32_t *labels_out, float *best_out) {
  /* Team size by the work: a thread per ~1 024 rows, at most 32.  (A team of every hardware thread for an 800-row
   * image costs more in wake-ups and barrier waits than the rows themselves -- on a 256-thread GPU box shared
   * with other jobs one E-step took 0.19 s instead of 0.2 ms and the small-map parity tests 17 s each.) */
  int nt = (int)(n / 1024);
  if (nt < 1) nt = 1;
  if (nt > 32) nt = 32;
  if (nt > omp_get_max_threads()) nt = omp_get_max_threads();
#pragma omp parallel num_threads(nt)
  {
    float *acc = (float *)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      const float *xr = x + r * (int64_t)d;
      for (int k = 0; k < K; ++k) acc[k] = 0.0f;
      for (int i = 0; i < d; ++i) {
        const float xv = xr[i];
        const float *c = ct + (size_t)i * K;
        for (int k = 0; k < K; ++k) acc[k] = __builtin_fmaf(xv, c[k], acc[k]);
      }
      int best = 0;
      float bv = acc[0];
      for (int k = 1; k < K; ++k)
        if (acc[k] > bv) { bv = acc[k]; best = k; }
      labels_out[r] = best;
      if (best_out) best_out[r] = bv;
    }
    free(acc);
  }
}

ORC_API void orc_assign(const float *x, int64_t n, int d, const float *cent,
                        int K, int32_t *labels_out, float *best_out) {
  float *ct = (float *)malloc(sizeof(float) * (size_t)K * d);
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < d; ++i) ct[(size_t)i * K + k] = cent[(size_t)k * d + i];
  orc_assign_rows(x, n, d, ct, K, labels_out, best_out);
  free(ct);
}

/* hsg/utils/segsort/common.py:67-97 kmeans_with_initial_labels:
 * `iters` x ( M-step :92 -> E-step :95 ), starting from the given labels.
 * cent_out (nullable) receives the centroids of the last M-step.           */
/* exact_sums != 0: M-step in order C2x (exact fixed-point sums; needs |x| <= 1), else C2. */
ORC_API void orc_kmeans_ex(const float *x, int64_t n, int d, const int32_t *init,
                           int K, int iters, int chunk, float eps, int exact_sums,
                           int32_t *labels_out, float *cent_out);

ORC_API void orc_kmeans(const float *x, int64_t n, int d, const int32_t *init,
                        int K, int iters, int chunk, float eps,
                        int32_t *labels_out, float *cent_out) {
  orc_kmeans_ex(x, n, d, init, K, iters, chunk, eps, 0, labels_out, cent_out);
}

ORC_API void orc_kmeans_ex(const float *x, int64_t n, int d, const int32_t *init,
                           int K, int iters, int chunk, float eps, int exact_sums,
                           int32_t *labels_out, float *cent_out) {
  int64_t *lab64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  float *cent = (float *)malloc(sizeof(float) * (size_t)K * d);
  for (int64_t r = 0; r < n; ++r) labels_out[r] = init[r];
  for (int it = 0; it < iters; ++it) {
    for (int64_t r = 0; r < n; ++r) lab64[r] = labels_out[r];
    if (exact_sums) {
      orc_segment_sums_fx(x, n, d, lab64, K, cent);
      orc_normalize_table(cent, K, d, eps);
    } else {
      orc_prototypes(x, n, d, lab64, K, chunk, eps, cent);
    }
    orc_assign(x, n, d, cent, K, labels_out, NULL);
  }
  if (cent_out) memcpy(cent_out, cent, sizeof(float) * (size_t)K * d);
  free(cent);
  free(lab64);
}

/* hsg/utils/general/common.py:123-147 segment_mean: scatter-sum / count with
 * empty segments dividing by 1; sums in order C2.                          */
ORC_API void orc_segment_mean(const float *x, int64_t n, int d,
                              const int64_t *index, int64_t P, int chunk,
                              float *out) {
  orc_segment_sums(x, n, d, index, P, chunk, out);
  float *cnt = (float *)calloc((size_t)P, sizeof(float));
  for (int64_t r = 0; r < n; ++r)
    if (index[r] >= 0 && index[r] < P) cnt[index[r]] += 1.0f;
  for (int64_t k = 0; k < P; ++k) {
    float c = cnt[k] == 0.0f ? 1.0f : cnt[k];
    for (int i = 0; i < d; ++i) out[k * (int64_t)d + i] /= c;
  }
  free(cnt);
}

/* hsg/utils/segsort/loss.py:15-82 _calculate_log_likelihood, per pixel.
 * Tolerance-based quantity (1e-4): evaluated in double from the float32
 * inputs; the f32 similarity itself follows C1 so exp() sees the same
 * argument the kernels use.  group_plus = 1 for 'segsort+', 0 for 'segsort'.
 * Also returns d(mean loss)/d(emb) and d(mean loss)/d(proto) when non-NULL
 * (scaled by `gscale`, i.e. 1/N for reduction='mean').                      */
/* label affinity of pixel i and prototype j: nc == 0 -> scalar labels, 1 if equal else 0
 * (loss.py:42-44); nc > 0 -> multi-hot rows, sum_c sem[i,c] * psem[j,c] clamped to its sign
 * (loss.py:108-110: "> 0" same, "== 0" different).                                    */
static int orc_affinity(const int64_t *sem, const int64_t *psem, int64_t i, int64_t j, int nc) {
  if (nc == 0) return psem[j] == sem[i] ? 1 : 0;
  int64_t a = 0;
  for (int t = 0; t < nc; ++t) a += sem[i * nc + t] * psem[j * nc + t];
  return a > 0 ? 1 : (a < 0 ? -1 : 0);
}

static void orc_segsort_nll_impl(const float *emb, int64_t n, int c, int nc,
                             const int64_t *sem, const int64_t *inst,
                             const float *proto, int64_t P,
                             const int64_t *psem, float kappa, int group_plus,
                             double *nll /* [n] */, double gscale,
                             double *gemb /* [n,c] or NULL */,
                             double *gproto /* [P,c] or NULL */) {
  if (gproto) memset(gproto, 0, sizeof(double) * (size_t)P * c);
  if (gemb) memset(gemb, 0, sizeof(double) * (size_t)n * c);
  double *s = (double *)malloc(sizeof(double) * (size_t)P);
  double *w = (double *)malloc(sizeof(double) * (size_t)P);
  for (int64_t i = 0; i < n; ++i) {
    const float *e = emb + i * (int64_t)c;
    double same = 0.0, diff = 0.0;
    for (int64_t j = 0; j < P; ++j) {
      const float *p = proto + j * (int64_t)c;
      float acc = 0.0f;
      for (int t = 0; t < c; ++t) acc = fmaf(e[t], p[t], acc);
      s[j] = exp((double)(acc * kappa));
      const int aff = orc_affinity(sem, psem, i, j, nc);
      if (aff > 0) same += s[j]; else if (aff == 0) diff += s[j];
    }
    double own = s[inst[i]];
    double same_wo = same - own;
    int use_same = group_plus && same_wo > 0.0;
    double num = group_plus ? (use_same ? same_wo : own) : own;
    double den = diff + num;
    nll[i] = -log(num / den);
    if (!gemb && !gproto) continue;
    /* nll = -log(num / den), den = diff + num.  With a_j = d num / d s_j and
     * b_j = d diff / d s_j:  d nll / d s_j = a_j (1/den - 1/num) + b_j / den.
     * num = sum_{aff>0} s_j - s_own ('segsort+' with a positive result), else s_own; so
     * a_j = [aff_j > 0] - [j == own] resp. [j == own]; b_j = [aff_j == 0].  The pixel's
     * own prototype can have b = 1 (no affinity with its pixel): it is then in both sums. */
    for (int64_t j = 0; j < P; ++j) {
      const int aff = orc_affinity(sem, psem, i, j, nc);
      const int own_j = (j == inst[i]);
      const double a = (group_plus && use_same) ? (double)((aff > 0) - own_j) : (double)own_j;
      const double b = aff == 0 ? 1.0 : 0.0;
      const double g = a * (1.0 / den - 1.0 / num) + b / den;
      w[j] = g * s[j] * (double)kappa * gscale;   /* d/d(dot_ij) */
    }
    for (int64_t j = 0; j < P; ++j) {
      if (w[j] == 0.0) continue;
      const float *p = proto + j * (int64_t)c;
      if (gemb)
        for (int t = 0; t < c; ++t) gemb[i * (int64_t)c + t] += w[j] * p[t];
      if (gproto)
        for (int t = 0; t < c; ++t) gproto[j * (int64_t)c + t] += w[j] * e[t];
    }
  }
  free(s);
  free(w);
}

ORC_API void orc_segsort_nll(const float *emb, int64_t n, int c,
                             const int64_t *sem, const int64_t *inst,
                             const float *proto, int64_t P,
                             const int64_t *psem, float kappa, int group_plus,
                             double *nll /* [n] */, double gscale,
                             double *gemb /* [n,c] or NULL */,
                             double *gproto /* [P,c] or NULL */) {
  orc_segsort_nll_impl(emb, n, c, 0, sem, inst, proto, P, psem, kappa, group_plus, nll, gscale, gemb,
                       gproto);
}

/* hsg/utils/segsort/loss.py:85-130 _one_hot_calculate_log_likelihood (SetSegSortLoss):
 * sem [n,nc], psem [P,nc] multi-hot labels, same / different by the label affinity.   */
ORC_API void orc_set_segsort_nll(const float *emb, int64_t n, int c, int nc,
                                 const int64_t *sem, const int64_t *inst,
                                 const float *proto, int64_t P,
                                 const int64_t *psem, float kappa, int group_plus,
                                 double *nll, double gscale, double *gemb, double *gproto) {
  orc_segsort_nll_impl(emb, n, c, nc, sem, inst, proto, P, psem, kappa, group_plus, nll, gscale, gemb,
                       gproto);
}


/* hsg/models/embeddings/transformer_clusters.py:99-102 logits of the TransformerClustering
 * tail: logits[b,i,j] = (sum_c cent[b,c,i] * feat[b,c,j]) / (float)sqrt(C), the sum as
 * one fmaf chain over ascending c from +0.0f (C1), then a correctly rounded division.  */
ORC_API void orc_cluster_logits(const float *cent, const float *feat, int B, int C, int tl,
                                int sl, float *out) {
  const float div = (float)sqrt((double)C);
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < tl; ++i)
      for (int j = 0; j < sl; ++j) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c)
          acc = fmaf(cent[((size_t)b * C + c) * tl + i], feat[((size_t)b * C + c) * sl + j], acc);
        out[((size_t)b * tl + i) * sl + j] = acc / div;
      }
}


/* hsg/utils/graph/common.py:23-36 exp_inner_product_kernel: A[b,i,j] = exp(conc * dot),
 * the dot product as one fmaf chain over ascending c (C1), x [B,C,N].  expf in float.   */
ORC_API void orc_exp_affinity(const float *x, int B, int C, int N, float conc, float *out) {
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c)
          acc = fmaf(x[((size_t)b * C + c) * N + i], x[((size_t)b * C + c) * N + j], acc);
        out[((size_t)b * N + i) * N + j] = expf(acc * conc);
      }
}
