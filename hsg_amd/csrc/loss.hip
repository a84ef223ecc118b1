// loss.hip -- pixel-to-segment contrastive ("SegSort") loss, reference
// hsg/utils/segsort/loss.py:15-82 (_calculate_log_likelihood) and :149-190.
//
// The reference materialises S = exp(kappa * E P^T) as an [N,P] matrix plus
// seven [N,P] temporaries.  Here the fp32-MFMA engine of score_tiles.h streams
// the pixel rows against 64-prototype blocks and the epilogue folds every
// score straight into three per-row sums (own / same-semantic / different-
// semantic); nothing of size N x P exists in the forward pass.  Tolerance
// quantity (north_star: loss within 1e-4): the dot products are still the
// canonical C1 chains, the sums over prototypes run in ascending prototype
// order per lane, then lane pair, then block order -- deterministic.
//
// Backward: the per-pair weights dL/d(e_i . p_j) are produced by the same
// engine, written transposed (W^T [P,N], coalesced along rows) and contracted
// with the two plain library GEMMs g_E = W P and g_P = W^T E on the host side
// (torch.mm -> rocBLAS), as the task allows for plain GEMMs.
#include "common.h"
#include "score_tiles.h"

namespace hsgk {

// "same semantic label": equal labels (SegSortLoss), or -- set mode, SetSegSortLoss --
// a non-zero label affinity: sem / psem then carry one bit per class and the affinity
// sum_c sem[i,c] * psem[p,c] of non-negative multi-hot labels is > 0 iff the masks meet
__device__ inline bool same_semantic(int64_t a, int64_t b, int set_mode) {
  return set_mode ? (a & b) != 0 : a == b;
}

struct LossFwdEpi {
  int kb0, nrows, pb;
  int64_t P, N, crow0;
  float kappa;
  const int64_t *sem, *inst, *psem;
  float *part;                                   // [npb][N][3]
  int set_mode;
  template <int MB>
  __device__ inline void operator()(int tile, const f32x16 (&acc)[MB]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const int64_t row = crow0 + (valid ? px : 0);
    const int64_t sj = sem[row], ij = inst[row];
    float own = 0.0f, same = 0.0f, diff = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t p = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (p < P) {
          const float s = expf(acc[m][r] * kappa);
          if (p == ij) own += s;
          if (same_semantic(psem[p], sj, set_mode)) same += s; else diff += s;
        }
      }
    own += __shfl_xor(own, 32);
    same += __shfl_xor(same, 32);
    diff += __shfl_xor(diff, 32);
    if (h == 0 && valid) {
      float *o = part + ((int64_t)pb * N + row) * 3;
      o[0] = own; o[1] = same; o[2] = diff;
    }
  }
};

struct LossBwdEpi {
  int kb0, nrows, group_plus;
  int64_t P, N, crow0;
  float kappa;
  const int64_t *sem, *inst, *psem;
  const float *num, *den, *gscale;
  const int32_t *use_same;
  float *wt;                                     // [P][N]
  int set_mode;
  template <int MB>
  __device__ inline void operator()(int tile, const f32x16 (&acc)[MB]) const {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int TPX = (int)(blockDim.x >> 1);
    const int px = tile * TPX + w * 32 + j;
    const bool valid = px < nrows;
    const int64_t row = crow0 + (valid ? px : 0);
    const int64_t sj = sem[row], ij = inst[row];
    const float inv_num = 1.0f / num[row], inv_den = 1.0f / den[row];
    const float gs = gscale[row] * kappa;
    const bool us = use_same[row] != 0;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t p = kb0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (p < P && valid) {
          const float s = expf(acc[m][r] * kappa);
          // nll = -log(num / den), den = diff + num; with a = d num / d s_p, b = d diff / d s_p:
          // d nll / d s_p = a (1/den - 1/num) + b / den.  num = sum_same - own ('segsort+',
          // positive) else own; the own prototype can also sit in the "different" sum
          // (labels without affinity), loss.py:71-78 / 118-127
          const bool same = same_semantic(psem[p], sj, set_mode);
          const float a = (group_plus && us) ? (float)((int)same - (int)(p == ij)) : (p == ij ? 1.0f : 0.0f);
          const float g = a * (inv_den - inv_num) + (same ? 0.0f : inv_den);
          wt[p * N + row] = g * s * gs;
        }
      }
  }
};

template <int KB, int NW, int KC, bool EVEN_D, class Epi>
__global__ __launch_bounds__(NW * 64) void loss_tiles_kernel(
    const float *__restrict__ emb, int c, const float *__restrict__ proto, int64_t P, int64_t N,
    int split, Epi epi_proto) {
  constexpr int TPX = NW * 32;
  extern __shared__ float lds[];
  const int chunk = blockIdx.x / split, part = blockIdx.x - chunk * split;
  const int pb = blockIdx.y;
  const int tps = (HSGK_CHUNK / TPX + split - 1) / split;
  const int64_t c_row0 = (int64_t)chunk * HSGK_CHUNK;
  const int c_rows = (int)((N - c_row0) < HSGK_CHUNK ? (N - c_row0) : HSGK_CHUNK);
  const int nrows = min(c_rows - part * tps * TPX, tps * TPX);
  if (nrows <= 0) return;
  Epi epi = epi_proto;
  epi.kb0 = pb * KB;
  epi.nrows = nrows;
  epi.crow0 = c_row0 + (int64_t)part * tps * TPX;
  if constexpr (requires { epi.pb; }) epi.pb = pb;
  const int kvalid = (int)((P - (int64_t)pb * KB) < KB ? (P - (int64_t)pb * KB) : KB);
  score_tiles<KB, NW, KC, EVEN_D>(emb, c, proto + (int64_t)pb * KB * c, kvalid, epi.crow0, nrows,
                                  lds, epi);
}

// per-row finish: sums over prototype blocks in block order, then
// loss.py:63-80 (numerator choice, -log(num / (num + diff)))
__global__ void loss_rows_kernel(const float *__restrict__ part, int npb, int64_t N,
                                 int group_plus, float *__restrict__ nll, float *__restrict__ num_o,
                                 float *__restrict__ den_o, int32_t *__restrict__ use_same) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  float own = 0.0f, same = 0.0f, diff = 0.0f;
  for (int b = 0; b < npb; ++b) {
    const float *p = part + ((int64_t)b * N + r) * 3;
    own += p[0]; same += p[1]; diff += p[2];
  }
  float num = own;
  int us = 0;
  if (group_plus) {
    const float same_wo = same - own;
    us = same_wo > 0.0f;
    num = us ? same_wo : own;
  }
  const float den = diff + num;
  nll[r] = -logf(num / den);
  num_o[r] = num;
  den_o[r] = den;
  use_same[r] = us;
}

template <class Epi>
static int launch_loss_tiles(const float *emb, int64_t N, int c, const float *proto, int64_t P,
                             Epi epi, hipStream_t s) {
  if (N <= 0 || P <= 0) return 0;
  const int nch = (int)((N + HSGK_CHUNK - 1) / HSGK_CHUNK);
  const int npb = (int)((P + 63) / 64);
  const bool even = (c & 1) == 0;
  auto go = [&](auto kern, size_t lds, int tiles_per_chunk) -> int {
    int split = 1;
    while (split < tiles_per_chunk && (int64_t)nch * npb * split < 1024) split *= 2;
    HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nch * split, npb), dim3(512), lds, s, emb, c, proto, P, N, split, epi);
    HSGK_LAUNCH_CHECK();
    return 0;
  };
  const size_t l32 = score_tiles_lds_bytes<64, 8, 32>(c), l16 = score_tiles_lds_bytes<64, 8, 16>(c);
  if (l32 <= 160 * 1024)
    return even ? go(loss_tiles_kernel<64, 8, 32, true, Epi>, l32, HSGK_CHUNK / 256)
                : go(loss_tiles_kernel<64, 8, 32, false, Epi>, l32, HSGK_CHUNK / 256);
  if (l16 <= 160 * 1024)
    return even ? go(loss_tiles_kernel<64, 8, 16, true, Epi>, l16, HSGK_CHUNK / 256)
                : go(loss_tiles_kernel<64, 8, 16, false, Epi>, l16, HSGK_CHUNK / 256);
  set_error("segsort loss: embedding dimension %d does not fit the LDS prototype block", c);
  return -1;
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

size_t hsgk_segsort_loss_workspace_bytes(int64_t n, int c, int64_t P) {
  (void)c;
  const int64_t npb = (P + 63) / 64;
  return (size_t)(npb > 0 ? npb : 1) * (size_t)(n > 0 ? n : 1) * 3 * sizeof(float) + 256;
}

int hsgk_segsort_loss_fwd(const float *emb, int64_t n, int c, const int64_t *sem,
                          const int64_t *inst, const float *proto, int64_t P, const int64_t *psem,
                          float kappa, int group_plus, float *nll, float *num, float *den,
                          int32_t *use_same, void *workspace, size_t workspace_bytes,
                          hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && c >= 1 && P >= 1, "bad shape");
  HSGK_REQUIRE(workspace_bytes >= hsgk_segsort_loss_workspace_bytes(n, c, P), "workspace too small");
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  float *part = static_cast<float *>(workspace);
  LossFwdEpi epi{0, 0, 0, P, n, 0, kappa, sem, inst, psem, part, (group_plus >> 1) & 1};
  if (int rc = launch_loss_tiles(emb, n, c, proto, P, epi, s)) return rc;
  const int npb = (int)((P + 63) / 64);
  hipLaunchKernelGGL(loss_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, npb,
                     n, group_plus & 1, nll, num, den, use_same);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_segsort_loss_bwd_weights(const float *emb, int64_t n, int c, const int64_t *sem,
                                  const int64_t *inst, const float *proto, int64_t P,
                                  const int64_t *psem, float kappa, int group_plus,
                                  const float *num, const float *den, const int32_t *use_same,
                                  const float *gscale, float *wt, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && c >= 1 && P >= 1, "bad shape");
  if (n == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  LossBwdEpi epi{0, 0, group_plus & 1, P, n, 0, kappa, sem, inst, psem, num, den, gscale, use_same, wt,
                 (group_plus >> 1) & 1};
  return launch_loss_tiles(emb, n, c, proto, P, epi, s);
}

}  // extern "C"
