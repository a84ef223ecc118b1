// infer.hip -- kernels of the full-resolution inference pipeline around k-means
// (pyscripts/inference/prototype.py:141-208, inference.py:141-200;
//  hsg/utils/segsort/common.py:221-268):
//   * overlap-averaged patch accumulation: every crop's embedding is L2-normalised
//     per pixel (general/common.py:101-120) and added into the full-resolution canvas,
//     a counter plane counts the covering crops, the canvas is divided at the end;
//   * find_majority_label_index: per-cluster class histogram, argmax, pixel selection.
#include "common.h"

namespace hsgk {

// one thread per crop pixel, lanes along x (coalesced plane accesses): C1 chain over
// the channels, then canvas += x / norm (second pass over the crop's planes, L2-resident)
__global__ __launch_bounds__(256) void overlap_accumulate_kernel(
    const float *__restrict__ crop, int C, int h, int w, float *__restrict__ canvas,
    float *__restrict__ counts, int H, int W, int sh, int sw, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)h * w) return;
  const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
  const int Y = sh + y, X = sw + x;
  if (Y < 0 || Y >= H || X < 0 || X >= W) return;
  const int64_t hw = (int64_t)h * w, HW = (int64_t)H * W;
  const float *src = crop + i;
  float ss = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float v = src[c * hw];
    ss = fmaf(v, v, ss);
  }
  float n = sqrtf(ss);
  if (!(n >= eps)) n = eps;
  float *dst = canvas + (int64_t)Y * W + X;
  for (int c = 0; c < C; ++c) dst[c * HW] = dst[c * HW] + src[c * hw] / n;
  counts[(int64_t)Y * W + X] += 1.0f;
}

__global__ __launch_bounds__(256) void overlap_finish_kernel(float *__restrict__ canvas,
                                                             const float *__restrict__ counts, int C,
                                                             int64_t HW) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float cnt = counts[i];
  for (int c = 0; c < C; ++c) canvas[c * HW + i] = canvas[c * HW + i] / cnt;
}

// ---- find_majority_label_index --------------------------------------------
__global__ __launch_bounds__(256) void majority_hist_kernel(const int64_t *__restrict__ sem,
                                                            const int64_t *__restrict__ clu, int64_t n,
                                                            int num_classes, int32_t *__restrict__ hist) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    atomicAdd(hist + clu[i] * num_classes + sem[i], 1);
}

// first maximal class per cluster (torch.argmax on CPU), one wave per cluster
__global__ __launch_bounds__(256) void majority_argmax_kernel(const int32_t *__restrict__ hist,
                                                              int64_t num_clusters, int num_classes,
                                                              int64_t *__restrict__ majority) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= num_clusters) return;
  int bv = -1, bi = 0x7fffffff;
  for (int c = lane; c < num_classes; c += 64) {
    const int v = hist[k * num_classes + c];
    if (v > bv) { bv = v; bi = c; }                  // ascending c per lane: first maximum
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int ov = __shfl_xor(bv, off), oi = __shfl_xor(bi, off);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) majority[k] = bi;
}

__global__ __launch_bounds__(256) void majority_select_kernel(const int64_t *__restrict__ sem,
                                                              const int64_t *__restrict__ clu, int64_t n,
                                                              const int64_t *__restrict__ majority,
                                                              uint8_t *__restrict__ select) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    select[i] = majority[clu[i]] == sem[i] ? 1 : 0;
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

int hsgk_overlap_accumulate(const float *crop, int C, int h, int w, float *canvas, float *counts,
                            int H, int W, int sh, int sw, float eps, hsgk_stream_t stream) {
  HSGK_REQUIRE(C >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "bad shape");
  HSGK_REQUIRE(crop && canvas && counts, "null argument");
  HSGK_REQUIRE(sh >= 0 && sw >= 0 && sh + h <= H && sw + w <= W, "crop outside the canvas");
  (void)hipGetLastError();
  const int64_t n = (int64_t)h * w;
  hipLaunchKernelGGL(overlap_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), crop, C, h, w, canvas, counts, H, W, sh, sw, eps);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_overlap_finish(float *canvas, const float *counts, int C, int H, int W, hsgk_stream_t stream) {
  HSGK_REQUIRE(C >= 1 && H >= 1 && W >= 1 && canvas && counts, "bad arguments");
  (void)hipGetLastError();
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(overlap_finish_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), canvas, counts, C, HW);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_majority_labels(const int64_t *semantic, const int64_t *cluster, int64_t n,
                         int64_t num_clusters, int num_classes, int32_t *hist, int64_t *majority,
                         uint8_t *select, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && num_clusters >= 1 && num_classes >= 1, "bad shape");
  HSGK_REQUIRE(semantic && cluster && hist && majority && select, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  HSGK_CHECK_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)num_clusters * num_classes, s));
  if (n > 0) {
    const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(majority_hist_kernel, dim3(grid), dim3(256), 0, s, semantic, cluster, n,
                       num_classes, hist);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(majority_argmax_kernel, dim3((unsigned)((num_clusters + 3) / 4)), dim3(256), 0, s,
                     hist, num_clusters, num_classes, majority);
  HSGK_LAUNCH_CHECK();
  if (n > 0) {
    const unsigned grid = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(majority_select_kernel, dim3(grid), dim3(256), 0, s, semantic, cluster, n,
                       majority, select);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
