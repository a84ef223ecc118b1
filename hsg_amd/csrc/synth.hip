// Portable synthetic pixel embeddings on the device: the same integer hash as
// hsg_amd/utils/synth.py (splitmix64 of the flat element index, four 16-bit
// fields summed, one exact float32 scaling), so the GPU box fills a full-size
// BASELINE batch in HBM with the very bits numpy produces for the fixtures.
#include "common.h"

namespace hsgk {
namespace {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ float gaussish_of(unsigned long long key, unsigned long long idx) {
  const unsigned long long h = splitmix64(idx ^ key);
  const int s = (int)(h & 0xFFFF) + (int)((h >> 16) & 0xFFFF) + (int)((h >> 32) & 0xFFFF) + (int)(h >> 48);
  return (float)(s - 2 * 65535) * (float)(1.0 / 37837.2);
}

__global__ void __launch_bounds__(256) synth_iid_kernel(unsigned long long key, unsigned long long offset,
                                                        int64_t n, float *out) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && (((uintptr_t)(out + i)) & 15) == 0) {
      float4 v;
      v.x = gaussish_of(key, offset + i);
      v.y = gaussish_of(key, offset + i + 1);
      v.z = gaussish_of(key, offset + i + 2);
      v.w = gaussish_of(key, offset + i + 3);
      *reinterpret_cast<float4 *>(out + i) = v;
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) out[j] = gaussish_of(key, offset + j);
    }
  }
}

// out[b][c][y][x] = centres[b][cid(b, y, x)][c] + 0.05f * noise (two roundings, like numpy)
__global__ void __launch_bounds__(256) synth_mixture_kernel(unsigned long long noise_key, unsigned long long seed,
                                                            const float *centres, int ncentres, int b0, int B,
                                                            int C, int H, int W, float *out) {
  const int64_t HW = (int64_t)H * W, n = (int64_t)B * C * HW;
  const int ch = H / 8 > 0 ? H / 8 : 1, cw = W / 8 > 0 ? W / 8 : 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    const int64_t bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const unsigned long long gy = (unsigned long long)((p / W) / ch), gx = (unsigned long long)((p % W) / cw);
    const unsigned long long cid =
        splitmix64((gy * 131ull + gx) * 2654435761ull + ((unsigned long long)(b + b0) * 7919ull + seed)) %
        (unsigned long long)ncentres;
    const float prod = 0.05f * gaussish_of(noise_key, (unsigned long long)(i + (int64_t)b0 * C * HW));
    out[i] = centres[((int64_t)b * ncentres + (int64_t)cid) * C + c] + prod;
  }
}

}  // namespace
}  // namespace hsgk

extern "C" {

int hsgk_synth_gaussish(uint64_t key, uint64_t offset, int64_t n, float *out, hsgk_stream_t stream) {
  using namespace hsgk;
  HSGK_REQUIRE(n >= 0 && (n == 0 || out), "bad arguments");
  if (n == 0) return 0;
  const int64_t want = (n + 1023) / 1024;
  const int grid = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(synth_iid_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned long long)key,
                     (unsigned long long)offset, n, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_synth_mixture(uint64_t noise_key, uint64_t seed, const float *centres, int ncentres, int first_image, int B, int C,
                       int H, int W, float *out, hsgk_stream_t stream) {
  using namespace hsgk;
  HSGK_REQUIRE(centres && out && ncentres > 0 && first_image >= 0 && B > 0 && C > 0 && H > 0 && W > 0, "bad arguments");
  const int64_t n = (int64_t)B * C * H * W;
  const int64_t want = (n + 255) / 256;
  const int grid = (int)(want < 16384 ? want : 16384);
  hipLaunchKernelGGL(synth_mixture_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long)noise_key, (unsigned long long)seed, centres, ncentres, first_image, B, C, H, W, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
