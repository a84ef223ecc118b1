// Probe (round 4): does a buffer that fits the 256 MiB Infinity Cache re-read faster
// than one that does not, with the E-step's access pattern (persistent workgroups, contiguous ranges,
// 16 B per lane)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l3_reread.hip -o tools/probes/l3_reread
// Modes per buffer size S:
//   relaunch   12 launches over the same S bytes, the last 10 timed        (iteration-major re-reads)
//   in-kernel  one launch, every workgroup re-reads its own range 10 times  (no launch boundaries)
//   cold       a 2 GiB flush read, then one timed read                      (first touch from HBM)
//   written    flush, a fill kernel writes S, then one timed read          (is freshly written data resident?)
//   mixed      per round: read S, then read 0.3 S of another buffer + write 0.01 S (what an iteration's M-step adds)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int T>
__global__ __launch_bounds__(T) void rd(const uint4 *__restrict__ p, size_t n16, int reps, unsigned *__restrict__ sink) {
  const size_t step = T * 4;
  const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + step - 1) / step * step;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = b0 + threadIdx.x; i < b1; i += step) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t j = i + (size_t)u * T;
        if (j < b1) { const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p) + j); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
        else v[u] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345677u) sink[0] = acc;
}
template <int T>
__global__ __launch_bounds__(T) void rd_plain(const uint4 *__restrict__ p, size_t n16, int reps, unsigned *__restrict__ sink) {
  const size_t step = T * 4;
  const size_t per = ((n16 + gridDim.x - 1) / gridDim.x + step - 1) / step * step;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = b0 + threadIdx.x; i < b1; i += step) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t j = i + (size_t)u * T;
        v[u] = j < b1 ? p[j] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (acc == 0x12345677u) sink[0] = acc;
}
__global__ void fill(uint4 *o, size_t n16, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    o[i] = make_uint4(v, v + 1, v + 2, v + 3);
}

int main() {
  const size_t MB = 1000 * 1000;
  const size_t big = 4970 * MB, flushb = (size_t)2 << 30;
  uint4 *buf, *fl, *other; unsigned *sink;
  if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&fl, flushb) != hipSuccess || hipMalloc(&other, big) != hipSuccess) return 1;
  (void)hipMalloc(&sink, 64);
  (void)hipMemset(buf, 1, big); (void)hipMemset(fl, 2, flushb); (void)hipMemset(other, 3, big);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto ms_of = [&](auto f) { (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                             float ms; (void)hipEventElapsedTime(&ms, e0, e1); return (double)ms; };
  auto flush = [&] { hipLaunchKernelGGL(rd_plain<256>, dim3(2048), dim3(256), 0, 0, fl, flushb / 16, 1, sink); };
  const int sizes[] = {16, 32, 64, 104, 160, 208, 256, 320, 416, 832, 4970};
  for (int variant = 0; variant < 3; ++variant) {
    const char *vn = variant == 0 ? "256 WG x 512 thr, nontemporal loads" : variant == 1 ? "256 WG x 512 thr, plain loads"
                                                                                         : "2048 WG x 256 thr, plain loads";
    printf("== %s ==\n%8s %12s %12s %12s %12s %12s   (GB/s)\n", vn, "S (MB)", "relaunch", "in-kernel", "cold", "written", "mixed");
    auto launch = [&](const uint4 *p, size_t n16, int reps) {
      if (variant == 0) hipLaunchKernelGGL(rd<512>, dim3(256), dim3(512), 0, 0, p, n16, reps, sink);
      else if (variant == 1) hipLaunchKernelGGL(rd_plain<512>, dim3(256), dim3(512), 0, 0, p, n16, reps, sink);
      else hipLaunchKernelGGL(rd_plain<256>, dim3(2048), dim3(256), 0, 0, p, n16, reps, sink);
    };
    for (int smb : sizes) {
      const size_t S = (size_t)smb * MB, n16 = S / 16;
      launch(buf, n16, 1); launch(buf, n16, 1);
      const double t_re = ms_of([&] { for (int i = 0; i < 10; ++i) launch(buf, n16, 1); }) / 10;
      const double t_in = ms_of([&] { launch(buf, n16, 10); }) / 10;
      flush(); (void)hipDeviceSynchronize();
      const double t_cold = ms_of([&] { launch(buf, n16, 1); });
      flush();
      hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, buf, n16, 7u);
      (void)hipDeviceSynchronize();
      const double t_wr = ms_of([&] { launch(buf, n16, 1); });
      // mixed: the timed figure counts only the S bytes (what the E-step would see)
      double t_mix = 0;
      for (int i = 0; i < 12; ++i) {
        const double t = ms_of([&] { launch(buf, n16, 1); });
        if (i >= 2) t_mix += t / 10;
        launch(other, (size_t)(0.3 * S) / 16, 1);
        hipLaunchKernelGGL(fill, dim3(512), dim3(256), 0, 0, other + big / 32, (size_t)(0.01 * S) / 16 + 1, 9u);
      }
      (void)hipDeviceSynchronize();
      printf("%8d %12.0f %12.0f %12.0f %12.0f %12.0f\n", smb, S / t_re / 1e6, S / t_in / 1e6, S / t_cold / 1e6, S / t_wr / 1e6,
             S / t_mix / 1e6);
    }
  }
  return 0;
}
