// Probe: HBM rate of write-only, copy (1R:1W) and the prep mix (1R : 2.5W) with plain float4 streams.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rw_mix.hip -o /tmp/rw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void fill(float4 *o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void copy(const float4 *a, float4 *o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i];
}
// 1 read, 2 full writes + 1 half write (the prep kernel's traffic mix)
__global__ void mix(const float4 *a, float4 *o1, float4 *o2, float2 *o3, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = a[i];
    o1[i] = v;
    o2[i] = make_float4(v.x * 2, v.y * 2, v.z * 2, v.w * 2);
    o3[i] = make_float2(v.x + v.y, v.z + v.w);
  }
}
int main() {
  const size_t n = (size_t)48 * 448 * 448 * 256 / 4;      // float4 elements of one [N][256] fp32 array (9.87 GB)
  float4 *a, *b, *c; float2 *h;
  (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&c, n * 16); (void)hipMalloc(&h, n * 8);
  (void)hipMemset(a, 0, n * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](const char *name, double gb, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms / 3, gb / (ms / 3) * 1e3);
  };
  const double g = n * 16 / 1e9;
  for (int grid : {2048, 8192}) {
    printf("grid %d\n", grid);
    time("write only", g, [&] { hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, b, n); });
    time("copy 1R:1W", 2 * g, [&] { hipLaunchKernelGGL(copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    time("mix 1R:2.5W", 3.5 * g, [&] { hipLaunchKernelGGL(mix, dim3(grid), dim3(256), 0, 0, a, b, c, h, n); });
  }
  return 0;
}
