// Probe: cycles per wave instruction of LDS read-modify-write flavours on gfx950, 64 lanes hitting
// 64 consecutive elements of one table row (no bank conflicts), many rows.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomics.hip -o /tmp/lds_at
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(long long *out, int iters) {
  extern __shared__ unsigned char raw[];
  unsigned long long *t64 = reinterpret_cast<unsigned long long *>(raw);   // [64 rows][64]
  float *t32 = reinterpret_cast<float *>(raw);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) t64[i] = 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int row = (it * 7 + w * 13) & 63;
    if (MODE == 0) atomicAdd(&t64[row * 64 + lane], (unsigned long long)(it + lane));          // ds_add_u64
    if (MODE == 1) t64[row * 64 + lane] += (unsigned long long)(it + lane);                      // plain 64-bit RMW
    if (MODE == 2) atomicAdd(&t32[row * 64 + lane], (float)(it + lane));                         // ds_add_f32
    if (MODE == 3) atomicAdd(reinterpret_cast<unsigned int *>(t32) + row * 64 + lane, (unsigned)(it + lane));  // ds_add_u32
    if (MODE == 4 && lane < 2) atomicAdd(&t64[row * 64 + lane], (unsigned long long)(it + lane));   // ds_add_u64, 2 active lanes
    if (MODE == 5 && lane < 16) atomicAdd(&t64[row * 64 + lane], (unsigned long long)(it + lane));  // ds_add_u64, 16 active lanes
    if (MODE == 6) atomicAdd(&t64[((row + lane) & 63) * 64 + lane], (unsigned long long)(it + lane));  // ds_add_u64, a different row per lane
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (t64[lane] == 12345 && t32[lane] == 1.5f) out[0] = 1;
}
int main() {
  long long *d; (void)hipMalloc(&d, 8 * 1024);
  const int iters = 20000;
  const char *names[] = {"ds_add_u64 (atomic)", "64-bit plain RMW", "ds_add_f32 (atomic)", "ds_add_u32 (atomic)",
                         "ds_add_u64, 2 lanes", "ds_add_u64, 16 lanes", "ds_add_u64, row per lane"};
  for (int m = 0; m < 7; ++m) {
    for (int grid : {1, 256}) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 32768, 0, d, iters);
      if (m == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 32768, 0, d, iters);
      long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("%-22s grid %3d: %.1f clock64 ticks per wave instruction (4 waves per WG issuing concurrently)\n", names[m], grid,
             (double)h / iters);
    }
  }
  return 0;
}
