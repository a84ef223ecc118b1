// Calibration of rocprofv3 FETCH_SIZE on gfx950 for the access widths libhsgk
// uses (MI355X_MICROARCH.md section HBM: 16 B/lane streams count at exactly 1/2;
// other widths must be calibrated on a known byte count).  Each kernel streams
// BYTES bytes exactly once with W bytes per lane.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/fetch_calib.hip -o tools/probes/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o c -- tools/probes/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ void stream_read(const T *p, size_t n, float *sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = p[i];
    const float *f = reinterpret_cast<const float *>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) s += f[k];
  }
  if (s == 123.456f) *sink = s;
}
// rows of 258 floats read as 8-byte pieces, 16 lanes per 128-byte segment of a
// row (the E-step staging pattern)
__global__ void rows_read8(const float *p, size_t rows, float *sink) {
  float s = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (size_t r0 = ((size_t)blockIdx.x * 8 + w) * 32; r0 < rows; r0 += (size_t)gridDim.x * 8 * 32)
    for (int q = 0; q < 8; ++q)
      for (int i = 0; i < 8; ++i) {
        const size_t row = r0 + (lane >> 4) + 4 * i;
        if (row < rows) {
          const float2 v = *reinterpret_cast<const float2 *>(p + row * 258 + q * 32 + 2 * (lane & 15));
          s += v.x + v.y;
        }
      }
  if (s == 123.456f) *sink = s;
}
int main() {
  const size_t BYTES = (size_t)4 << 30;
  float *buf, *sink;
  hipMalloc(&buf, BYTES);
  hipMalloc(&sink, 4);
  hipMemset(buf, 0, BYTES);
  hipLaunchKernelGGL(stream_read<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, BYTES / 16, sink);
  hipLaunchKernelGGL(stream_read<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, BYTES / 8, sink);
  hipLaunchKernelGGL(stream_read<float>, dim3(4096), dim3(256), 0, 0, (const float *)buf, BYTES / 4, sink);
  const size_t rows = BYTES / (258 * 4);
  hipLaunchKernelGGL(rows_read8, dim3(4096), dim3(512), 0, 0, buf, rows, sink);
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu (rows kernel: %zu = rows*256 floats)\n", BYTES, rows * 256 * 4);
  return 0;
}
