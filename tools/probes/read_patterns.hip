// Probe: achievable HBM read rate of the row-tile access patterns considered for
// the E-step, on [rows][258] float32 data (rows 1032 B apart, 8-byte aligned).
//   A  32 rows x 128 B per step (current split engine: lane -> (row = lane>>4 + 4i, 8 B))
//   B  16 rows x 256 B per step
//   C   8 rows x 512 B per step
//   D  linear 16 B per lane over the whole buffer (upper bound)
// Every wave walks its own rows, 8 steps per row tile (A), prefetch depth PD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_patterns.hip -o tools/probes/read_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int D = 258;
template <int ROWS_PER_STEP, int PD>
__global__ __launch_bounds__(512) void pat(const float *x, long rows, float *sink) {
  // per step a wave reads ROWS_PER_STEP rows x (4096 / ROWS_PER_STEP) bytes
  constexpr int BYTES_PER_ROW = 4096 / ROWS_PER_STEP;       // 128 / 256 / 512
  constexpr int LANES_PER_ROW = BYTES_PER_ROW / 8;          // float2 per lane
  constexpr int ROWS_PER_LOAD = 64 / LANES_PER_ROW;
  constexpr int LOADS = ROWS_PER_STEP / ROWS_PER_LOAD;      // == 8 always
  constexpr int STEPS = 1024 / BYTES_PER_ROW;               // steps to cover 256 floats of a row
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const int lrow = lane / LANES_PER_ROW, lcol = (lane % LANES_PER_ROW) * 2;
  float s = 0.f;
  float2 buf[PD][LOADS];
  const long ntiles = rows / ROWS_PER_STEP;
  for (long t = wave; t < ntiles; t += nwaves) {
    const float *base = x + t * ROWS_PER_STEP * D;
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
      for (int i = 0; i < LOADS; ++i)
        buf[p][i] = *(const float2 *)(base + (long)(lrow + ROWS_PER_LOAD * i) * D + p * (BYTES_PER_ROW / 4) + lcol);
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
#pragma unroll
      for (int i = 0; i < LOADS; ++i) s += buf[st % PD][i].x + buf[st % PD][i].y;
      if (st + PD < STEPS) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i)
          buf[st % PD][i] = *(const float2 *)(base + (long)(lrow + ROWS_PER_LOAD * i) * D + (st + PD) * (BYTES_PER_ROW / 4) + lcol);
      }
    }
  }
  if (s == 123.456f) *sink = s;
}
__global__ void fill_random(float *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = (float)(int)(z & 0xffff) * (1.0f / 65536.0f) - 0.5f;
  }
}
__global__ void lin(const float4 *p, size_t n, float *sink) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}
int main() {
  const long rows = 48L * 448 * 448;
  const size_t bytes = (size_t)rows * D * 4;
  float *x, *sink;
  hipMalloc(&x, bytes);
  hipMalloc(&sink, 4);
  hipMemset(x, 0, bytes);
  if (getenv("RANDOM_FILL")) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, x, bytes / 4); printf("random fill\n"); }
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto time = [&](const char *name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-34s %.3f ms  %.0f GB/s\n", name, ms / 5, bytes / (ms / 5) / 1e6 * (256.0 / 258.0));
  };
  time("A 32 rows x 128 B, depth 4", [&] { hipLaunchKernelGGL((pat<32, 4>), dim3(256), dim3(512), 0, 0, x, rows, sink); });
  time("A 32 rows x 128 B, depth 4, 2 WG/CU", [&] { hipLaunchKernelGGL((pat<32, 4>), dim3(512), dim3(512), 0, 0, x, rows, sink); });
  time("B 16 rows x 256 B, depth 4", [&] { hipLaunchKernelGGL((pat<16, 4>), dim3(256), dim3(512), 0, 0, x, rows, sink); });
  time("B 16 rows x 256 B, depth 4, 2 WG/CU", [&] { hipLaunchKernelGGL((pat<16, 4>), dim3(512), dim3(512), 0, 0, x, rows, sink); });
  time("C  8 rows x 512 B, depth 2", [&] { hipLaunchKernelGGL((pat<8, 2>), dim3(256), dim3(512), 0, 0, x, rows, sink); });
  time("C  8 rows x 512 B, depth 2, 2 WG/CU", [&] { hipLaunchKernelGGL((pat<8, 2>), dim3(512), dim3(512), 0, 0, x, rows, sink); });
  time("D linear float4, 2048 WGs", [&] { hipLaunchKernelGGL(lin, dim3(2048), dim3(256), 0, 0, (const float4 *)x, bytes / 16, sink); });
  return 0;
}
