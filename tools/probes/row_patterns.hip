// Probe: achievable HBM read rate of the M-step access pattern on [rows][258] float32
// (whole rows per wave: one 16 B-per-lane load = 1024 B, plus a 2-lane tail load), rows of a
// chunk dealt to the 4 waves of a workgroup in runs of RUN consecutive rows.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/row_patterns.hip -o tools/probes/row_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int D = 258;
constexpr int CHUNK = 2048;
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
// MODE 0: x4 + tail dword; 1: x4 only (256 cols); 2: two x2 loads (lanes cover 128 floats each) + tail
// 3: x4 + tail, all waves of the WG walk the SAME run (row = run*RUN + i*NW + w) -> adjacent rows in flight
template <int MODE, int UNROLL, int NW, int RUN>
__global__ __launch_bounds__(NW * 64) void rowpat(const float *x, long rows, float *sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long nchunks = rows / CHUNK;
  float s = 0.f;
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float *xr = x + c * CHUNK * D;
    // rows of this wave: runs q = w, w+NW, ... of RUN rows
    constexpr int PER = CHUNK / NW;     // rows per wave per chunk
    f4 va[UNROLL], vb[UNROLL];
    float ta[UNROLL], tb[UNROLL];
    auto rowof = [&](int i) {
      if (MODE == 3) return i * NW + w;
      const int q = i / RUN, o = i % RUN;
      return (q * NW + w) * RUN + o;
    };
    auto issue = [&](int i0, f4 (&v)[UNROLL], float (&t)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int r = rowof(i0 + u);
        if (MODE == 2) {
          f2 a = *(const f2 *)(xr + (long)r * D + lane * 2);
          f2 b = *(const f2 *)(xr + (long)r * D + 128 + lane * 2);
          v[u] = f4{a.x, a.y, b.x, b.y};
        } else
          v[u] = *(const f4 *)(xr + (long)r * D + lane * 4);
        if (MODE != 1) t[u] = lane < 2 ? xr[(long)r * D + 256 + lane] : 0.f;
        else t[u] = 0.f;
      }
    };
    auto fold = [&](const f4 (&v)[UNROLL], const float (&t)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w + t[u];
    };
    issue(0, va, ta);
    for (int i = 0; i < PER; i += 2 * UNROLL) {
      issue(i + UNROLL, vb, tb);
      __builtin_amdgcn_sched_barrier(0);
      fold(va, ta);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 * UNROLL < PER) issue(i + 2 * UNROLL, va, ta);
      __builtin_amdgcn_sched_barrier(0);
      fold(vb, tb);
      __builtin_amdgcn_sched_barrier(0);
    }
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
    printf("%-52s %.3f ms  %.0f GB/s\n", name, ms / 5, bytes / (ms / 5) / 1e6);
  };
#define T(name, MODE, UNROLL, NW, RUN, GRID) \
  time(name, [&] { hipLaunchKernelGGL((rowpat<MODE, UNROLL, NW, RUN>), dim3(GRID), dim3(NW * 64), 0, 0, x, rows, sink); })
  T("x4+tail  U16 4w run64 grid 512", 0, 16, 4, 64, 512);
  T("x4+tail  U16 4w run64 grid 4704 (1 chunk/WG)", 0, 16, 4, 64, 4704);
  T("x4+tail  U16 4w run64 grid 1024", 0, 16, 4, 64, 1024);
  T("x4 only  U16 4w run64 grid 512", 1, 16, 4, 64, 512);
  T("2*x2+tail U16 4w run64 grid 512", 2, 16, 4, 64, 512);
  T("x4+tail  U16 4w interleaved rows grid 512", 3, 16, 4, 64, 512);
  T("x4+tail  U8  4w run64 grid 512", 0, 8, 4, 64, 512);
  T("x4+tail  U8  4w run64 grid 1024", 0, 8, 4, 64, 1024);
  T("x4+tail  U8  8w run64 grid 512", 0, 8, 8, 64, 512);
  T("x4+tail  U16 8w run64 grid 256", 0, 16, 8, 64, 256);
  T("x4+tail  U16 8w run64 grid 512", 0, 16, 8, 64, 512);
  T("x4+tail  U16 8w interleaved grid 512", 3, 16, 8, 64, 512);
  T("x4+tail  U16 4w run16 grid 512", 0, 16, 4, 16, 512);
  return 0;
}
