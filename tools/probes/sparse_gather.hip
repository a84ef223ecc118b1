// Probe: achievable HBM rate when only a fraction of the 1032-byte rows is read (the exact-sum
// M-step update reads the rows whose label changed).  256 persistent workgroups x 8 waves, each
// wave walks its own contiguous row range and reads every row r with jitter(r) spacing (mean gap
// G/2 + 1 rows), whole row per load instruction group (lane x 16 B + 8-byte tail), 16 rows in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/sparse_gather.hip -o /tmp/sg && /tmp/sg
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ inline unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int U>
__global__ __launch_bounds__(512) void gather(const float *x, long rows, int G, float *sink, unsigned long long *count) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long per = rows / (gridDim.x * 8);
  const long r0 = ((long)blockIdx.x * 8 + w) * per, r1 = r0 + per;
  float s = 0.f;
  long r = r0;
  unsigned it = (unsigned)r0;
  unsigned long long n = 0;
  auto next = [&]() { const long c = r; r += 1 + (G > 0 ? hsh(it++) % (unsigned)(G + 1) : 0); return c < r1 ? c : r1 - 1; };
  f4 va[U], vb[U]; float ta[U], tb[U];
  auto issue = [&](f4 (&v)[U], float (&t)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r < r1) ++n;
      const float *src = x + next() * 258;
      v[u] = *(const f4 *)(src + 4 * lane);
      t[u] = src[256 + (lane & 1)];
    }
  };
  auto fold = [&](const f4 (&v)[U], const float (&t)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u][0] + v[u][1] + v[u][2] + v[u][3] + t[u];
  };
  issue(va, ta);
  while (r < r1) {
    issue(vb, tb);
    __builtin_amdgcn_sched_barrier(0);
    fold(va, ta);
    __builtin_amdgcn_sched_barrier(0);
    issue(va, ta);
    __builtin_amdgcn_sched_barrier(0);
    fold(vb, tb);
    __builtin_amdgcn_sched_barrier(0);
  }
  fold(va, ta);
  if (s == 123.456f) *sink = s;
  if (lane == 0) atomicAdd(count, n);
}
int main() {
  const long rows = 48L * 448 * 448;
  float *x, *sink; unsigned long long *cnt;
  (void)hipMalloc(&x, (size_t)rows * 1032); (void)hipMalloc(&sink, 4); (void)hipMalloc(&cnt, 8);
  (void)hipMemset(x, 0, (size_t)rows * 1032);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int Gs[] = {0, 1, 2, 5, 12, 18, 38, 76};
  for (int G : Gs) {
    float best = 1e9f; unsigned long long h = 0;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipMemset(cnt, 0, 8);
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(gather<8>, dim3(256), dim3(512), 0, 0, x, rows, G, sink, cnt);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      (void)hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost);
    }
    printf("G %3d  rows read %9llu (%.1f %%)  %.3f ms  %.0f GB/s useful\n", G, h, 100.0 * h / rows, best, h * 1032.0 / best / 1e6);
  }
  return 0;
}
