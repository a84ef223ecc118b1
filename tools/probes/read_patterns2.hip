// Probe 2: does the DISTRIBUTION of row tiles over workgroups matter for the achievable HBM
// read rate?  Rows of RB bytes (1032 = fp32 x 258, 528 = fp16 x 264); a wave reads tiles of 32
// rows, 128 B per row per step (8 B per lane, 4 rows per load instruction, 8 loads per step),
// prefetch depth 4 steps across tile boundaries.
//   LAYOUT 0: tiles dealt round-robin to all waves of the chip (chip sweeps one window)
//   LAYOUT 1: every workgroup owns one contiguous range of rows (persistent E-step)
//   LAYOUT 2: workgroup per 2048-row chunk, chunks dealt round-robin to workgroups
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_patterns2.hip -o /tmp/rp2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int LAYOUT, int NW>
__global__ __launch_bounds__(NW * 64) void pat(const char *x, long rows, int RB, float *sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int steps = RB / 128;                 // full 128-byte steps per row
  const long ntiles = rows / 32;              // wave tiles
  long t0, t1, tstride;
  if (LAYOUT == 0) { t0 = (long)blockIdx.x * NW + w; t1 = ntiles; tstride = (long)gridDim.x * NW; }
  else if (LAYOUT == 1) {
    const long per = ntiles / gridDim.x / NW * NW;          // tiles per workgroup
    t0 = blockIdx.x * per + w; t1 = (blockIdx.x + 1) * per; tstride = NW;
  } else { t0 = 0; t1 = 0; tstride = 1; }
  const int lrow = lane >> 4, lcol = (lane & 15) * 8;
  float s = 0.f;
  uint2 buf[4][8];
  auto run = [&](long ta, long tb, long ts) {
    const long nt = ta < tb ? (tb - ta + ts - 1) / ts : 0;
    const long nsteps = nt * steps;
    auto load = [&](long gi, uint2 (&b)[8]) {
      const long tile = ta + (gi / steps) * ts;
      const int st = (int)(gi % steps);
      const char *base = x + tile * 32 * RB + st * 128 + lcol;
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = *(const uint2 *)(base + (long)(lrow + 4 * i) * RB);
    };
    for (int p = 0; p < 4; ++p) if (p < nsteps) load(p, buf[p]);
    for (long gi = 0; gi < nsteps; gi += 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s += __uint_as_float(buf[p][i].x) + __uint_as_float(buf[p][i].y);
        if (gi + p + 4 < nsteps) load(gi + p + 4, buf[p]);
      }
    }
  };
  if (LAYOUT == 2) {
    const long nchunks = rows / 2048;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) run(c * 64 + w, (c + 1) * 64, NW);
  } else run(t0, t1, tstride);
  if (s == 123.456f) *sink = s;
}
int main() {
  const long rows = 48L * 448 * 448;
  float *sink; char *x;
  (void)hipMalloc(&x, (size_t)rows * 1032);
  (void)hipMalloc(&sink, 4);
  (void)hipMemset(x, 0, (size_t)rows * 1032);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char *name, int RB, auto launch) {
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)rows * (RB / 128) * 128;
    printf("%-58s RB %4d  %.3f ms  %.0f GB/s\n", name, RB, ms / 5, bytes / (ms / 5) / 1e6);
  };
#define T(name, L, NW, GRID, RB) time(name, RB, [&] { hipLaunchKernelGGL((pat<L, NW>), dim3(GRID), dim3(NW * 64), 0, 0, x, rows, RB, sink); })
  for (int RB : {1032, 1024, 528, 512}) {
    T("round-robin tiles, 8 waves, 256 WGs", 0, 8, 256, RB);
    T("round-robin tiles, 8 waves, 512 WGs", 0, 8, 512, RB);
    T("contiguous range per WG, 8 waves, 256 WGs", 1, 8, 256, RB);
    T("contiguous range per WG, 8 waves, 512 WGs", 1, 8, 512, RB);
    T("contiguous range per WG, 8 waves, 1024 WGs", 1, 8, 1024, RB);
    T("chunk per WG round-robin, 8 waves, 256 WGs", 2, 8, 256, RB);
    T("chunk per WG round-robin, 8 waves, 512 WGs", 2, 8, 512, RB);
    T("chunk per WG one-shot, 8 waves, 4704 WGs", 2, 8, 4704, RB);
    T("round-robin tiles, 4 waves, 512 WGs", 0, 4, 512, RB);
    T("contiguous range per WG, 4 waves, 512 WGs", 1, 4, 512, RB);
  }
  return 0;
}
