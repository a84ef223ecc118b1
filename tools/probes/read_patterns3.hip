// Probe 3 (round 4): which LOAD SHAPE streams the fp16 row copy (rows of 512 B, a wave owns tiles of 32 rows =
// 16 KiB contiguous) fastest, on one box, same grid as the E-step (256 persistent workgroups x 8 waves, a
// contiguous row range each), four 4-KiB sets in flight per wave, counted waits:
//   p1   dwordx2, 4 rows x 128 B per instruction, 8 instructions per set     (the round-1..3 engine)
//   p1n  the same with nt
//   p2   dwordx4, 8 rows x 128 B per instruction, 4 instructions per set, nt (same 128-B chunk order)
//   p3   dwordx4, 1 KiB contiguous per instruction (2 rows), 4 per set, nt    (whole-row order)
//   p5   dwordx4, 32 rows x 32 B per instruction (lane j / j + 32 = the two 16-B halves of row j's 32 bytes: the
//        MFMA 32x32x16 B-operand layout, rows straight into registers with no LDS staging), plain and nt
//   p4   p2's shape as LDS-DMA (global_load_lds_dwordx4), 2 or 3 sets in flight per wave
//   hipcc --offload-arch=gfx950 -O3 tools/probes/read_patterns3.hip -o tools/probes/read_patterns3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAIT4(N, P) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]) : "n"(N))
#define WAIT8(N, P) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]) : "n"(N))

// MODE 0: p1, 1: p1n, 2: p2, 3: p3, 4: p5, 5: p5n
template <int MODE>
__global__ __launch_bounds__(512) void pat(const char *__restrict__ x, long rows, unsigned *sink) {
  constexpr int NW = 8, RB = 512;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long ntiles = rows / 32;
  const long per = ntiles / gridDim.x / NW * NW;
  const long t0 = blockIdx.x * per + w, t1 = (blockIdx.x + 1) * per;
  const long nt = t0 < t1 ? (t1 - t0 + NW - 1) / NW : 0;
  const long nsets = nt * 4;                       // a set = 4 KiB of the wave's tile
  unsigned acc = 0;
  long ld = 0;
  if constexpr (MODE <= 1) {
    const unsigned voff = (unsigned)((lane >> 4) * RB + (lane & 15) * 8);
    auto load = [&](u32x2 (&b)[8]) {
      const long tile = t0 + (ld >> 2) * NW;
      const char *base = x + tile * 32 * RB + (ld & 3) * 128;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(b[i]) : "v"(voff), "s"(base + (long)i * 4 * RB));
        else asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(b[i]) : "v"(voff), "s"(base + (long)i * 4 * RB));
      }
      ++ld;
    };
    u32x2 A[8], B[8], C[8], D[8];
    load(A); load(B); load(C); load(D);
    for (long s = 0; s < nsets; s += 4) {
#define STEP(P) WAIT8(24, P); _Pragma("unroll") for (int i = 0; i < 8; ++i) acc ^= P[i].x ^ P[i].y; load(P);
      STEP(A) STEP(B) STEP(C) STEP(D)
#undef STEP
    }
    WAIT8(0, A); WAIT8(0, B); WAIT8(0, C); WAIT8(0, D);
  } else {
    const unsigned voff = MODE == 2 ? (unsigned)((lane >> 3) * RB + (lane & 7) * 16)
                          : MODE >= 4 ? (unsigned)((lane & 31) * RB + (lane >> 5) * 16) : (unsigned)(lane * 16);
    auto load = [&](u32x4 (&b)[4]) {
      const long tile = t0 + (ld >> 2) * NW;
      const char *base = x + tile * 32 * RB + (MODE == 2 || MODE >= 4 ? (ld & 3) * 128 : (ld & 3) * 4096);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[i]) : "v"(voff), "s"(base + (long)i * 32));
        else if constexpr (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(b[i]) : "v"(voff), "s"(base + (long)i * 32));
        else asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(b[i]) : "v"(voff), "s"(base + (long)i * (MODE == 2 ? 8 * RB : 1024)));
      }
      ++ld;
    };
    u32x4 A[4], B[4], C[4], D[4];
    load(A); load(B); load(C); load(D);
    for (long s = 0; s < nsets; s += 4) {
#define STEP(P) WAIT4(12, P); _Pragma("unroll") for (int i = 0; i < 4; ++i) acc ^= P[i].x ^ P[i].y ^ P[i].z ^ P[i].w; load(P);
      STEP(A) STEP(B) STEP(C) STEP(D)
#undef STEP
    }
    WAIT4(0, A); WAIT4(0, B); WAIT4(0, C); WAIT4(0, D);
  }
  if (acc == 0x12345677u) sink[0] = acc;
}

// p4: LDS-DMA, a ring of NB 4-KiB buffers per wave, NB - 1 sets in flight
template <int NB>
__global__ __launch_bounds__(512) void pat_lds(const char *__restrict__ x, long rows, unsigned *sink) {
  constexpr int NW = 8, RB = 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long ntiles = rows / 32;
  const long per = ntiles / gridDim.x / NW * NW;
  const long t0 = blockIdx.x * per + w, t1 = (blockIdx.x + 1) * per;
  const long nt = t0 < t1 ? (t1 - t0 + NW - 1) / NW : 0;
  const long nsets = nt * 4;
  unsigned char *ring = lds + (size_t)w * NB * 4096;
  const unsigned voff = (unsigned)((lane >> 3) * RB + (lane & 7) * 16);
  long ld = 0;
  auto load = [&](int buf) {
    const long tile = t0 + (ld >> 2) * NW;
    const char *base = x + tile * 32 * RB + (ld & 3) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char *g = base + (long)i * 8 * RB + voff;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)g,
                                       (void __attribute__((address_space(3))) *)(ring + buf * 4096 + i * 1024), 16, 0, 0);
    }
    ++ld;
  };
  unsigned acc = 0;
  for (int p = 0; p < NB - 1; ++p) load(p);
  int buf = 0;
  for (long s = 0; s < nsets; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * 4) : "memory");
    const u32x4 v = *reinterpret_cast<const u32x4 *>(ring + buf * 4096 + lane * 16);
    const u32x4 v2 = *reinterpret_cast<const u32x4 *>(ring + buf * 4096 + 2048 + lane * 16);
    acc ^= v.x ^ v.y ^ v.z ^ v.w ^ v2.x ^ v2.w;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int nb = buf + NB - 1; if (nb >= NB) nb -= NB;
    load(nb);
    buf = buf + 1 == NB ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345677u) sink[0] = acc;
}

int main() {
  const long rows = 48L * 448 * 448;
  unsigned *sink; char *x;
  if (hipMalloc(&x, (size_t)(rows + 4096) * 512) != hipSuccess) return 1;
  (void)hipMalloc(&sink, 64);
  (void)hipMemset(x, 1, (size_t)(rows + 4096) * 512);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto time = [&](const char *name, auto launch) {
    launch(); launch();
    (void)hipDeviceSynchronize();
    float best = 1e9f, tot = 0;
    for (int r = 0; r < 3; ++r) {
      (void)hipEventRecord(a);
      for (int i = 0; i < 5; ++i) launch();
      (void)hipEventRecord(b);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      best = ms / 5 < best ? ms / 5 : best; tot += ms / 5;
    }
    const double bytes = (double)rows * 512;
    printf("%-72s best %.3f ms  %.0f GB/s   mean %.3f ms\n", name, best, bytes / best / 1e6, tot / 3);
  };
  for (int rep = 0; rep < 2; ++rep) {
    for (int grid : {256, 512}) {
      printf("grid %d x 512 threads\n", grid);
      time("p1  dwordx2, 4 rows x 128 B / instr", [&] { hipLaunchKernelGGL(pat<0>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      time("p1n dwordx2 nt, 4 rows x 128 B / instr", [&] { hipLaunchKernelGGL(pat<1>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      time("p2  dwordx4 nt, 8 rows x 128 B / instr", [&] { hipLaunchKernelGGL(pat<2>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      time("p3  dwordx4 nt, 1 KiB contiguous / instr", [&] { hipLaunchKernelGGL(pat<3>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      time("p5  dwordx4, 32 rows x 32 B / instr (MFMA B layout)", [&] { hipLaunchKernelGGL(pat<4>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      time("p5n dwordx4 nt, 32 rows x 32 B / instr (MFMA B layout)", [&] { hipLaunchKernelGGL(pat<5>, dim3(grid), dim3(512), 0, 0, x, rows, sink); });
      if (grid == 256) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pat_lds<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * 4096);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(pat_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 4096);
        time("p4  LDS-DMA dwordx4, 8 rows x 128 B / instr, ring 3 (2 in flight)", [&] { hipLaunchKernelGGL(pat_lds<3>, dim3(grid), dim3(512), 8 * 3 * 4096, 0, x, rows, sink); });
        time("p4  LDS-DMA dwordx4, 8 rows x 128 B / instr, ring 4 (3 in flight)", [&] { hipLaunchKernelGGL(pat_lds<4>, dim3(grid), dim3(512), 8 * 4 * 4096, 0, x, rows, sink); });
      }
    }
  }
  return 0;
}
