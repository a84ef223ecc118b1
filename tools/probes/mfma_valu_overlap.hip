// Do vector instructions overlap matrix instructions on a gfx950 SIMD -- (a) across the two waves of a SIMD,
// (b) inside one wave?   hipcc -O2 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
// One workgroup per CU, 8 waves (2 per SIMD) or 4 (1 per SIMD).  Modes: waves do `nm` matrix instructions
// (v_mfma_f32_32x32x16_f16, independent accumulators) and / or `nv` independent v_med3 per loop iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: every wave MFMA only; 1: every wave VALU only; 2: waves 0-3 MFMA, 4-7 VALU; 3: every wave MFMA phase then VALU phase; 4: every wave interleaved
__global__ __launch_bounds__(512) void probe(float *out, int iters) {
  const int w = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc0 = {0}, acc1 = {0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
  float lo = 1.0f, hi = 1000.0f;
  const bool do_m = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && w < 4);
  const bool do_v = MODE == 1 || MODE == 3 || MODE == 4 || (MODE == 2 && w >= 4);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {                          // 16 MFMAs + 128 VALU interleaved 1 : 8
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], lo, hi);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
        for (int i = 8; i < 16; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], lo, hi);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_fmed3f(v[i], lo, hi);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float *out, int iters, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  // per iteration and wave: 16 MFMAs (32x32x16 f16) and / or 128 v_med3
  for (int threads : {512, 256}) {
    printf("%d waves per workgroup (%d per SIMD), %d iterations of 16 MFMA and / or 128 VALU per wave:\n", threads / 64, threads / 256, iters);
    printf("  every wave MFMA only            %.3f ms\n", run<0>(out, iters, threads));
    printf("  every wave VALU only            %.3f ms\n", run<1>(out, iters, threads));
    if (threads == 512) printf("  waves 0-3 MFMA, waves 4-7 VALU  %.3f ms\n", run<2>(out, iters, threads));
    printf("  every wave MFMA then VALU       %.3f ms\n", run<3>(out, iters, threads));
    printf("  every wave interleaved 1 : 8    %.3f ms\n", run<4>(out, iters, threads));
  }
  return 0;
}
