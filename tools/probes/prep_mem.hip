// Probe (round 6, review item 1): the MEMORY side of prep_fast32_kernel alone.  Same access pattern as the kernel --
// 32 pixels x C planes in (128-byte pieces, one per channel plane), swizzled LDS transpose, three row streams out
// (emb: 16 B per lane; emb_loc: rows of 1 032 B written as two 8-byte pieces per lane + a tail; fp16 copy: 8 B per
// lane), the per-row xt word, labels, seed labels, the first M-step's partial rows -- and NO arithmetic, no chains,
// no divisions.  If this takes the kernel's 7.4 ms the access pattern is the limit; if it takes the 5.8 ms of the
// plain stream with the same mix (rw_mix.hip) the instruction stream is.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/prep_mem.hip -o /tmp/prep_mem && /tmp/prep_mem
// Variants (template switches): PIX 32 | 64 pixels per workgroup (PIX * 8 threads), LOC 0 = emb_loc as the kernel
// writes it, 1 = the block's rows staged flat in LDS and streamed out as 16-byte pieces, M0 = partial rows, F16 =
// the half-precision copy, LDSPAD = extra LDS to force fewer workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int PIX, int LOC, int M0, int F16, int ORD = 0, int XCD = 0, int H16 = 0>
__global__ __launch_bounds__(PIX * 8) void mimic(const float *__restrict__ in, int C, int64_t HW,
                                                 float *__restrict__ emb, float *__restrict__ emb_loc,
                                                 _Float16 *__restrict__ xh, uint2 *__restrict__ xt,
                                                 int64_t *__restrict__ labels_out, int32_t *__restrict__ klab,
                                                 unsigned long long *__restrict__ part) {
  extern __shared__ float lds[];
  float *tile = lds;                                  // [PIX][C] swizzled, LOC = 1: reused as [PIX][C + 2] flat
  constexpr int NW = PIX / 8;                         // waves
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.y;
  // XCD = 1: workgroup ids go round-robin over the 8 XCDs; give each XCD one contiguous eighth of the image's tiles
  const unsigned bx = XCD ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
  const int64_t q0 = (int64_t)bx * PIX;
  const int D = C + 2, NQ = C >> 2;
  // lanes: pixel jl, sub-quad sub (32 pixels: two quads per wave instruction; 64 pixels: one)
  constexpr int LPP = PIX == 32 ? 32 : 64;
  // (128 pixels: waves 2k / 2k + 1 take the two 64-pixel halves of quad wave k)
  constexpr int NWQ = PIX == 128 ? 8 : NW;
  const int wq = PIX == 128 ? w >> 1 : w;
  const int jl = lane % LPP + (PIX == 128 ? 64 * (w & 1) : 0), sub = lane / LPP;
  constexpr int QPW = 64 / LPP;                       // quads per wave instruction
  const int sw = jl & 15;
  const float *src = in + (int64_t)b * C * HW + q0 + jl;
  constexpr int NU = 8;
  float4 v0[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = QPW * wq + sub + QPW * NWQ * u;     // NQ == QPW * NWQ * NU for C = 256
    v0[u].x = src[(int64_t)(4 * q + 0) * HW];
    v0[u].y = src[(int64_t)(4 * q + 1) * HW];
    v0[u].z = src[(int64_t)(4 * q + 2) * HW];
    v0[u].w = src[(int64_t)(4 * q + 3) * HW];
  }
  const int64_t row0 = (int64_t)b * HW + q0;
  if (tid < PIX) {
    labels_out[row0 + tid] = tid;
    klab[row0 + tid] = tid & 7;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = QPW * wq + sub + QPW * NWQ * u;
    *reinterpret_cast<float4 *>(tile + jl * C + ((q ^ sw) << 2)) = v0[u];
  }
  __syncthreads();
  if (LOC == 0) {
    for (int jj = 0; jj < 8; ++jj) {
      const int j = ORD ? 8 * w + jj : w + NW * jj;     // ORD = 1: a wave owns 8 consecutive rows (8 KiB runs)
      const int64_t row = row0 + j;
      const float *r = tile + j * C;
      float *eo = emb + row * C, *lo = emb_loc + row * D;
      const int sj = j & 15;
      for (int q = lane; q < NQ; q += 64) {
        const float4 v = *reinterpret_cast<const float4 *>(r + ((q ^ sj) << 2));
        *reinterpret_cast<float4 *>(eo + 4 * q) = v;
        *reinterpret_cast<float2 *>(lo + 4 * q) = make_float2(v.x * 2, v.y * 2);
        *reinterpret_cast<float2 *>(lo + 4 * q + 2) = make_float2(v.z * 2, v.w * 2);
        if (F16) {
          const h4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
          *reinterpret_cast<h4 *>(xh + row * C + 4 * q) = hv;
        }
      }
      if (lane == 0) {
        *reinterpret_cast<float2 *>(lo + C) = make_float2(1.f, 2.f);
        if (F16) xt[row] = make_uint2(1u, 2u);
      }
    }
  } else {
    // rows to registers, emb and the fp16 copy straight from them; emb_loc goes back to LDS in its final flat
    // layout (rows of D floats, the whole block contiguous) and leaves as 16-byte pieces
    constexpr int RPW = 8;                             // rows per wave
    float4 rv[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int j = ORD ? 8 * w + i : w + NW * i;
      rv[i] = *reinterpret_cast<const float4 *>(tile + j * C + ((lane ^ (j & 15)) << 2));
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int j = ORD ? 8 * w + i : w + NW * i;
      const int64_t row = row0 + j;
      *reinterpret_cast<float4 *>(emb + row * C + 4 * lane) = rv[i];
      if (F16 && !H16) {
        const h4 hv = {(_Float16)rv[i].x, (_Float16)rv[i].y, (_Float16)rv[i].z, (_Float16)rv[i].w};
        *reinterpret_cast<h4 *>(xh + row * C + 4 * lane) = hv;
      }
    }
    if (F16 && H16 && ORD) {
      // the fp16 copy as 16-byte pieces: two consecutive rows per wave instruction (lanes 0..31 row 2p, 32..63 row
      // 2p + 1; a lane fetches its partner's quad through a shuffle)
#pragma unroll
      for (int p = 0; p < RPW / 2; ++p) {
        const int j = 8 * w + 2 * p + (lane >> 5);
        const float4 a = rv[2 * p], c = rv[2 * p + 1];
        const int sl = 2 * (lane & 31);
        float4 u0, u1;   // quads 2 * (lane & 31), + 1 of this lane's row
        u0.x = __shfl(lane < 32 ? a.x : c.x, sl); u0.y = __shfl(lane < 32 ? a.y : c.y, sl);
        u0.z = __shfl(lane < 32 ? a.z : c.z, sl); u0.w = __shfl(lane < 32 ? a.w : c.w, sl);
        u1.x = __shfl(lane < 32 ? a.x : c.x, sl + 1); u1.y = __shfl(lane < 32 ? a.y : c.y, sl + 1);
        u1.z = __shfl(lane < 32 ? a.z : c.z, sl + 1); u1.w = __shfl(lane < 32 ? a.w : c.w, sl + 1);
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 hv = {(_Float16)u0.x, (_Float16)u0.y, (_Float16)u0.z, (_Float16)u0.w,
                       (_Float16)u1.x, (_Float16)u1.y, (_Float16)u1.z, (_Float16)u1.w};
        *reinterpret_cast<h8 *>(xh + (row0 + j) * C + 8 * (lane & 31)) = hv;
      }
    }
    if (F16 && tid < PIX) xt[row0 + tid] = make_uint2(1u, 2u);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int j = ORD ? 8 * w + i : w + NW * i;
      float *fr = tile + j * D + 4 * lane;             // 8-byte aligned
      *reinterpret_cast<float2 *>(fr) = make_float2(rv[i].x * 2, rv[i].y * 2);
      *reinterpret_cast<float2 *>(fr + 2) = make_float2(rv[i].z * 2, rv[i].w * 2);
    }
    if (tid < PIX) *reinterpret_cast<float2 *>(tile + tid * D + C) = make_float2(1.f, 2.f);
    __syncthreads();
    float4 *dst = reinterpret_cast<float4 *>(emb_loc + row0 * D);   // row0 even here: 16-byte aligned
    const int n4 = PIX * D / 4;
    for (int i = tid; i < n4; i += PIX * 8) dst[i] = *reinterpret_cast<const float4 *>(tile + 4 * i);
  }
  if (M0) {
    unsigned long long *dst = part + ((int64_t)b * gridDim.x + bx) * 2 * D;
    for (int i = tid; i < 2 * D; i += PIX * 8) dst[i] = (unsigned long long)i;
  }
}

int main(int argc, char **argv) {
  const int B = 48, C = 256, D = C + 2;
  const int64_t HW = 448 * 448, N = B * HW;
  float *in, *emb, *emb_loc; _Float16 *xh; uint2 *xt; int64_t *lab; int32_t *klab; unsigned long long *part;
  (void)hipMalloc(&in, N * C * 4); (void)hipMalloc(&emb, N * C * 4); (void)hipMalloc(&emb_loc, N * D * 4 + 64);
  (void)hipMalloc(&xh, N * C * 2); (void)hipMalloc(&xt, N * 8); (void)hipMalloc(&lab, N * 8); (void)hipMalloc(&klab, N * 4);
  (void)hipMalloc(&part, (N / 32) * 2 * D * 8);
  (void)hipMemset(in, 0, N * C * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](const char *name, double gb, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int i = 0; i < 5; ++i) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-58s mean %.3f ms  best %.3f ms  %.0f GB/s (mean)\n", name, sum / 5, best, gb / (sum / 5) * 1e3);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
  };
  const double g_in = N * C * 4 / 1e9, g_loc = N * D * 4 / 1e9, g_h = N * C * 2 / 1e9 + N * 8 / 1e9,
               g_misc = N * 12 / 1e9, g_m0 = (N / 32) * 2.0 * D * 8 / 1e9;
#define RUN(PIX, LOC, M0, F16, PAD, NAME) RUNX(PIX, LOC, M0, F16, 0, 0, 0, PAD, NAME)
#define RUNX(PIX, LOC, M0, F16, ORD, XCD, H16, PAD, NAME)                                                                         \
  {                                                                                                                \
    const size_t sh = (size_t)PIX * (C + 2) * 4 + 1024 + PAD;                                                      \
    (void)hipFuncSetAttribute((const void *)mimic<PIX, LOC, M0, F16, ORD, XCD, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)sh);                                                                            \
    const double gb = 2 * g_in + g_loc + (F16 ? g_h : 0) + g_misc + (M0 ? g_m0 * (32.0 / PIX) : 0);               \
    time(NAME, gb, [&] {                                                                                           \
      hipLaunchKernelGGL((mimic<PIX, LOC, M0, F16, ORD, XCD, H16>), dim3((unsigned)(HW / PIX), B), dim3(PIX * 8), sh, 0, in, C,   \
                         HW, emb, emb_loc, xh, xt, lab, klab, part);                                               \
    });                                                                                                            \
  }
  for (int rep = 0; rep < 2; ++rep) {
    RUN(32, 0, 1, 1, 4096, "32 px, kernel's stores, partials, fp16 (= prep)       4 wg/cu");
    RUN(32, 0, 0, 1, 4096, "32 px, kernel's stores, no partials, fp16             4 wg/cu");
    RUN(32, 0, 0, 0, 4096, "32 px, kernel's stores, no partials, no fp16          4 wg/cu");
    RUN(32, 1, 1, 1, 4096, "32 px, emb_loc flat 16-B pieces, partials, fp16       4 wg/cu");
    RUN(32, 1, 0, 1, 4096, "32 px, emb_loc flat 16-B pieces, no partials, fp16    4 wg/cu");
    RUN(32, 0, 1, 1, 0, "32 px, kernel's stores, partials, fp16 (LDS 34 KB)    4 wg/cu");
    RUN(32, 0, 1, 1, 20480, "32 px, kernel's stores, partials, fp16 (LDS 54 KB)    2 wg/cu");
    RUN(64, 0, 1, 1, 0, "64 px (512 thr), kernel's stores, partials, fp16      2 wg/cu");
    RUN(64, 1, 1, 1, 0, "64 px (512 thr), flat emb_loc, partials, fp16         2 wg/cu");
    RUN(64, 1, 0, 1, 0, "64 px (512 thr), flat emb_loc, no partials, fp16      2 wg/cu");
    RUNX(32, 0, 1, 1, 1, 0, 0, 4096, "32 px, kernel's stores, 8 consecutive rows per wave   4 wg/cu");
    RUNX(32, 1, 1, 1, 1, 0, 0, 4096, "32 px, flat emb_loc, 8 consecutive rows per wave      4 wg/cu");
    RUNX(32, 1, 1, 1, 1, 0, 1, 4096, "32 px, flat, consecutive rows, fp16 as 16-B pieces    4 wg/cu");
    RUNX(32, 1, 1, 1, 1, 1, 1, 4096, "32 px, flat, consec., fp16 16 B, XCD-contiguous tiles 4 wg/cu");
    RUNX(32, 0, 1, 1, 0, 1, 0, 4096, "32 px, kernel's stores, XCD-contiguous tiles          4 wg/cu");
    RUNX(32, 1, 1, 1, 0, 1, 0, 4096, "32 px, flat emb_loc, XCD-contiguous (= round-6 kernel) 4 wg/cu");
    RUNX(32, 1, 1, 1, 1, 1, 0, 4096, "32 px, flat, consecutive rows, XCD-contiguous          4 wg/cu");
    RUNX(32, 1, 0, 1, 0, 1, 0, 4096, "32 px, flat, XCD-contiguous, no partials               4 wg/cu");
    RUNX(64, 1, 1, 1, 1, 0, 1, 0, "64 px, flat, consecutive rows, fp16 as 16-B pieces    2 wg/cu");
    RUNX(64, 1, 1, 1, 1, 1, 1, 0, "64 px, flat, consec., fp16 16 B, XCD-contiguous tiles 2 wg/cu");
    RUNX(128, 1, 1, 1, 1, 0, 1, 0, "128 px (1024 thr), flat, consec., fp16 16 B           1 wg/cu");
  }
  return 0;
}
