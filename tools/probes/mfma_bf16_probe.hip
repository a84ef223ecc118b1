// Probe: how does v_mfma_f32_32x32x16_bf16 round its 16-term dot product + C?
// Compares D against (a) the exact sum (double) rounded once to f32 and (b) a
// sequential f32 fma/add chain, for random bf16 operands and a random f32 C.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_bf16_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const uint16_t *A, const uint16_t *B, const float *C, float *D) {
  // A [32][16] row-major bf16, B [16][32] (k-major) bf16, C/D [32][32]
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (short)A[i * 16 + 8 * g + t]; b[t] = (short)B[(8 * g + t) * 32 + i]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i];
  f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = d[r];
}
static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  uint16_t hA[512], hB[512]; float hC[1024], hD[1024];
  uint16_t *dA, *dB; float *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  srand(1);
  long n = 0, eq_exact = 0, eq_seq = 0, eq_seq_rev = 0; double max_rel_exact = 0;
  for (int trial = 0; trial < 200; ++trial) {
    for (int i = 0; i < 512; ++i) {
      // magnitudes spread over a few binades, random signs
      float fa = ((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, -(rand() % 6));
      float fb = ((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, -(rand() % 6));
      uint32_t ua, ub; memcpy(&ua, &fa, 4); memcpy(&ub, &fb, 4);
      hA[i] = ua >> 16; hB[i] = ub >> 16;
    }
    for (int i = 0; i < 1024; ++i) hC[i] = (trial & 1) ? ((rand() % 2001) - 1000) / 700.0f : 0.0f;
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ex = hC[i * 32 + j]; float sq = hC[i * 32 + j], sr = hC[i * 32 + j];
      for (int t = 0; t < 16; ++t) { ex += (double)bf(hA[i * 16 + t]) * bf(hB[t * 32 + j]); sq = fmaf(bf(hA[i * 16 + t]), bf(hB[t * 32 + j]), sq); }
      for (int t = 15; t >= 0; --t) sr = fmaf(bf(hA[i * 16 + t]), bf(hB[t * 32 + j]), sr);
      float d = hD[i * 32 + j];
      ++n; eq_exact += (d == (float)ex); eq_seq += (d == sq); eq_seq_rev += (d == sr);
      double rel = fabs((double)d - ex) / (fabs(ex) + 1e-30);
      // error relative to sum of |terms|
      double sabs = fabs(hC[i * 32 + j]); for (int t = 0; t < 16; ++t) sabs += fabs((double)bf(hA[i * 16 + t]) * bf(hB[t * 32 + j]));
      double rel2 = fabs((double)d - ex) / (sabs + 1e-30);
      if (rel2 > max_rel_exact) max_rel_exact = rel2;
      (void)rel;
    }
  }
  printf("outputs %ld  == exact-rounded-once %ld (%.2f%%)  == seq fma chain %ld (%.2f%%)  == reverse chain %ld\n", n,
         eq_exact, 100.0 * eq_exact / n, eq_seq, 100.0 * eq_seq / n, eq_seq_rev);
  printf("max |D - exact| / sum|terms| = %.3e  (2^-24 = %.3e)\n", max_rel_exact, ldexp(1.0, -24));
  return 0;
}
