// Probe: accumulation error of v_mfma_f32_32x32x16_f16 (16 fp16 products + fp32 C) against the
// exact sum, relative to sum|terms| -- the constant used in the error bound of the fp16 filter
// engine (score_tiles_f16.h).  Operands cover normal and subnormal fp16 magnitudes.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f16_probe.hip -o /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
  const int l = threadIdx.x, i = l & 31, g = l >> 5;
  f16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = A[i * 16 + 8 * g + t]; b[t] = B[(8 * g + t) * 32 + i]; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i];
  f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = d[r];
}
__global__ void cvt(const float *in, float *out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (float)(_Float16)in[i];          // device fp32 -> fp16 conversion (RNE, subnormals kept?)
}
int main() {
  {
    float hin[8] = {3.1e-6f, 1.234e-5f, 5.55e-5f, 6.2e-5f, 0.33333334f, 0.99951172f, 1.0004883f, 5.9e-8f}, hout[8];
    float *di, *dout;
    (void)hipMalloc(&di, sizeof hin); (void)hipMalloc(&dout, sizeof hout);
    (void)hipMemcpy(di, hin, sizeof hin, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, di, dout, 8);
    (void)hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
      printf("cvt %.9g -> device %.9g host %.9g %s\n", hin[i], hout[i], (float)(_Float16)hin[i],
             hout[i] == (float)(_Float16)hin[i] ? "same" : "DIFFERENT");
  }
  _Float16 hA[512], hB[512]; float hC[1024], hD[1024];
  _Float16 *dA, *dB; float *dC, *dD;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dC, sizeof hC); (void)hipMalloc(&dD, sizeof hD);
  srand(1);
  long n = 0, eq_exact = 0; double worst = 0;
  for (int trial = 0; trial < 400; ++trial) {
    const int spread = (trial % 4 == 3) ? 20 : 6;        // every 4th trial reaches fp16 subnormals
    for (int i = 0; i < 512; ++i) {
      hA[i] = (_Float16)(((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, -(rand() % spread)));
      hB[i] = (_Float16)(((rand() % 2001) - 1000) / 1000.0f * ldexpf(1.0f, -(rand() % spread)));
    }
    for (int i = 0; i < 1024; ++i) hC[i] = (trial & 1) ? ((rand() % 2001) - 1000) / 700.0f : 0.0f;
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ex = hC[i * 32 + j], sabs = fabs((double)hC[i * 32 + j]);
      for (int t = 0; t < 16; ++t) {
        const double p = (double)(float)hA[i * 16 + t] * (double)(float)hB[t * 32 + j];
        ex += p; sabs += fabs(p);
      }
      const float d = hD[i * 32 + j];
      ++n; eq_exact += (d == (float)ex);
      const double rel = fabs((double)d - ex) / (sabs + 1e-300);
      if (rel > worst) worst = rel;
    }
  }
  printf("outputs %ld  == exact sum rounded once %ld (%.2f%%)\n", n, eq_exact, 100.0 * eq_exact / n);
  printf("max |D - exact| / sum|terms| = %.3e = 2^%.2f  (2^-22 = %.3e)\n", worst, log2(worst), ldexp(1.0, -22));
  return 0;
}
