// What ds_read_b64_tr_b16 returns: LDS plane P[32][72] halfs with P[r][c] = r * 64 + c; lane l = 16 g + p supplies
// the address of P[4 g + p / 4][4 (p % 4)]; expectation: lane gets P[4 g + k][p], k = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 llvm_h4;
__global__ void probe(float *out) {
  __shared__ __attribute__((aligned(16))) _Float16 P[32 * 72];
  for (int i = threadIdx.x; i < 32 * 72; i += 64) P[i] = (_Float16)((i / 72) * 64 + (i % 72));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, p = l & 15;
  const _Float16 *src = &P[(4 * g + p / 4) * 72 + 4 * (p % 4)];
  auto lp = (__attribute__((address_space(3))) llvm_h4 *)(src);
  llvm_h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(lp);
  for (int k = 0; k < 4; ++k) out[l * 4 + k] = (float)v[k];
}
int main() {
  float *d; hipMalloc(&d, 256 * 4);
  probe<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, p = l & 15;
    printf("lane %2d:", l);
    for (int k = 0; k < 4; ++k) {
      const int r = (int)h[l * 4 + k] / 64, c = (int)h[l * 4 + k] % 64;
      printf(" (%d,%d)", r, c);
      if (r != 4 * g + k || c != p) ++bad;
    }
    printf("\n");
  }
  printf("mismatches vs expectation: %d\n", bad);
  return 0;
}
