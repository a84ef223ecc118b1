// From the bare sparse gather (sparse_gather.hip: 5.3 TB/s at 28 % of the rows) towards the shape of the exact-sum
// update kernel, one feature at a time -- which one halves the rate?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gather_steps.hip -o tools/probes/gather_steps
// F bits: 1 = 16 waves x 4 rows in flight per buffer (else 8 x 8); 2 = row numbers through a wave-private LDS list
// (written per 256-row strip, read back per row); 4 = the conversion work (4 x fp64 fma + 64-bit accumulate, two
// sides); 8 = LDS atomics (8 x ds_add_u64 per row into a 132 KB table); 16 = strips dealt round robin to the waves
// of a workgroup (else one contiguous range per wave); 32 = the two tail columns: one per-lane load and two atomics per
// batch of rows; 64 = the selection from two BYTE label arrays in memory (flat loads through a tagged pointer, as
// get_label does) instead of a hash; 128 = run-length carries (accumulate per side, flush with zeroing on a label change);
// 256 = the strip's row count / first row from per-chunk tables in memory
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ inline unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ inline long long to_fixed(float x) {
  const double magic = 6755399441055744.0;
  const double t = __builtin_fma((double)x, 1099511627776.0, magic);
  return __builtin_bit_cast(long long, t) - __builtin_bit_cast(long long, magic);
}
template <int F>
__global__ __launch_bounds__((F & 1) ? 1024 : 512) void gather(const float *x, long rows, int G, float *sink, unsigned long long *count,
    const uint8_t *lab_prev, const uint8_t *lab_cur, const long *chunk_row0, const int *chunk_rows) {
  constexpr int NW = (F & 1) ? 16 : 8, U = (F & 1) ? 4 : 8;
  extern __shared__ unsigned long long lds[];
  unsigned long long *tab = lds;                                  // [64][258]
  uint32_t *lists = reinterpret_cast<uint32_t *>(lds + 64 * 258 + 2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t *list = lists + w * 256;
  if (F & 8) { for (int i = threadIdx.x; i < 64 * 258; i += NW * 64) tab[i] = 0; __syncthreads(); }
  const long per_wg = rows / gridDim.x;
  const long nstrips = per_wg / 256;
  float s = 0.f;
  long long acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  long long cq[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  int clab[2] = {-1, -1};
  const int tu = lane >> 1, tc = lane & 1;
  unsigned long long n = 0;
  for (long st = (F & 16) ? w : (long)w * (nstrips / NW); st < ((F & 16) ? nstrips : (long)(w + 1) * (nstrips / NW)); st += (F & 16) ? NW : 1) {
    long row0 = (long)blockIdx.x * per_wg + st * 256;
    if (F & 256) {
      const long ch = row0 / 2048;
      const int nn = chunk_rows[ch];
      row0 = chunk_row0[ch] + (row0 & 2047);
      if (nn <= 0) continue;
    }
    // the strip's selected rows: row r is taken with probability ~ 1 / (G / 2 + 1)
    int total = 0;
    uint32_t mine[4]; bool sel[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 64 * i + lane;
      int pl = 0, cl = 0;
      if (F & 64) {
        const uintptr_t pa = reinterpret_cast<uintptr_t>(lab_prev) | 1, ca = reinterpret_cast<uintptr_t>(lab_cur) | 1;
        pl = (pa & 1) ? (int)reinterpret_cast<const uint8_t *>(pa ^ 1)[row0 + r] : reinterpret_cast<const int *>(pa)[row0 + r];
        cl = (ca & 1) ? (int)reinterpret_cast<const uint8_t *>(ca ^ 1)[row0 + r] : reinterpret_cast<const int *>(ca)[row0 + r];
        sel[i] = pl != cl;
      } else
      sel[i] = hsh((unsigned)(row0 + r)) % (unsigned)(G + 2) < 2u;
      const unsigned long long m = __ballot(sel[i]);
      mine[i] = (uint32_t)total + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if ((F & 2) && sel[i]) list[mine[i]] = ((uint32_t)r << 22) | ((uint32_t)((F & 64) ? cl & 63 : hsh(r + 7u) & 63u) << 11) | (uint32_t)((F & 64) ? pl & 63 : hsh(r + 9u) & 63u);
      total += __popcll(m);
    }
    if (total == 0) continue;
    n += (lane == 0) ? total : 0;
    const float *xr = x + row0 * 258;
    auto rowof = [&](int i) -> uint32_t {
      i = min(i, total - 1);
      if (F & 2) return (uint32_t)__builtin_amdgcn_readfirstlane((int)list[i]);
      // without the list: the i-th selected row recomputed from the ballots (scalar)
      return ((uint32_t)((i * (G / 2 + 1)) & 255) << 22) | ((hsh(i + 7u) & 63u) << 11) | (hsh(i + 9u) & 63u);
    };
    f4 va[U], vb[U];
    float ta = 0.f, tb = 0.f;
    auto rowof_v = [&](int i) -> uint32_t { i = min(i, total - 1); return (F & 2) ? list[i] : (uint32_t)((i * (G / 2 + 1)) & 255) << 22; };
    auto issue = [&](int i0, f4 (&v)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *(const f4 *)(xr + (long)(rowof(i0 + u) >> 22) * 258 + 4 * lane);
    };
    auto issue_t = [&](int i0, float &t) { if (F & 32) t = xr[(long)(rowof_v(i0 + min(tu, U - 1)) >> 22) * 258 + 256 + tc]; };
    auto fold_t = [&](int i0, const float &t) {
      if ((F & 32) && tu < U && i0 + tu < total) {
        const uint32_t e = rowof_v(i0 + tu);
        const long long qt = to_fixed(t);
        if (F & 8) { atomicAdd(tab + (size_t)((e >> 11) & 63u) * 258 + 256 + tc, (unsigned long long)qt); atomicAdd(tab + (size_t)(e & 63u) * 258 + 256 + tc, (unsigned long long)(-qt)); }
        else acc[0][0] += qt;
      }
    };
    auto fold = [&](int i0, const f4 (&v)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + u >= total) continue;
        if (F & 4) {
          const uint32_t e = rowof(i0 + u);
          long long q[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) q[j] = to_fixed(v[u][j]);
          if ((F & 8) && (F & 128)) {
            const int labs[2] = {(int)((e >> 11) & 63u), (int)(e & 63u)};
#pragma unroll
            for (int side = 0; side < 2; ++side) {
              if (labs[side] != clab[side]) {
                if (clab[side] >= 0) {
                  unsigned long long *rp = tab + (size_t)clab[side] * 258;
#pragma unroll
                  for (int j = 0; j < 4; ++j) atomicAdd(rp + j * 64 + lane, (unsigned long long)cq[side][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) cq[side][j] = 0;
                clab[side] = labs[side];
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) cq[side][j] += side ? -q[j] : q[j];
            }
          } else if (F & 8) {
            unsigned long long *a = tab + (size_t)((e >> 11) & 63u) * 258, *b = tab + (size_t)(e & 63u) * 258;
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicAdd(a + j * 64 + lane, (unsigned long long)q[j]); atomicAdd(b + j * 64 + lane, (unsigned long long)(-q[j])); }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][j] += q[j]; acc[1][j] -= q[j] >> 1; }
          }
        } else {
          s += v[u][0] + v[u][1] + v[u][2] + v[u][3];
        }
      }
    };
    issue(0, va); issue_t(0, ta);
    for (int i0 = 0; i0 < total; i0 += 2 * U) {
      issue(i0 + U, vb); issue_t(i0 + U, tb);
      __builtin_amdgcn_sched_barrier(0);
      fold(i0, va); fold_t(i0, ta);
      __builtin_amdgcn_sched_barrier(0);
      issue(i0 + 2 * U, va); issue_t(i0 + 2 * U, ta);
      __builtin_amdgcn_sched_barrier(0);
      fold(i0 + U, vb); fold_t(i0 + U, tb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  s += (float)(cq[0][0] + cq[1][1]) + (float)(acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3]);
  if (F & 8) { __syncthreads(); s += (float)tab[threadIdx.x]; }
  if (s == 123.456f) *sink = s;
  if (lane == 0) atomicAdd(count, n);
}
template <int F>
void run(const float *x, long rows, float *sink, unsigned long long *cnt, const char *what, const uint8_t *lp, const uint8_t *lc5, const uint8_t *lc76,
         const long *cr0, const int *crn) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const size_t lds = (size_t)(64 * 258 + 2) * 8 + 16 * 256 * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gather<F>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int G : {5, 76}) {
    float best = 1e9f; unsigned long long h = 0;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipMemset(cnt, 0, 8);
      (void)hipEventRecord(a);
      hipLaunchKernelGGL(gather<F>, dim3(256), dim3((F & 1) ? 1024 : 512), lds, 0, x, rows, G, sink, cnt, lp, G == 5 ? lc5 : lc76, cr0, crn);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      (void)hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost);
    }
    printf("%-64s G %2d rows %8llu (%4.1f %%)  %.3f ms  %5.0f GB/s\n", what, G, h, 100.0 * h / rows, best, h * 1032.0 / best / 1e6);
  }
}
int main() {
  const long rows = 48L * 448 * 448;
  float *x, *sink; unsigned long long *cnt;
  (void)hipMalloc(&x, (size_t)rows * 1032); (void)hipMalloc(&sink, 4); (void)hipMalloc(&cnt, 8);
  (void)hipMemset(x, 0x3c, (size_t)rows * 1032);
  // label arrays: prev random in 0..63; cur differs from prev on ~28.6 % (G = 5) / ~2.6 % (G = 76) of the rows
  uint8_t *hp = (uint8_t *)malloc(rows), *h5 = (uint8_t *)malloc(rows), *h76 = (uint8_t *)malloc(rows);
  unsigned st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (long i = 0; i < rows; ++i) {
    hp[i] = rnd() & 63;
    h5[i] = (rnd() % 7u) < 2u ? (uint8_t)((hp[i] + 1 + rnd() % 63u) & 63) : hp[i];
    h76[i] = (rnd() % 78u) < 2u ? (uint8_t)((hp[i] + 1 + rnd() % 63u) & 63) : hp[i];
  }
  uint8_t *lp, *lc5, *lc76; long *cr0; int *crn;
  (void)hipMalloc(&lp, rows); (void)hipMalloc(&lc5, rows); (void)hipMalloc(&lc76, rows);
  (void)hipMemcpy(lp, hp, rows, hipMemcpyHostToDevice); (void)hipMemcpy(lc5, h5, rows, hipMemcpyHostToDevice); (void)hipMemcpy(lc76, h76, rows, hipMemcpyHostToDevice);
  const long nch = rows / 2048;
  long *hr0 = (long *)malloc(nch * 8); int *hrn = (int *)malloc(nch * 4);
  for (long c = 0; c < nch; ++c) { hr0[c] = c * 2048; hrn[c] = 2048; }
  (void)hipMalloc(&cr0, nch * 8); (void)hipMalloc(&crn, nch * 4);
  (void)hipMemcpy(cr0, hr0, nch * 8, hipMemcpyHostToDevice); (void)hipMemcpy(crn, hrn, nch * 4, hipMemcpyHostToDevice);
#define RUN(F, what) run<F>(x, rows, sink, cnt, what, lp, lc5, lc76, cr0, crn)
  RUN(0, "8 waves x 8 rows, contiguous range per wave");
  RUN(16, "  + strips dealt round robin to the waves");
  RUN(17, "  + 16 waves x 4 rows");
  RUN(19, "  + row numbers through the LDS list");
  RUN(31, "  + fp64 conversion + LDS atomics (8 ds_add_u64 per row)");
  RUN(63, "  + tail columns (per-lane load, 2 atomics per batch)");
  RUN(127, "  + labels from two byte arrays (flat loads), pl != cl");
  RUN(255, "  + run-length carries (accumulate, flush + zero on change)");
  RUN(511, "  + chunk tables per strip");
  RUN(383, "  all but the carries (511 - 128)");
  return 0;
}
