// accumulate.h -- device-side chunk accumulation shared by the k-means M-step
// (kmeans.hip) and segment_reduce (segreduce.hip).  See kmeans.hip for the
// ordering argument (canonical order C2).
#pragma once
#include "common.h"

namespace hsgk {

template <int NW>
__device__ inline int owner_wave(int label) {
  return (label ^ (label >> 2) ^ (label >> 4) ^ (label >> 6)) & (NW - 1);
}

// Accumulates the rows of one chunk (n <= HSGK_CHUNK) whose label lies in [lo, lo+cnt_lab) into
// the zeroed LDS table sums[cnt_lab][DS]; present[l] (optional, LDS, zeroed by the
// caller) is set for every label that owns at least one row.  Shared by the k-means M-step
// (int32 working labels, window = cluster block) and segment_reduce (int64
// labels, window = the chunk's own label range).
//
// TAILV = 1 (VEC = 4, rows of 256 + 65..255 columns, e.g. d = 386): the columns after the
// first 256 are prefetched too -- one more x4 per lane for the whole quads, one float per
// lane for the last d mod 4 columns -- instead of being loaded where they are folded (a
// load consumed right after its issue drains the in-order prefetch queue).
template <int VEC, int UNROLL, typename LabT, int NW = 4, int TAILV = 0>
__device__ inline void chunk_accumulate(const float *__restrict__ xr, int d, int DS,
                                        const LabT *__restrict__ lab, int n, int64_t lo,
                                        int cnt_lab, float *sums, uint32_t *rlist,
                                        int *wcount, unsigned char *present = nullptr) {
  typedef float gvec_t __attribute__((ext_vector_type(VEC), aligned(4)));       // global: dword aligned
  typedef float lvec_t __attribute__((ext_vector_type(VEC), aligned(4 * VEC))); // LDS: natural
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

  // ---- the chunk's labels, window-relative, 64 rows per register (n <= HSGK_CHUNK):
  //      all loads are issued back to back (a load-use loop here costs one exposed
  //      memory latency per 64 rows, twice)
  constexpr int NL = HSGK_CHUNK / 64;
  int lw[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int r = 64 * i + lane;
    const int64_t l = (int64_t)lab[min(r, n - 1)] - lo;
    lw[i] = (r < n && l >= 0 && l < cnt_lab) ? (int)l : -1;
  }
  // ---- pass 1: every wave counts the rows it owns; pass 2: ordered row list
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const bool mine = lw[i] >= 0 && owner_wave<NW>(lw[i]) == w;
    cnt += __popcll(__ballot(mine));
  }
  if (lane == 0) wcount[w] = cnt;
  __syncthreads();
  int lbeg = 0;
  for (int i = 0; i < w; ++i) lbeg += wcount[i];
  {
    int pos = lbeg;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const bool mine = lw[i] >= 0 && owner_wave<NW>(lw[i]) == w;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        rlist[pos + __popcll(m & ((1ull << lane) - 1ull))] = ((uint32_t)(64 * i + lane) << 10) | (uint32_t)lw[i];
        if (present) present[lw[i]] = 1;          // (zeroed by the caller; benign same-value race)
      }
      pos += __popcll(m);
    }
  }
  // (a wave reads back only its own list entries: in-order LDS, no barrier)

  constexpr int PW = 64 * VEC;
  const int npass = d / PW;
  const int tail0 = npass * PW;
  const int tail = d - tail0;          // < 64*VEC columns, one float per lane per step
  const uint32_t *mylist = rlist + lbeg;

  // Batches of UNROLL rows; the loads of batch i+1 are issued before batch i is
  // folded into LDS (only the first column pass is double buffered: npass == 1
  // for d < 2*64*VEC, the shapes this kernel is tuned for).
  gvec_t va[UNROLL], vb[UNROLL];
  gvec_t wa[TAILV ? UNROLL : 1], wb[TAILV ? UNROLL : 1];      // second prefetched vector (TAILV)
  float ta[UNROLL], tb[UNROLL];
  const int nv2 = TAILV ? tail / VEC : 0;                      // lanes with a whole tail quad
  const int tcol = tail0 + nv2 * VEC;                          // first column of the scalar tail
  const bool on = lane < d - tcol;
  auto issue = [&](int b0, gvec_t (&v)[UNROLL], gvec_t (&v2)[TAILV ? UNROLL : 1], float (&t)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int idx = min(b0 + u, cnt - 1);
      const int r = (int)(mylist[idx] >> 10);
      // all loads unconditional (clamped columns; unused values are ignored by fold)
      v[u] = *reinterpret_cast<const gvec_t *>(xr + (int64_t)r * d + min(lane * VEC, d - VEC));
      if constexpr (TAILV != 0)
        v2[u] = *reinterpret_cast<const gvec_t *>(xr + (int64_t)r * d + min(tail0 + lane * VEC, d - VEC));
      t[u] = xr[(int64_t)r * d + min(tcol + lane, d - 1)];
    }
  };
  // The prefetched columns of the CURRENT label stay in registers while consecutive rows of the
  // wave's list carry the same label (image-major rows: runs of tens to hundreds) and go back to
  // the LDS table when it changes: a read-add-write through LDS per row is a ~150-cycle dependent
  // chain, which bounded chunks with few distinct labels (one wave then owns >1000 rows).  Same
  // additions in the same order.
  int curl = -1;
  lvec_t racc, racc2;
  float rt = 0.0f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) { racc[i] = 0.0f; racc2[i] = 0.0f; }
  auto switch_to = [&](int l) {
    if (curl >= 0) {
      if (npass > 0) *reinterpret_cast<lvec_t *>(sums + curl * DS + lane * VEC) = racc;
      if constexpr (TAILV != 0)
        if (lane < nv2) *reinterpret_cast<lvec_t *>(sums + curl * DS + tail0 + lane * VEC) = racc2;
      if (on) sums[curl * DS + tcol + lane] = rt;
    }
    curl = l;
    if (l >= 0) {
      if (npass > 0) racc = *reinterpret_cast<const lvec_t *>(sums + l * DS + lane * VEC);
      if constexpr (TAILV != 0)
        if (lane < nv2) racc2 = *reinterpret_cast<const lvec_t *>(sums + l * DS + tail0 + lane * VEC);
      if (on) rt = sums[l * DS + tcol + lane];
    }
  };
  auto fold = [&](int b0, const gvec_t (&v)[UNROLL], const gvec_t (&v2)[TAILV ? UNROLL : 1],
                  const float (&t)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (b0 + u < cnt) {
        const uint32_t e = mylist[b0 + u];
        const int r = (int)(e >> 10), l = (int)(e & 1023u);
        if (l != curl) switch_to(l);
        if (npass > 0) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) racc[i] = racc[i] + v[u][i];
        }
        for (int p = 1; p < npass; ++p) {                     // further full passes (wide rows)
          const gvec_t vv = *reinterpret_cast<const gvec_t *>(xr + (int64_t)r * d + p * PW + lane * VEC);
          lvec_t *dst = reinterpret_cast<lvec_t *>(sums + l * DS + p * PW + lane * VEC);
          lvec_t acc = *dst;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = acc[i] + vv[i];
          *dst = acc;
        }
        if constexpr (TAILV != 0) {
          if (lane < nv2) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) racc2[i] = racc2[i] + v2[u][i];
          }
        }
        if (on) rt = rt + t[u];
        if constexpr (TAILV == 0)
          for (int t0 = 64; t0 < tail; t0 += 64)              // wider tails, loaded in place (slow path)
            if (t0 + lane < tail) {
              float *dst = sums + l * DS + tail0 + t0 + lane;
              *dst = *dst + xr[(int64_t)r * d + tail0 + t0 + lane];
            }
      }
    }
  };
  if (cnt > 0) {
    // issue() clamps its row indices, so it runs UNCONDITIONALLY: with a conditional
    // issue in the loop the compiler's s_waitcnt insertion (pessimistic at control-flow
    // joins) waits for vmcnt(0) before every fold and the double buffering is lost.
    issue(0, va, wa, ta);
    for (int b0 = 0; b0 < cnt; b0 += 2 * UNROLL) {
      issue(b0 + UNROLL, vb, wb, tb);
      __builtin_amdgcn_sched_barrier(0);
      fold(b0, va, wa, ta);
      __builtin_amdgcn_sched_barrier(0);
      issue(b0 + 2 * UNROLL, va, wa, ta);
      __builtin_amdgcn_sched_barrier(0);
      fold(b0 + UNROLL, vb, wb, tb);
      __builtin_amdgcn_sched_barrier(0);
    }
    switch_to(-1);                                   // the last label's registers -> table
  }
}


}  // namespace hsgk
