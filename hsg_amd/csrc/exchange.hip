// exchange.hip -- the cross-GPU prototype exchange of hsg/models/utils.py:127-217
// (gather_clustering_and_update_prototypes), in-stream: no sort over the N pixel keys, no host
// read, one fused pass over both row sets, only tables on the wire.
//
//   keys    (batch, cluster, semantic, instance) per pixel -> packed 63-bit key (radices = the
//           maxima, found on the device) -> per-1024-row LDS hash set -> the distinct keys of a
//           workgroup go into a global open-addressing table (one CAS per distinct key and
//           workgroup, not per pixel) -> compaction -> bitonic sort of the P_local distinct keys
//           -> sorted tuple list with its count in the header (the all_gather payload)
//   merge   the ranks' sorted lists -> global dense ids: rank of a tuple among the distinct
//           tuples of all ranks in lexicographic order (the reference's two nested sorted
//           `unique`s, utils.py:181-193) by binary searches, no global sort
//   sums    per chunk of 2048 consecutive rows (canonical order C2) the rows are stably sorted by
//           id in LDS; every wave owns a COLUMN slice of [embeddings | embeddings_with_loc] and
//           walks the same sorted list, a run of one id adds up in registers and leaves as one
//           partial row: no LDS table, no label hashing, all waves evenly loaded, both row sets
//           read once; per segment the chunk partials are summed in chunk order
//   finish  all_reduce(sum) of the [rows, C + D] table over RCCL, then both halves normalised
//           (C1 norm chain, general/common.py:101-120)
//
// The collectives are RCCL calls on the caller's stream (ncclComm_t handed in by the caller);
// librccl is bound at run time (dlopen: the instance torch already loaded, else the system one),
// so libhsgk.so itself carries no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace hsgk {

constexpr int kXHdr = 8;                          // int64 words in front of a rank's tuple list
constexpr int kXKeyRows = 1024;                   // rows per workgroup of the key pass
constexpr int kXLdsSlots = 2048;                  // LDS hash slots of the key pass (2 x rows)
constexpr int kSortTile = 2048;
// (unused ckeys entries are filled by hipMemset(0x7F): 0x7F7F... sorts behind every real key, which is < 2^62)

struct XCtrl {                                    // device control block, zeroed per call
  unsigned long long mx[4];                       // maxima of cluster, batch, semantic, instance
  unsigned long long gmx[4];                      // the same over the gathered tuples
  int32_t n_keys;                                 // distinct local keys (compaction counter)
  int32_t error;                                  // bit 0 negative component, 1 capacity, 2 key overflow, 3 table rows
  int32_t pool_used;                              // partial rows handed out to chunks
  int32_t pad;
};

__device__ inline unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// ---------------------------------------------------------------------------------------------
// keys
__global__ __launch_bounds__(256) void xk_ranges_kernel(const int64_t *__restrict__ c, const int64_t *__restrict__ b,
                                                        const int64_t *__restrict__ s, const int64_t *__restrict__ t,
                                                        int64_t n, XCtrl *ctrl) {
  int64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  bool neg = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t v0 = c[i], v1 = b[i], v2 = s[i], v3 = t[i];
    neg |= (v0 | v1 | v2 | v3) < 0;
    m0 = v0 > m0 ? v0 : m0; m1 = v1 > m1 ? v1 : m1; m2 = v2 > m2 ? v2 : m2; m3 = v3 > m3 ? v3 : m3;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t o0 = __shfl_xor(m0, off), o1 = __shfl_xor(m1, off), o2 = __shfl_xor(m2, off), o3 = __shfl_xor(m3, off);
    m0 = o0 > m0 ? o0 : m0; m1 = o1 > m1 ? o1 : m1; m2 = o2 > m2 ? o2 : m2; m3 = o3 > m3 ? o3 : m3;
  }
  // one set of atomics per WORKGROUP (one per wave on the same four words serialised in L2: 0.40 ms
  // for 9.6 M rows against 0.06 ms of reads)
  __shared__ int64_t red[4][4];
  __shared__ int anyneg;
  const int w = threadIdx.x >> 6;
  if (threadIdx.x == 0) anyneg = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[0][w] = m0; red[1][w] = m1; red[2][w] = m2; red[3][w] = m3; }
  if (__ballot(neg) && (threadIdx.x & 63) == 0) anyneg = 1;
  __syncthreads();
  if (threadIdx.x < 4) {
    int64_t m = red[threadIdx.x][0];
    for (int i = 1; i < 4; ++i) m = red[threadIdx.x][i] > m ? red[threadIdx.x][i] : m;
    if (m > 0) atomicMax(&ctrl->mx[threadIdx.x], (unsigned long long)m);
  }
  if (threadIdx.x == 0 && anyneg) atomicOr(&ctrl->error, 1);
}

struct Radix { long long rc, rl; bool ok; };
__device__ inline Radix radix_of(const unsigned long long *mx) {
  Radix r;
  r.rc = (long long)mx[0] + 1;
  const long long a = (long long)mx[2], b = (long long)mx[3];
  r.rl = (a > b ? a : b) + 1;
  // ((batch * rc + cluster) * rl + sem) * rl + inst must stay below 2^62
  r.ok = ((double)mx[1] + 1.0) * (double)r.rc * (double)r.rl * (double)r.rl < 4.0e18;
  return r;
}
__device__ inline long long pack_key(long long b, long long c, long long s, long long t, long long rc, long long rl) {
  return ((b * rc + c) * rl + s) * rl + t;
}

// One workgroup per 1024 rows: LDS hash set of the rows' keys, its distinct keys into the global
// table (linear probing, device-scope CAS), every row's global slot out.
__global__ __launch_bounds__(256) void xk_insert_kernel(const int64_t *__restrict__ c, const int64_t *__restrict__ b,
                                                        const int64_t *__restrict__ s, const int64_t *__restrict__ t,
                                                        int64_t n, long long *__restrict__ table, int hmask,
                                                        int32_t *__restrict__ hslot, XCtrl *ctrl) {
  __shared__ long long lkey[kXLdsSlots];
  __shared__ int32_t lslot[kXLdsSlots];
  const Radix rx = radix_of(ctrl->mx);
  if (!rx.ok || (ctrl->error & 1)) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && !rx.ok) atomicOr(&ctrl->error, 4);
    return;
  }
  const int tid = threadIdx.x;
  for (int i = tid; i < kXLdsSlots; i += 256) lkey[i] = -1;
  __syncthreads();
  const int64_t row0 = (int64_t)blockIdx.x * kXKeyRows;
  long long key[kXKeyRows / 256];
  int pos[kXKeyRows / 256];
#pragma unroll
  for (int u = 0; u < kXKeyRows / 256; ++u) {
    const int64_t i = row0 + u * 256 + tid;
    key[u] = -1;
    pos[u] = 0;
    if (i < n) {
      key[u] = pack_key(b[i], c[i], s[i], t[i], rx.rc, rx.rl);
      int h = (int)(mix64((unsigned long long)key[u]) & (kXLdsSlots - 1));
      for (;;) {                                   // <= 1024 keys in 2048 slots: always terminates
        const long long cur = lkey[h];
        if (cur == key[u]) break;
        if (cur == -1) {
          const long long old = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(&lkey[h]),
                                                     (unsigned long long)-1LL, (unsigned long long)key[u]);
          if (old == -1 || old == key[u]) break;
        }
        h = (h + 1) & (kXLdsSlots - 1);
      }
      pos[u] = h;
    }
  }
  __syncthreads();
  for (int i = tid; i < kXLdsSlots; i += 256) {
    const long long k = lkey[i];
    if (k == -1) continue;
    int h = (int)((mix64((unsigned long long)k) >> 20) & (unsigned)hmask);
    int probes = 0;
    for (;; ++probes) {
      if (probes > hmask) { atomicOr(&ctrl->error, 2); h = 0; break; }      // table full
      const long long old = (long long)atomicCAS(reinterpret_cast<unsigned long long *>(&table[h]),
                                                 (unsigned long long)-1LL, (unsigned long long)k);
      if (old == -1 || old == k) break;
      h = (h + 1) & hmask;
    }
    lslot[i] = h;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kXKeyRows / 256; ++u) {
    const int64_t i = row0 + u * 256 + tid;
    if (i < n) hslot[i] = lslot[pos[u]];
  }
}

// occupied table slots -> (key, slot) pairs in any order
__global__ __launch_bounds__(256) void xk_compact_kernel(const long long *__restrict__ table, int hsize, int cap,
                                                         long long *__restrict__ ckeys, int32_t *__restrict__ cvals,
                                                         XCtrl *ctrl) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const long long k = i < hsize ? table[i] : -1;
  const bool on = k != -1;
  const unsigned long long m = __ballot(on);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(&ctrl->n_keys, __popcll(m));
  base = __shfl(base, 0);
  if (on) {
    const int p = base + __popcll(m & ((1ull << lane) - 1ull));
    if (p < cap) { ckeys[p] = k; cvals[p] = i; }
  }
}

// ---- bitonic sort of (key, value) pairs, n = power of two >= kSortTile ------------------------
// kglobal == 0: a tile is sorted completely (all k <= kSortTile); otherwise only the strides
// j < kSortTile of merge step kglobal
__global__ __launch_bounds__(256) void xk_sort_tile_kernel(long long *__restrict__ keys, int32_t *__restrict__ vals,
                                                           int kglobal, const XCtrl *ctrl) {
  __shared__ long long sk[kSortTile];
  __shared__ int32_t sv[kSortTile];
  const int tid = threadIdx.x;
  const int64_t g0 = (int64_t)blockIdx.x * kSortTile;
  // the compaction fills positions [0, n_keys); everything behind is padding that sorts last.  A tile of
  // padding only needs no work, and with all keys inside the first tile its sorted order is final
  // (merging an ascending tile with padding leaves it in place)
  const int64_t nk = ctrl->n_keys;
  if (kglobal ? nk <= kSortTile : g0 >= nk) return;
  // with all keys inside the first tile, its sort only has to order the first m = 2^ceil(log2(keys)) entries: the
  // padding behind them sorts last and already sits there (a training step's exchange sorts ~100 keys: 28 network
  // steps instead of 66)
  int m = kSortTile;
  if (!kglobal && nk <= kSortTile) {           // (all keys in tile 0, which sorts ascending; later tiles alternate)
    m = 2;
    while (m < (int)nk) m <<= 1;
  }
  for (int i = tid; i < kSortTile; i += 256) { sk[i] = keys[g0 + i]; sv[i] = vals[g0 + i]; }
  __syncthreads();
  for (int k = kglobal ? kSortTile : 2; k <= m; k <<= 1) {
    for (int j = (kglobal ? kSortTile : k) >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < m / 2; p += 256) {
        const int i = 2 * j * (p / j) + (p % j);
        const int64_t gi = g0 + i;
        const bool asc = ((gi & (kglobal ? (int64_t)kglobal : (int64_t)k)) == 0);
        const long long a = sk[i], b = sk[i + j];
        if ((a > b) == asc) {
          sk[i] = b; sk[i + j] = a;
          const int32_t va = sv[i]; sv[i] = sv[i + j]; sv[i + j] = va;
        }
      }
      __syncthreads();
    }
    if (kglobal) break;
  }
  for (int i = tid; i < kSortTile; i += 256) { keys[g0 + i] = sk[i]; vals[g0 + i] = sv[i]; }
}

__global__ __launch_bounds__(256) void xk_sort_global_kernel(long long *__restrict__ keys, int32_t *__restrict__ vals,
                                                             int64_t half, int k, int j, const XCtrl *ctrl) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= half || ctrl->n_keys <= kSortTile) return;
  const int64_t i = 2 * (int64_t)j * (p / j) + (p % j);
  const bool asc = (i & (int64_t)k) == 0;
  const long long a = keys[i], b = keys[i + j];
  if ((a > b) == asc) {
    keys[i] = b; keys[i + j] = a;
    const int32_t va = vals[i]; vals[i] = vals[i + j]; vals[i + j] = va;
  }
}

static int launch_sort_pairs(long long *keys, int32_t *vals, int64_t n, const XCtrl *ctrl, hipStream_t s) {
  const unsigned tiles = (unsigned)(n / kSortTile);
  hipLaunchKernelGGL(xk_sort_tile_kernel, dim3(tiles), dim3(256), 0, s, keys, vals, 0, ctrl);
  HSGK_LAUNCH_CHECK();
  for (int64_t k = 2 * kSortTile; k <= n; k <<= 1) {
    for (int64_t j = k >> 1; j >= kSortTile; j >>= 1) {
      hipLaunchKernelGGL(xk_sort_global_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, keys, vals,
                         n / 2, (int)k, (int)j, ctrl);
      HSGK_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(xk_sort_tile_kernel, dim3(tiles), dim3(256), 0, s, keys, vals, (int)k, ctrl);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

// sorted keys -> rank of every table slot, the rank's tuple list (the gather payload) with its header
__global__ __launch_bounds__(256) void xk_rank_kernel(const long long *__restrict__ ckeys, const int32_t *__restrict__ cvals,
                                                      int cap, int32_t *__restrict__ rank_of_slot,
                                                      int64_t *__restrict__ send, XCtrl *ctrl) {
  const int nk = ctrl->n_keys;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    if (nk > cap) atomicOr(&ctrl->error, 2);
    send[0] = nk;
    send[1] = ctrl->error | (nk > cap ? 2 : 0);
    for (int w = 2; w < kXHdr; ++w) send[w] = 0;
  }
  if (i >= (nk < cap ? nk : cap)) return;
  const Radix rx = radix_of(ctrl->mx);
  long long k = ckeys[i];
  rank_of_slot[cvals[i]] = i;
  int64_t *tp = send + kXHdr + 4 * (int64_t)i;
  const long long t3 = k % rx.rl; k /= rx.rl;
  const long long t2 = k % rx.rl; k /= rx.rl;
  tp[0] = k / rx.rc; tp[1] = k % rx.rc; tp[2] = t2; tp[3] = t3;
}

// ---------------------------------------------------------------------------------------------
// merge of `world` sorted tuple lists (blocks of kXHdr + 4 * cap words)
__global__ __launch_bounds__(256) void xk_merge_radix_kernel(const int64_t *__restrict__ recv, int world, int cap,
                                                             XCtrl *ctrl) {
  const int r = blockIdx.y;
  const int64_t *blk = recv + (int64_t)r * (kXHdr + 4 * (int64_t)cap);
  const int64_t cnt = blk[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int e = (int)blk[1];
    if (cnt > cap) e |= 2;
    if (e) atomicOr(&ctrl->error, e);
  }
  const int64_t m = cnt < cap ? cnt : cap;
  int64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const int64_t *tp = blk + kXHdr + 4 * i;
    m1 = tp[0] > m1 ? tp[0] : m1; m0 = tp[1] > m0 ? tp[1] : m0;
    m2 = tp[2] > m2 ? tp[2] : m2; m3 = tp[3] > m3 ? tp[3] : m3;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const int64_t o0 = __shfl_xor(m0, off), o1 = __shfl_xor(m1, off), o2 = __shfl_xor(m2, off), o3 = __shfl_xor(m3, off);
    m0 = o0 > m0 ? o0 : m0; m1 = o1 > m1 ? o1 : m1; m2 = o2 > m2 ? o2 : m2; m3 = o3 > m3 ? o3 : m3;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&ctrl->gmx[0], (unsigned long long)m0);
    atomicMax(&ctrl->gmx[1], (unsigned long long)m1);
    atomicMax(&ctrl->gmx[2], (unsigned long long)m2);
    atomicMax(&ctrl->gmx[3], (unsigned long long)m3);
  }
}

__global__ __launch_bounds__(256) void xk_merge_pack_kernel(const int64_t *__restrict__ recv, int cap,
                                                            long long *__restrict__ gkey, XCtrl *ctrl) {
  const int r = blockIdx.y;
  const int64_t *blk = recv + (int64_t)r * (kXHdr + 4 * (int64_t)cap);
  const int64_t m = blk[0] < cap ? blk[0] : cap;
  const Radix rx = radix_of(ctrl->gmx);
  if (!rx.ok) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&ctrl->error, 4);
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const int64_t *tp = blk + kXHdr + 4 * i;
    gkey[(int64_t)r * cap + i] = pack_key(tp[0], tp[1], tp[2], tp[3], rx.rc, rx.rl);
  }
}

__device__ inline int lower_bound_ll(const long long *a, int n, long long key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// prim[r][i] = 1 when no list r' < r holds the same key (its first occurrence over the ranks)
__global__ __launch_bounds__(256) void xk_merge_primary_kernel(const int64_t *__restrict__ recv, int cap,
                                                               const long long *__restrict__ gkey,
                                                               int32_t *__restrict__ prim) {
  const int r = blockIdx.y;
  const int64_t stride = kXHdr + 4 * (int64_t)cap;
  const int64_t m = recv[r * stride] < cap ? recv[r * stride] : cap;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const long long k = gkey[(int64_t)r * cap + i];
    int p = 1;
    for (int q = 0; q < r && p; ++q) {
      const int mq = (int)(recv[q * stride] < cap ? recv[q * stride] : cap);
      const int lb = lower_bound_ll(gkey + (int64_t)q * cap, mq, k);
      if (lb < mq && gkey[(int64_t)q * cap + lb] == k) p = 0;
    }
    prim[(int64_t)r * (cap + 1) + i] = p;
  }
}

// one workgroup per list: pre[r][i] = number of primary entries before i (i = 0 .. count)
__global__ __launch_bounds__(256) void xk_merge_scan_kernel(const int64_t *__restrict__ recv, int cap,
                                                            int32_t *__restrict__ prim) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int r = blockIdx.x;
  const int64_t stride = kXHdr + 4 * (int64_t)cap;
  const int m = (int)(recv[r * stride] < cap ? recv[r * stride] : cap);
  int32_t *p = prim + (int64_t)r * (cap + 1);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 <= m; i0 += 256) {
    const int i = i0 + tid;
    const int v = i < m ? p[i] : 0;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(inc, off);
      if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = carry;
    for (int q = 0; q < w; ++q) base += wsum[q];
    if (i <= m) p[i] = base + inc - v;
    __syncthreads();
    if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

// global position of every entry = number of distinct keys below it; the first occurrence of a key
// writes the prototype labels (utils.py:193-197).  gslot [world][cap].
__global__ __launch_bounds__(256) void xk_merge_place_kernel(const int64_t *__restrict__ recv, int world, int cap,
                                                             const long long *__restrict__ gkey,
                                                             const int32_t *__restrict__ pre, int32_t *__restrict__ gslot,
                                                             int64_t cap_total, int64_t *__restrict__ psem,
                                                             int64_t *__restrict__ pinst, int64_t *__restrict__ pbatch,
                                                             XCtrl *ctrl, int64_t *__restrict__ meta, int my_rank) {
  const int r = blockIdx.y;
  const int64_t stride = kXHdr + 4 * (int64_t)cap;
  const int64_t m = recv[r * stride] < cap ? recv[r * stride] : cap;
  if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    int64_t total = 0;
    for (int q = 0; q < world; ++q) {
      const int mq = (int)(recv[q * stride] < cap ? recv[q * stride] : cap);
      total += pre[(int64_t)q * (cap + 1) + mq];
    }
    int64_t most = 0;
    for (int q = 0; q < world; ++q) most = recv[q * stride] > most ? recv[q * stride] : most;
    if (total > cap_total) atomicOr(&ctrl->error, 8);
    meta[1] = total;
    meta[2] = ctrl->error | (total > cap_total ? 8 : 0);   // (complete: the kernels after this one set no error bit)
    meta[3] = most;                                  // rows the largest block needs (capacity regrowth)
    meta[4] = -1; meta[5] = -1;
    if (my_rank >= 0) meta[0] = recv[my_rank * stride];
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const long long k = gkey[(int64_t)r * cap + i];
    int64_t pos = 0;
    for (int q = 0; q < world; ++q) {
      const int mq = (int)(recv[q * stride] < cap ? recv[q * stride] : cap);
      const int lb = q == r ? (int)i : lower_bound_ll(gkey + (int64_t)q * cap, mq, k);
      pos += pre[(int64_t)q * (cap + 1) + lb];
    }
    gslot[(int64_t)r * cap + i] = (int32_t)(pos < cap_total ? pos : cap_total - 1);
    const int64_t *tp = recv + r * stride + kXHdr + 4 * i;
    const bool first = pre[(int64_t)r * (cap + 1) + i + 1] != pre[(int64_t)r * (cap + 1) + i];
    if (first && pos < cap_total) { pbatch[pos] = tp[0]; psem[pos] = tp[2]; pinst[pos] = tp[3]; }
  }
}

// world == 1: the local ranks are the global ids
__global__ __launch_bounds__(256) void xk_single_place_kernel(const int64_t *__restrict__ send, int cap,
                                                              int32_t *__restrict__ gslot, int64_t cap_total,
                                                              int64_t *__restrict__ psem, int64_t *__restrict__ pinst,
                                                              int64_t *__restrict__ pbatch, XCtrl *ctrl,
                                                              int64_t *__restrict__ meta) {
  const int64_t cnt = send[0];
  const int64_t m = cnt < cap ? cnt : cap;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int e = (int)send[1];
    if (cnt > cap) e |= 2;
    if (m > cap_total) e |= 8;
    if (e) atomicOr(&ctrl->error, e);
    meta[0] = cnt;
    meta[1] = m;
    meta[2] = e | ctrl->error;          // every error bit is known here (keys + merge): the host may read meta before the sums
    meta[3] = cnt;
  }
  // meta[4], meta[5]: distinct FIRST keys of the sorted tuples and the longest run of one (the per-image prototype
  // tables of the embedding models are [distinct images, most segments of an image]: hierarchy.py reads them from
  // the same host copy as the table length instead of a second stalling read).  Block 0, tables up to 64 K rows.
  if (blockIdx.x == 0) {
    __shared__ int s_carry, s_cnt, s_most, s_wmax[4];
    if (threadIdx.x == 0) { s_carry = -1; s_cnt = 0; s_most = 0; }
    __syncthreads();
    if (m <= 65536) {
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      for (int64_t i0 = 0; i0 < m; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < m;
        const bool first = in && (i == 0 || send[kXHdr + 4 * i] != send[kXHdr + 4 * (i - 1)]);
        int v = first ? (int)i : -1;                       // start of the run entry i belongs to: inclusive max-scan
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(v, off); if (lane >= off) v = o > v ? o : v; }
        if (lane == 63) s_wmax[wv] = v;
        __syncthreads();
        int carry = s_carry;
        for (int q = 0; q < wv; ++q) carry = s_wmax[q] > carry ? s_wmax[q] : carry;
        v = carry > v ? carry : v;
        if (first) atomicAdd(&s_cnt, 1);
        if (in) atomicMax(&s_most, (int)i - v + 1);
        __syncthreads();
        if (threadIdx.x == 255) s_carry = v;
        __syncthreads();
      }
      if (threadIdx.x == 0) { meta[4] = s_cnt; meta[5] = s_most; }
    } else if (threadIdx.x == 0) {
      meta[4] = -1; meta[5] = -1;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    gslot[i] = (int32_t)(i < cap_total ? i : cap_total - 1);
    const int64_t *tp = send + kXHdr + 4 * i;
    if (i < cap_total) { pbatch[i] = tp[0]; psem[i] = tp[2]; pinst[i] = tp[3]; }
  }
}

__global__ __launch_bounds__(256) void xk_ids_kernel(const int32_t *__restrict__ hslot, int64_t n,
                                                     const int32_t *__restrict__ rank_of_slot,
                                                     const int32_t *__restrict__ gslot, int64_t *__restrict__ out,
                                                     const XCtrl *ctrl, int64_t *__restrict__ meta) {
  const int err = ctrl->error;
  if (blockIdx.x == 0 && threadIdx.x == 0) meta[2] = err;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = err ? 0 : (int64_t)gslot[rank_of_slot[hslot[i]]];
}

// ---------------------------------------------------------------------------------------------
// sums: both row sets, canonical order C2 (chunks of HSGK_CHUNK rows)
struct XChunk { int32_t base; int32_t ndist; int32_t slow; int32_t pad; int64_t lo; int64_t hi; };

// a wave's column slice: nmain lanes x vec columns from col0, then ntail single columns from tcol0
struct XSlice { const float *x; int d; int col0; int nmain; int vec; int tcol0; int ntail; int out0; };

// Slices of a d-column row.  Rows of (a multiple of) 256 columns plus at most 64 more take whole
// 1-KiB pieces per wave (float4 per lane; the remainder rides on the last slice as its scalar tail):
// 256 / 258-column rows are then ONE wave each and every request is a contiguous KiB (512-byte
// pieces streamed at 5.1 TB/s, whole rows at 5.9).  Otherwise slices of 128 columns (float2 per lane);
// a remainder <= 64 is the scalar tail of the last full slice, a larger one (or a row shorter than
// 128) is a slice of its own.
__host__ __device__ inline bool slices_wide(int d) { return d >= 256 && (d % 256) <= 64; }
__host__ __device__ inline int slices_count(int d) {
  if (d <= 0) return 0;
  if (slices_wide(d)) return d / 256;
  const int f = d / 128, rem = d % 128;
  return f + ((rem > 64 || (rem > 0 && f == 0)) ? 1 : 0);
}
__device__ inline XSlice slice_of(int w, const float *xa, int da, const float *xb, int db) {
  const int na = slices_count(da);
  const bool second = w >= na;
  const int d = second ? db : da;
  const int j = second ? w - na : w;
  XSlice s;
  s.x = second ? xb : xa;
  s.d = d;
  s.out0 = second ? da : 0;
  if (slices_wide(d)) {
    const int f = d / 256, rem = d % 256;
    s.vec = 4; s.col0 = 256 * j; s.nmain = 64; s.tcol0 = 256 * f; s.ntail = (j == f - 1) ? rem : 0;
    return s;
  }
  const int f = d / 128, rem = d % 128;
  s.vec = 2;
  if (j < f) {
    s.col0 = 128 * j; s.nmain = 64; s.tcol0 = 128 * f; s.ntail = (j == f - 1 && rem > 0 && rem <= 64) ? rem : 0;
  } else {
    s.col0 = 128 * f; s.nmain = rem / 2; s.tcol0 = 128 * f + 2 * (rem / 2); s.ntail = rem & 1;
  }
  return s;
}
static int slices_of_host(int d) { return slices_count(d); }

// U rows in flight per wave and buffer (two buffers)
// slot_first / slot_step: partial row of the list's first run and the distance between consecutive runs' rows
// (0 / 1 for the whole sorted list; rg / RG when the runs of a chunk are dealt out to RG groups of waves)
template <bool TAIL, int V, int U>
__device__ inline void stream_runs(const XSlice sl, int64_t row0, const uint16_t *__restrict__ srow, int nvalid,
                                   float *__restrict__ prow /* partial row 0 of this chunk */, int T,
                                   int slot_first = 0, int slot_step = 1) {
  typedef float fv __attribute__((ext_vector_type(V), aligned(4)));
  const int lane = threadIdx.x & 63;
  const int ml = lane < sl.nmain ? lane : (sl.nmain > 0 ? sl.nmain - 1 : 0);
  const int tl = lane < sl.ntail ? lane : (sl.ntail > 0 ? sl.ntail - 1 : 0);
  const float *xm = sl.x + row0 * sl.d + sl.col0 + V * ml;
  const float *xt = sl.x + row0 * sl.d + sl.tcol0 + tl;
  const bool has_main = sl.nmain > 0;
  fv va[U], vb[U];
  float ta[TAIL ? U : 1], tb[TAIL ? U : 1];
  auto issue = [&](int e0, fv (&v)[U], float (&t)[TAIL ? U : 1]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u < nvalid ? e0 + u : nvalid - 1;
      const int r = __builtin_amdgcn_readfirstlane((int)(srow[e] & 2047));
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const fv *>(xm + (int64_t)r * sl.d));
      if constexpr (TAIL) t[u] = __builtin_nontemporal_load(xt + (int64_t)r * sl.d);
    }
  };
  fv acc;
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.0f;
  float tacc = 0.0f;
  int slot = slot_first - slot_step;
  auto flush = [&]() {
    float *dst = prow + (int64_t)slot * T + sl.out0;
    if (has_main && lane < sl.nmain) *reinterpret_cast<fv *>(dst + sl.col0 + V * lane) = acc;
    if constexpr (TAIL) { if (lane < sl.ntail) dst[sl.tcol0 + lane] = tacc; }
  };
  auto fold = [&](int e0, const fv (&v)[U], const float (&t)[TAIL ? U : 1]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e0 + u < nvalid) {
        const int start = __builtin_amdgcn_readfirstlane((int)(srow[e0 + u] >> 15));
        if (start) {
          if (slot >= slot_first) flush();
          slot += slot_step;
#pragma unroll
          for (int i = 0; i < V; ++i) acc[i] = 0.0f;
          tacc = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = acc[i] + v[u][i];
        if constexpr (TAIL) tacc = tacc + t[u];
      }
    }
  };
  if (nvalid <= 0) return;
  issue(0, va, ta);
  for (int e0 = 0; e0 < nvalid; e0 += 2 * U) {
    issue(e0 + U, vb, tb);
    __builtin_amdgcn_sched_barrier(0);
    fold(e0, va, ta);
    __builtin_amdgcn_sched_barrier(0);
    issue(e0 + 2 * U, va, ta);
    __builtin_amdgcn_sched_barrier(0);
    fold(e0 + U, vb, tb);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (slot >= slot_first) flush();
}

// blockDim = 64 * (slices(da) + slices(db)) * RG; one workgroup per chunk.  RG > 1 (small inputs: a few chunks
// on 256 CUs, where one wave per column slice walking all 2048 rows of a chunk is a chain of ~130 exposed
// memory round trips): the chunk's RUNS are dealt out to RG groups of waves (run s to group s % RG), every
// group streams only its runs' rows -- a run is still summed by one wave in row order (order C2 unchanged).
// Dynamic LDS: RG lists of 2048 entries (none for RG = 1).
__global__ void xk_sums_chunk_kernel(const float *__restrict__ xa, int da, const float *__restrict__ xb, int db,
                                     const int64_t *__restrict__ ids, int64_t n, int64_t P,
                                     float *__restrict__ pool, int64_t *__restrict__ pool_ids, int pool_rows,
                                     XChunk *__restrict__ chunks, int32_t *__restrict__ seg_range, int32_t *pool_used,
                                     int RG, int32_t *__restrict__ status) {
  extern __shared__ uint16_t glist[];             // [RG][HSGK_CHUNK] (RG > 1)
  __shared__ int gcount[8];
  __shared__ unsigned long long skey[HSGK_CHUNK];
  __shared__ uint16_t srow[HSGK_CHUNK];
  __shared__ long long red[2][16];
  __shared__ int wcnt[16];
  __shared__ int sh_base, sh_nvalid, sh_ndist;
  const int c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  const int64_t row0 = (int64_t)c * HSGK_CHUNK;
  const int nrows = (int)((n - row0) < HSGK_CHUNK ? (n - row0) : HSGK_CHUNK);
  const int T = da + db;

  // id range of the chunk's valid rows
  long long lo = INT64_MAX, hi = INT64_MIN;
  bool bad = false;
  for (int r = tid; r < nrows; r += nt) {
    const long long l = ids[row0 + r];
    if (l >= 0 && l < P) { lo = l < lo ? l : lo; hi = l > hi ? l : hi; } else bad = true;
  }
  if (status && __ballot(bad) && lane == 0) atomicOr(status, 2);        // (such rows are skipped)
  for (int off = 32; off > 0; off >>= 1) {
    const long long olo = __shfl_xor(lo, off), ohi = __shfl_xor(hi, off);
    lo = olo < lo ? olo : lo;
    hi = ohi > hi ? ohi : hi;
  }
  if (lane == 0) { red[0][w] = lo; red[1][w] = hi; }
  __syncthreads();
  for (int i = 0; i < nw; ++i) { lo = red[0][i] < lo ? red[0][i] : lo; hi = red[1][i] > hi ? red[1][i] : hi; }
  if (hi < lo) {                                   // no valid row
    if (tid == 0) chunks[c] = XChunk{0, 0, 0, 0, 0, -1};
    return;
  }
  // stable sort of the rows by id: key = (id - lo) << 11 | row (ids < P <= 2^30), invalid rows last
  for (int r = tid; r < HSGK_CHUNK; r += nt) {
    unsigned long long k = ~0ull;
    if (r < nrows) {
      const long long l = ids[row0 + r];
      if (l >= 0 && l < P) k = ((unsigned long long)(l - lo) << 11) | (unsigned)r;
    }
    skey[r] = k;
  }
  __syncthreads();
  // Bitonic network.  A wave's 64 pairs of a step with j <= 64 lie inside ONE aligned block of 128 keys, and the same
  // wave owns that block in every such step (p = tid + m nt, nt a multiple of 64): those steps need no workgroup
  // barrier -- a wave's LDS accesses execute in order -- only the steps that cross blocks (j >= 128) and the hand-over
  // to them do: 15 barriers instead of 66.  With 16 waves at a barrier the sort was ~60 of the ~80 us this kernel
  // costs on ANY small input (12 544 rows of 16 or 128 columns alike).
  for (int k = 2; k <= HSGK_CHUNK; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < HSGK_CHUNK / 2; p += nt) {
        const int i = 2 * j * (p / j) + (p % j);
        const bool asc = (i & k) == 0;
        const unsigned long long a = skey[i], b = skey[i + j];
        if ((a > b) == asc) { skey[i] = b; skey[i + j] = a; }
      }
      if (j >= 128 || (j == 1 && k >= 128)) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  // runs: entries per thread are consecutive so that a block-wide exclusive scan gives the slots
  const int per = (HSGK_CHUNK + nt - 1) / nt;
  const int e0 = tid * per, e1 = (e0 + per) < HSGK_CHUNK ? (e0 + per) : HSGK_CHUNK;
  int starts = 0, valid = 0;
  for (int e = e0; e < e1; ++e) {
    const unsigned long long k = skey[e];
    if (k == ~0ull) break;
    ++valid;
    starts += (e == 0 || (skey[e - 1] >> 11) != (k >> 11)) ? 1 : 0;
  }
  int inc = starts, vinc = valid;
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off);
    if (lane >= off) inc += o;
  }
  for (int off = 32; off > 0; off >>= 1) vinc += __shfl_xor(vinc, off);
  if (lane == 63) wcnt[w] = inc;
  if (lane == 0) red[0][w] = vinc;
  __syncthreads();
  int before = inc - starts;
  for (int i = 0; i < w; ++i) before += wcnt[i];
  if (tid == 0) {
    int nd = 0, nv = 0;
    for (int i = 0; i < nw; ++i) { nd += wcnt[i]; nv += (int)red[0][i]; }
    int base = atomicAdd(pool_used, nd);
    if (base + nd > pool_rows) { atomicSub(pool_used, nd); base = -1; }
    sh_base = base; sh_ndist = nd; sh_nvalid = nv;
  }
  __syncthreads();
  const int base = sh_base, nvalid = sh_nvalid, ndist = sh_ndist;
  if (base < 0) {                                   // pool exhausted: the per-segment kernel sums this chunk's rows
    if (tid == 0) chunks[c] = XChunk{0, ndist, 1, 0, lo, hi};
    // its segments still need the chunk in their range
  }
  {
    int slot = before;
    for (int e = e0; e < e1; ++e) {
      const unsigned long long k = skey[e];
      if (k == ~0ull) break;
      const bool st = (e == 0 || (skey[e - 1] >> 11) != (k >> 11));
      srow[e] = (uint16_t)((k & 2047) | (st ? 0x8000u : 0u));
      if (st) {
        const long long id = lo + (long long)(k >> 11);
        if (base >= 0) pool_ids[base + slot] = id;
        atomicMin(&seg_range[2 * id], c);
        atomicMax(&seg_range[2 * id + 1], c);
        ++slot;
      }
    }
  }
  if (base < 0) return;
  if (tid == 0) chunks[c] = XChunk{base, ndist, 0, 0, lo, hi};
  __syncthreads();
  const int nsl = nw / RG;                          // column slices
  const int rg = w / nsl;
  const uint16_t *list = srow;
  int nlist = nvalid;
  if (RG > 1) {
    if (w % nsl == 0) {                             // the first wave of a group compacts the group's entries
      uint16_t *mine = glist + rg * HSGK_CHUNK;
      int runs_before = 0, pos = 0;
      for (int e0 = 0; e0 < nvalid; e0 += 64) {
        const int e = e0 + lane;
        const uint16_t ent = e < nvalid ? srow[e] : 0;
        const unsigned long long sm = __ballot(e < nvalid && (ent >> 15));
        const int run = runs_before + __popcll(sm & ((2ull << lane) - 1ull)) - 1;
        const bool take = e < nvalid && (run % RG) == rg;
        const unsigned long long tm = __ballot(take);
        if (take) mine[pos + __popcll(tm & ((1ull << lane) - 1ull))] = ent;
        pos += __popcll(tm);
        runs_before += __popcll(sm);
      }
      if (lane == 0) gcount[rg] = pos;
    }
    __syncthreads();
    list = glist + rg * HSGK_CHUNK;
    nlist = gcount[rg];
  }
  const XSlice sl = slice_of(w % nsl, xa, da, xb, db);
  float *prow = pool + (int64_t)base * T;
  if (sl.vec == 4) {
    if (sl.ntail > 0) stream_runs<true, 4, 8>(sl, row0, list, nlist, prow, T, rg, RG);
    else stream_runs<false, 4, 8>(sl, row0, list, nlist, prow, T, rg, RG);
  } else {
    if (sl.ntail > 0) stream_runs<true, 2, 16>(sl, row0, list, nlist, prow, T, rg, RG);
    else stream_runs<false, 2, 16>(sl, row0, list, nlist, prow, T, rg, RG);
  }
}

__global__ __launch_bounds__(256) void xk_seg_range_init_kernel(int32_t *__restrict__ seg_range, int64_t rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < rows) { seg_range[2 * i] = INT32_MAX; seg_range[2 * i + 1] = -1; }
}

// One workgroup per table row k: the chunk partials of segment k in chunk order (C2) -> table[k][0 .. T).
// A slow chunk has no partial row: its ids are scanned here and the matching rows summed in row order.
constexpr int kXListCap = 2048;
__global__ __launch_bounds__(256) void xk_sums_final_kernel(const float *__restrict__ pool, const int64_t *__restrict__ pool_ids,
                                                            const XChunk *__restrict__ chunks,
                                                            const int32_t *__restrict__ seg_range,
                                                            const float *__restrict__ xa, int da,
                                                            const float *__restrict__ xb, int db,
                                                            const int64_t *__restrict__ ids, int64_t n,
                                                            float *__restrict__ table) {
  __shared__ int32_t clist[kXListCap][2];
  __shared__ uint16_t rows[HSGK_CHUNK];
  __shared__ int wn[4];
  __shared__ int total;
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = da + db;
  const int c_first = seg_range[2 * k], c_last = seg_range[2 * k + 1];
  constexpr int NC = 4;                             // columns tid + 256 u (T <= 1024)
  float t[NC] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int cb = c_first; cb <= c_last; cb += kXListCap) {
    // list of the chunks of this block that hold segment k, in chunk order
    __syncthreads();
    if (tid == 0) total = 0;
    __syncthreads();
    const int cend = (cb + kXListCap - 1) < c_last ? (cb + kXListCap - 1) : c_last;
    for (int c0 = cb; c0 <= cend; c0 += 256) {
      const int c = c0 + tid;
      bool hit = false;
      int slot = -1;
      if (c <= cend) {
        const XChunk ch = chunks[c];
        if (k >= ch.lo && k <= ch.hi) {
          if (ch.slow) {
            hit = true;
          } else {
            const int64_t *cid = pool_ids + ch.base;
            int a = 0, b = ch.ndist - 1;
            while (a < b) {
              const int m = (a + b) >> 1;
              if (cid[m] < k) a = m + 1; else b = m;
            }
            if (cid[a] == k) { hit = true; slot = ch.base + a; }
          }
        }
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wn[w] = __popcll(m);
      __syncthreads();
      int base = total;
      for (int i = 0; i < w; ++i) base += wn[i];
      if (hit) {
        const int q = base + __popcll(m & ((1ull << lane) - 1ull));
        clist[q][0] = c;
        clist[q][1] = slot;
      }
      __syncthreads();
      if (tid == 0) total += wn[0] + wn[1] + wn[2] + wn[3];
      __syncthreads();
    }
    const int nl = total;
    int q = 0;
    // batches of 8 partial rows in flight (all fast), added in order
    while (q < nl) {
      if (clist[q][1] >= 0) {
        int len = 1;
        while (len < 8 && q + len < nl && clist[q + len][1] >= 0) ++len;
        float v[8][NC];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int slot = clist[q + (e < len ? e : len - 1)][1];
#pragma unroll
          for (int u = 0; u < NC; ++u) {
            const int i = u * 256 + tid;
            v[e][u] = i < T ? pool[(int64_t)slot * T + i] : 0.0f;
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < len) {
#pragma unroll
            for (int u = 0; u < NC; ++u) t[u] = t[u] + v[e][u];
          }
        q += len;
      } else {
        const int c = clist[q][0];
        const int64_t r0 = (int64_t)c * HSGK_CHUNK;
        const int nr = (int)((n - r0) < HSGK_CHUNK ? (n - r0) : HSGK_CHUNK);
        __syncthreads();
        if (tid == 0) total = 0;
        __syncthreads();
        for (int b0 = 0; b0 < nr; b0 += 256) {
          const int r = b0 + tid;
          const bool mine = r < nr && ids[r0 + r] == k;
          const unsigned long long m = __ballot(mine);
          if (lane == 0) wn[w] = __popcll(m);
          __syncthreads();
          int base = total;
          for (int i = 0; i < w; ++i) base += wn[i];
          if (mine) rows[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)r;
          __syncthreads();
          if (tid == 0) total += wn[0] + wn[1] + wn[2] + wn[3];
          __syncthreads();
        }
        const int nm = total;
#pragma unroll
        for (int u = 0; u < NC; ++u) {
          const int i = u * 256 + tid;
          if (i < T) {
            const float *x = i < da ? xa + i : xb + (i - da);
            const int d = i < da ? da : db;
            float sacc = 0.0f;
            for (int e = 0; e < nm; ++e) sacc = sacc + x[(r0 + rows[e]) * d];
            t[u] = t[u] + sacc;
          }
        }
        __syncthreads();
        ++q;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NC; ++u) {
    const int i = u * 256 + tid;
    if (i < T) table[k * T + i] = t[u];
  }
}

// table [rows][da + db] raw sums -> pa [rows][da], pb [rows][db] normalised (eps-clamped norm, C1 chain),
// norms [rows][2]
__global__ __launch_bounds__(256) void xk_normalize_kernel(const float *__restrict__ table, int da, int db, float eps,
                                                           float *__restrict__ pa, float *__restrict__ pb,
                                                           float *__restrict__ norms) {
  extern __shared__ float row[];                  // [da + db + 2]
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x;
  const int T = da + db;
  for (int i = tid; i < T; i += 256) row[i] = table[k * T + i];
  __syncthreads();
  if (tid == 0 || tid == 64) {
    const int o = tid ? da : 0, d = tid ? db : da;
    float ss = 0.0f;
    for (int i = 0; i < d; ++i) ss = fmaf(row[o + i], row[o + i], ss);
    float a = sqrtf(ss);
    if (!(a >= eps)) a = eps;
    row[T + (tid ? 1 : 0)] = a;
    if (norms) norms[2 * k + (tid ? 1 : 0)] = a;
  }
  __syncthreads();
  const float na = row[T], nb = row[T + 1];
  for (int i = tid; i < da; i += 256) pa[k * da + i] = row[i] / na;
  for (int i = tid; i < db; i += 256) pb[k * db + i] = row[da + i] / nb;
}

// ---------------------------------------------------------------------------------------------
// RCCL, bound at run time
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static const RcclApi *rccl() {
  std::call_once(g_rccl_once, [] {
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names)                    // the instance this process already uses (torch's)
      if (!h) h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
    for (const char *nm : names)
      if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    RcclApi a;
    a.handle = h;
#define HSGK_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
    HSGK_SYM(GetUniqueId, "ncclGetUniqueId");
    HSGK_SYM(CommInitRank, "ncclCommInitRank");
    HSGK_SYM(CommDestroy, "ncclCommDestroy");
    HSGK_SYM(CommCount, "ncclCommCount");
    HSGK_SYM(CommUserRank, "ncclCommUserRank");
    HSGK_SYM(AllGather, "ncclAllGather");
    HSGK_SYM(AllReduce, "ncclAllReduce");
    HSGK_SYM(GetErrorString, "ncclGetErrorString");
#undef HSGK_SYM
    if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce) g_rccl = a;
  });
  return g_rccl.handle ? &g_rccl : nullptr;
}

#define HSGK_CHECK_NCCL(expr)                                                                  \
  do {                                                                                         \
    ncclResult_t r__ = (expr);                                                                 \
    if (r__ != ncclSuccess) {                                                                  \
      hsgk::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,                           \
                      api->GetErrorString ? api->GetErrorString(r__) : "rccl error");          \
      return -4;                                                                               \
    }                                                                                          \
  } while (0)

// ---------------------------------------------------------------------------------------------
struct XWs {
  XCtrl *ctrl;
  long long *table;          // [hsize] open-addressing key table
  int32_t *hslot;            // [n]
  long long *ckeys;          // [capp]
  int32_t *cvals;            // [capp]
  int32_t *rank_of_slot;     // [hsize]
  int64_t *send;             // [kXHdr + 4 cap]
  int64_t *recv;             // [world][kXHdr + 4 cap]
  long long *gkey;           // [world][cap]
  int32_t *prim;             // [world][cap + 1]
  int32_t *gslot;            // [world][cap]
  XChunk *chunks;            // [nch]
  int32_t *seg_range;        // [cap_total][2]
  int64_t *pool_ids;         // [pool_rows]
  float *pool;               // [pool_rows][T]
  int hsize;
  int64_t capp;
  int nch;
};

static int64_t pow2_at_least(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

static void carve_exchange(Carver &cv, int64_t n, int T, int64_t cap, int64_t cap_total, int world, int64_t pool_rows,
                           XWs *w) {
  const int64_t capp = pow2_at_least(cap < kSortTile ? kSortTile : cap);
  const int64_t hsize = pow2_at_least(2 * cap < 1024 ? 1024 : 2 * cap);
  const int64_t nch = (n + HSGK_CHUNK - 1) / HSGK_CHUNK;
  w->hsize = (int)hsize;
  w->capp = capp;
  w->nch = (int)nch;
  w->ctrl = cv.take<XCtrl>(1);
  w->table = cv.take<long long>((size_t)hsize);
  w->hslot = cv.take<int32_t>((size_t)(n > 0 ? n : 1));
  w->ckeys = cv.take<long long>((size_t)capp);
  w->cvals = cv.take<int32_t>((size_t)capp);
  w->rank_of_slot = cv.take<int32_t>((size_t)hsize);
  w->send = cv.take<int64_t>((size_t)(kXHdr + 4 * cap));
  w->recv = cv.take<int64_t>((size_t)world * (kXHdr + 4 * cap));
  w->gkey = cv.take<long long>((size_t)world * cap);
  w->prim = cv.take<int32_t>((size_t)world * (cap + 1));
  w->gslot = cv.take<int32_t>((size_t)world * cap);
  w->chunks = cv.take<XChunk>((size_t)(nch > 0 ? nch : 1));
  w->seg_range = cv.take<int32_t>((size_t)2 * (cap_total > 0 ? cap_total : 1));
  w->pool_ids = cv.take<int64_t>((size_t)(pool_rows > 0 ? pool_rows : 1));
  w->pool = cv.take<float>((size_t)(pool_rows > 0 ? pool_rows : 1) * T);
}

static int check_args(const hsgk_exchange_args *a, int world) {
  HSGK_REQUIRE(a != nullptr, "null args");
  HSGK_REQUIRE(a->n >= 0 && a->C >= 1 && a->D >= 1, "bad sizes");
  HSGK_REQUIRE(a->C + a->D <= 1024, "rows longer than 1024 columns in total are not supported");
  HSGK_REQUIRE(a->cap_local >= 1 && a->cap_local <= ((int64_t)1 << 28), "cap_local out of range");
  HSGK_REQUIRE(a->cap_total >= 1 && a->cap_total <= ((int64_t)1 << 30), "cap_total out of range");
  HSGK_REQUIRE(a->n < ((int64_t)1 << 40), "too many rows");
  HSGK_REQUIRE(world >= 1 && world <= 4096, "bad world size");
  HSGK_REQUIRE(a->pool_rows >= 0 && a->pool_rows < ((int64_t)1 << 31), "pool_rows out of range");
  HSGK_REQUIRE(a->meta != nullptr && a->workspace != nullptr, "meta / workspace required");
  HSGK_REQUIRE(a->workspace_bytes >= hsgk_exchange_workspace_bytes(a->n, a->C, a->D, a->cap_local, a->cap_total, world,
                                                                   a->pool_rows),
               "workspace too small");
  return 0;
}

static XWs carve_from(const hsgk_exchange_args *a, int world) {
  XWs w;
  Carver cv(a->workspace);
  carve_exchange(cv, a->n, a->C + a->D, a->cap_local, a->cap_total, world, a->pool_rows, &w);
  return w;
}

// the per-call fills in ONE launch (four memsets before: empty hash slots = -1, unused sort keys = 0x7F.. (sort last),
// control block and meta = 0); small exchanges are launch-bound and run several times per training step
__global__ __launch_bounds__(256) void xk_init_kernel(XCtrl *__restrict__ ctrl, long long *__restrict__ table,
                                                      int64_t hsize, long long *__restrict__ ckeys, int64_t capp,
                                                      int64_t *__restrict__ meta) {
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
  for (int64_t i = i0; i < hsize; i += step) table[i] = -1ll;
  for (int64_t i = i0; i < capp; i += step) ckeys[i] = 0x7F7F7F7F7F7F7F7Fll;
  if (i0 < (int64_t)(sizeof(XCtrl) / 4)) reinterpret_cast<int32_t *>(ctrl)[i0] = 0;
  if (i0 < 8) meta[i0] = 0;
}

static int launch_keys(const hsgk_exchange_args *a, const XWs &w, hipStream_t s) {
  {
    const int64_t most = (int64_t)w.hsize > (int64_t)w.capp ? (int64_t)w.hsize : (int64_t)w.capp;
    const int64_t g = (most + 256 * 4 - 1) / (256 * 4);
    hipLaunchKernelGGL(xk_init_kernel, dim3((unsigned)(g < 1 ? 1 : g > 2048 ? 2048 : g)), dim3(256), 0, s, w.ctrl,
                       w.table, (int64_t)w.hsize, w.ckeys, (int64_t)w.capp, a->meta);
    HSGK_LAUNCH_CHECK();
  }
  if (a->n > 0) {
    const int64_t g = (a->n + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(xk_ranges_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, s, a->cluster, a->batch,
                       a->semantic, a->instance, a->n, w.ctrl);
    HSGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(xk_insert_kernel, dim3((unsigned)((a->n + kXKeyRows - 1) / kXKeyRows)), dim3(256), 0, s,
                       a->cluster, a->batch, a->semantic, a->instance, a->n, w.table, w.hsize - 1, w.hslot, w.ctrl);
    HSGK_LAUNCH_CHECK();
    hipLaunchKernelGGL(xk_compact_kernel, dim3((unsigned)((w.hsize + 255) / 256)), dim3(256), 0, s, w.table, w.hsize,
                       (int)a->cap_local, w.ckeys, w.cvals, w.ctrl);
    HSGK_LAUNCH_CHECK();
    if (int rc = launch_sort_pairs(w.ckeys, w.cvals, w.capp, w.ctrl, s)) return rc;
  }
  hipLaunchKernelGGL(xk_rank_kernel, dim3((unsigned)((a->cap_local + 255) / 256)), dim3(256), 0, s, w.ckeys, w.cvals,
                     (int)a->cap_local, w.rank_of_slot, w.send, w.ctrl);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// recv: [world] blocks; my_rank < 0: list mode (gslot for every source, meta[0] untouched)
static int launch_merge(const int64_t *recv, int world, int my_rank, int64_t cap, int64_t cap_total, const XWs &w,
                        int32_t *gslot, int64_t *psem, int64_t *pinst, int64_t *pbatch, int64_t *meta, hipStream_t s) {
  const unsigned gx = (unsigned)((cap + 255) / 256 > 64 ? 64 : (cap + 255) / 256);
  hipLaunchKernelGGL(xk_merge_radix_kernel, dim3(gx, world), dim3(256), 0, s, recv, world, (int)cap, w.ctrl);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(xk_merge_pack_kernel, dim3(gx, world), dim3(256), 0, s, recv, (int)cap, w.gkey, w.ctrl);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(xk_merge_primary_kernel, dim3(gx, world), dim3(256), 0, s, recv, (int)cap, w.gkey, w.prim);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(xk_merge_scan_kernel, dim3(world), dim3(256), 0, s, recv, (int)cap, w.prim);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(xk_merge_place_kernel, dim3(gx, world), dim3(256), 0, s, recv, world, (int)cap, w.gkey, w.prim,
                     gslot, cap_total, psem, pinst, pbatch, w.ctrl, meta, my_rank);
  HSGK_LAUNCH_CHECK();
  return 0;
}

// Segment sums of one or two row sets by int64 ids in [0, P), canonical order C2, into table [P][da + db]
// (db = 0: one row set).  Scratch: chunks [nch], seg_range [P][2], pool_ids [pool_rows], pool [pool_rows][da + db],
// pool_used (zeroed by the caller).  status (nullable): bit 1 set when an id lies outside [0, P).
int launch_sorted_sums(const float *xa, int da, const float *xb, int db, const int64_t *ids, int64_t n, int64_t P,
                       void *chunks_v, int32_t *seg_range, int64_t *pool_ids, float *pool, int64_t pool_rows,
                       int32_t *pool_used, float *table, int32_t *status, hipStream_t s) {
  XChunk *chunks = static_cast<XChunk *>(chunks_v);
  const int nch = (int)((n + HSGK_CHUNK - 1) / HSGK_CHUNK);
  if (P <= 0) return 0;
  hipLaunchKernelGGL(xk_seg_range_init_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, seg_range, P);
  HSGK_LAUNCH_CHECK();
  if (nch > 0) {
    const int nw = slices_of_host(da) + (db > 0 ? slices_of_host(db) : 0);
    HSGK_REQUIRE(nw >= 1 && nw <= 16, "rows too long");
    int rg = 1;                                     // run groups per chunk: small inputs only (see the kernel)
    if (nch <= 512) rg = nw <= 2 ? 8 : nw <= 4 ? 4 : nw <= 8 ? 2 : 1;
    hipLaunchKernelGGL(xk_sums_chunk_kernel, dim3((unsigned)nch), dim3(64 * nw * rg),
                       rg > 1 ? (size_t)rg * HSGK_CHUNK * 2 : 0, s, xa, da, xb, db, ids, n, P, pool, pool_ids,
                       (int)pool_rows, chunks, seg_range, pool_used, rg, status);
    HSGK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(xk_sums_final_kernel, dim3((unsigned)P), dim3(256), 0, s, pool, pool_ids, chunks, seg_range, xa, da,
                     xb, db, ids, n, table);
  HSGK_LAUNCH_CHECK();
  return 0;
}
size_t sorted_sums_chunk_bytes() { return sizeof(XChunk); }

static int launch_ids_and_sums(const hsgk_exchange_args *a, const XWs &w, const int32_t *gslot_mine, hipStream_t s) {
  const int T = a->C + a->D;
  {
    const int64_t g = a->n > 0 ? (a->n + 256 * 4 - 1) / (256 * 4) : 1;
    hipLaunchKernelGGL(xk_ids_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, w.hslot, a->n,
                       w.rank_of_slot, gslot_mine, a->updated_cluster, w.ctrl, a->meta);
    HSGK_LAUNCH_CHECK();
  }
  if (int rc = launch_sorted_sums(a->embeddings, a->C, a->embeddings_loc, a->D, a->updated_cluster, a->n, a->cap_total,
                                  w.chunks, w.seg_range, w.pool_ids, w.pool, a->pool_rows, &w.ctrl->pool_used, a->table,
                                  nullptr, s))
    return rc;
  (void)T;
  return 0;
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

size_t hsgk_exchange_workspace_bytes(int64_t n, int C, int D, int64_t cap_local, int64_t cap_total, int world,
                                     int64_t pool_rows) {
  if (n < 0 || C < 1 || D < 1 || cap_local < 1 || cap_total < 1 || world < 1 || pool_rows < 0) return 0;
  XWs w;
  Carver cv(nullptr);
  carve_exchange(cv, n, C + D, cap_local, cap_total, world, pool_rows, &w);
  return cv.off + 256;
}

int hsgk_exchange_keys(const hsgk_exchange_args *a, int world, hsgk_stream_t stream) {
  if (int rc = check_args(a, world)) return rc;
  (void)hipGetLastError();
  const XWs w = carve_from(a, world);
  return launch_keys(a, w, static_cast<hipStream_t>(stream));
}

const int64_t *hsgk_exchange_send_block(const hsgk_exchange_args *a, int world, size_t *bytes) {
  if (check_args(a, world)) return nullptr;
  const XWs w = carve_from(a, world);
  if (bytes) *bytes = (size_t)(kXHdr + 4 * a->cap_local) * 8;
  return w.send;
}

int64_t *hsgk_exchange_recv_blocks(const hsgk_exchange_args *a, int world) {
  if (check_args(a, world)) return nullptr;
  const XWs w = carve_from(a, world);
  return w.recv;
}

int hsgk_exchange_merge(const hsgk_exchange_args *a, int my_rank, int world, int32_t *slots_out, hsgk_stream_t stream) {
  if (int rc = check_args(a, world)) return rc;
  HSGK_REQUIRE(my_rank < world, "rank outside the world");
  (void)hipGetLastError();
  const XWs w = carve_from(a, world);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (world == 1 && my_rank == 0) {
    hipLaunchKernelGGL(xk_single_place_kernel, dim3((unsigned)((a->cap_local + 255) / 256 > 64 ? 64 : (a->cap_local + 255) / 256)),
                       dim3(256), 0, s, w.send, (int)a->cap_local, w.gslot, a->cap_total, a->proto_semantic,
                       a->proto_instance, a->proto_batch, w.ctrl, a->meta);
    HSGK_LAUNCH_CHECK();
    return 0;
  }
  return launch_merge(w.recv, world, my_rank, a->cap_local, a->cap_total, w, slots_out ? slots_out : w.gslot,
                      a->proto_semantic, a->proto_instance, a->proto_batch, a->meta, s);
}

int hsgk_exchange_sums(const hsgk_exchange_args *a, int my_rank, int world, const int32_t *slots, hsgk_stream_t stream) {
  if (int rc = check_args(a, world)) return rc;
  HSGK_REQUIRE(a->table != nullptr && (a->updated_cluster != nullptr || a->n == 0), "table / updated_cluster required");
  (void)hipGetLastError();
  const XWs w = carve_from(a, world);
  const int32_t *mine = slots ? slots : w.gslot + (int64_t)(my_rank > 0 ? my_rank : 0) * a->cap_local;
  return launch_ids_and_sums(a, w, mine, static_cast<hipStream_t>(stream));
}

int hsgk_exchange_finish(const hsgk_exchange_args *a, int64_t rows, void *comm, int world, hsgk_stream_t stream) {
  HSGK_REQUIRE(a != nullptr && a->table != nullptr, "table required");
  HSGK_REQUIRE(rows >= 0 && rows <= a->cap_total, "rows outside the table");
  HSGK_REQUIRE(a->prototypes != nullptr && a->prototypes_loc != nullptr, "outputs required");
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows == 0) return 0;
  const int T = a->C + a->D;
  if (world > 1 && comm != nullptr) {
    const RcclApi *api = rccl();
    HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
    HSGK_CHECK_NCCL(api->AllReduce(a->table, a->table, (size_t)rows * T, ncclFloat, ncclSum,
                                   static_cast<ncclComm_t>(comm), s));
  }
  hipLaunchKernelGGL(xk_normalize_kernel, dim3((unsigned)rows), dim3(256), (size_t)(T + 2) * 4, s, a->table, a->C, a->D,
                     a->eps, a->prototypes, a->prototypes_loc, a->norms);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_exchange_begin(const hsgk_exchange_args *a, void *comm, int rank, int world, hsgk_stream_t stream) {
  if (int rc = check_args(a, world)) return rc;
  HSGK_REQUIRE(rank >= 0 && rank < world, "rank outside the world");
  HSGK_REQUIRE(world == 1 || comm != nullptr, "a communicator is required for world > 1");
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  const XWs w = carve_from(a, world);
  if (int rc = launch_keys(a, w, s)) return rc;
  if (world > 1) {
    const RcclApi *api = rccl();
    HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
    HSGK_CHECK_NCCL(api->AllGather(w.send, w.recv, (size_t)(kXHdr + 4 * a->cap_local), ncclInt64,
                                   static_cast<ncclComm_t>(comm), s));
  }
  if (int rc = hsgk_exchange_merge(a, rank, world, nullptr, stream)) return rc;
  return hsgk_exchange_sums(a, rank, world, nullptr, stream);
}

int hsgk_exchange_prototypes(const hsgk_exchange_args *a, void *comm, int rank, int world, hsgk_stream_t stream) {
  if (int rc = hsgk_exchange_begin(a, comm, rank, world, stream)) return rc;
  return hsgk_exchange_finish(a, a->cap_total, comm, world, stream);
}

// ---- communicator helpers (host) ------------------------------------------------------------------
int hsgk_comm_unique_id(void *id_out, size_t bytes) {
  const RcclApi *api = rccl();
  HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
  HSGK_REQUIRE(id_out != nullptr && bytes >= sizeof(ncclUniqueId), "id buffer too small (HSGK_COMM_ID_BYTES)");
  ncclUniqueId id;
  HSGK_CHECK_NCCL(api->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int hsgk_comm_init_rank(void **comm_out, int world, int rank, const void *id_in, size_t bytes) {
  const RcclApi *api = rccl();
  HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
  HSGK_REQUIRE(comm_out != nullptr && id_in != nullptr && bytes >= sizeof(ncclUniqueId), "bad arguments");
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof(id));
  ncclComm_t c = nullptr;
  HSGK_CHECK_NCCL(api->CommInitRank(&c, world, id, rank));
  *comm_out = c;
  return 0;
}

int hsgk_comm_destroy(void *comm) {
  const RcclApi *api = rccl();
  HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
  if (comm) HSGK_CHECK_NCCL(api->CommDestroy(static_cast<ncclComm_t>(comm)));
  return 0;
}

int hsgk_comm_all_reduce_f32(float *buf, int64_t count, void *comm, hsgk_stream_t stream) {
  const RcclApi *api = rccl();
  HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
  HSGK_REQUIRE(buf != nullptr && count >= 0 && comm != nullptr, "bad arguments");
  if (count == 0) return 0;
  HSGK_CHECK_NCCL(api->AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, static_cast<ncclComm_t>(comm),
                                 static_cast<hipStream_t>(stream)));
  return 0;
}

int hsgk_comm_all_gather_bytes(const void *send, void *recv, size_t bytes_per_rank, void *comm, hsgk_stream_t stream) {
  const RcclApi *api = rccl();
  HSGK_REQUIRE(api != nullptr, "librccl could not be loaded");
  HSGK_REQUIRE(send != nullptr && recv != nullptr && comm != nullptr, "bad arguments");
  if (bytes_per_rank == 0) return 0;
  HSGK_CHECK_NCCL(api->AllGather(send, recv, bytes_per_rank, ncclInt8, static_cast<ncclComm_t>(comm),
                                 static_cast<hipStream_t>(stream)));
  return 0;
}

}  // extern "C"
