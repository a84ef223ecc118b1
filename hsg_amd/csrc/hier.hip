// hier.hip -- hierarchical segment assignment (reference
// hsg/models/embeddings/resnet_fcn_hsg.py):
//   hier_assign   :638-672  softmax over clusters, argmax -> fine labels;
//                           coarse probs = softmax(coarse) x fine probs
//                           (Bayes chain, einsum 'bij,bjk->bik'), argmax
//   group_mean    :683-748  _collect_nd_coarser_prototype: masked scatter-mean
//                           of [B,C,N] node features into groups (+ optional
//                           L2 normalisation)
//   gather_labels :751-780  pixel -> segment -> group label lookup
// All three are launch-bound in the reference (tens of tiny ATen ops and a
// Python loop per image); here each is one launch, one workgroup per image.
// Canonical arithmetic: sums run sequentially in ascending index, dot products
// are fmaf chains (C1), softmax = exp(x - max) / sum with expf.
#include "common.h"

namespace hsgk {

__global__ __launch_bounds__(256) void hier_assign_kernel(
    const float *__restrict__ fine_logits, const float *__restrict__ coarse_logits, int KF, int KC,
    int N, float *__restrict__ fine_prob, int64_t *__restrict__ fine_lab,
    float *__restrict__ coarse_prob, int64_t *__restrict__ coarse_lab) {
  extern __shared__ float pc[];              // [KC][KF] coarse softmax (over KC, per fine column)
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *fl = fine_logits + (int64_t)b * KF * N;
  float *fp = fine_prob + (int64_t)b * KF * N;
  if (coarse_logits) {
    const float *cl = coarse_logits + (int64_t)b * KC * KF;
    for (int c1 = tid; c1 < KF; c1 += 256) {
      float m = -INFINITY;
      for (int c2 = 0; c2 < KC; ++c2) m = fmaxf(m, cl[c2 * KF + c1]);
      float s = 0.0f;
      for (int c2 = 0; c2 < KC; ++c2) s = s + expf(cl[c2 * KF + c1] - m);
      for (int c2 = 0; c2 < KC; ++c2) pc[c2 * KF + c1] = expf(cl[c2 * KF + c1] - m) / s;
    }
  }
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    float m = -INFINITY;
    for (int c = 0; c < KF; ++c) m = fmaxf(m, fl[c * N + n]);
    float s = 0.0f;
    for (int c = 0; c < KF; ++c) s = s + expf(fl[c * N + n] - m);
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < KF; ++c) {
      const float p = expf(fl[c * N + n] - m) / s;
      fp[c * N + n] = p;
      if (p > best) { best = p; bi = c; }
    }
    fine_lab[(int64_t)b * N + n] = bi;
    if (coarse_logits) {
      float *cp = coarse_prob + (int64_t)b * KC * N;
      float cbest = -INFINITY;
      int ci = 0;
      for (int c2 = 0; c2 < KC; ++c2) {
        float acc = 0.0f;
        for (int c = 0; c < KF; ++c) acc = fmaf(pc[c2 * KF + c], fp[c * N + n], acc);
        cp[c2 * N + n] = acc;
        if (acc > cbest) { cbest = acc; ci = c2; }
      }
      coarse_lab[(int64_t)b * N + n] = ci;
    }
  }
}

// Backward of hier_assign_kernel's two probability outputs (the labels carry no gradient):
//   pf = softmax_KF(fine), pc = softmax_KC(coarse) per fine column, coarse_prob = pc x pf
//   g_pf[f][n] = G1[f][n] + sum_c pc[c][f] G2[c][n]
//   g_fine[f][n]   = pf[f][n] (g_pf[f][n] - sum_f' pf[f'][n] g_pf[f'][n])
//   g_pc[c][f]     = sum_n G2[c][n] pf[f][n]
//   g_coarse[c][f] = pc[c][f] (g_pc[c][f] - sum_c' pc[c'][f] g_pc[c'][f])
// One workgroup per image: thread = node for the fine part, the node sum of g_pc through per-thread partials
// summed in a fixed order (wave shuffles, then the four waves in order).  KF * KC <= 1024.
__global__ __launch_bounds__(256) void hier_assign_bwd_kernel(
    const float *__restrict__ fine_logits, const float *__restrict__ coarse_logits, int KF, int KC, int N,
    const float *__restrict__ g_fprob, const float *__restrict__ g_cprob, float *__restrict__ g_fine,
    float *__restrict__ g_coarse) {
  extern __shared__ float hb[];
  float *pc = hb;                              // [KC][KF]
  float *gpc = hb + (size_t)KC * KF;           // [4 waves][KC][KF] partials, then [KC][KF] totals in the first slab
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float *fl = fine_logits + (int64_t)b * KF * N;
  const float *g1 = g_fprob ? g_fprob + (int64_t)b * KF * N : nullptr;
  const float *g2 = g_cprob ? g_cprob + (int64_t)b * KC * N : nullptr;
  const bool coarse = coarse_logits != nullptr && g2 != nullptr;
  if (coarse) {
    const float *cl = coarse_logits + (int64_t)b * KC * KF;
    for (int f = tid; f < KF; f += 256) {
      float m = -INFINITY;
      for (int c = 0; c < KC; ++c) m = fmaxf(m, cl[c * KF + f]);
      float sum = 0.0f;
      for (int c = 0; c < KC; ++c) sum = sum + expf(cl[c * KF + f] - m);
      for (int c = 0; c < KC; ++c) pc[c * KF + f] = expf(cl[c * KF + f] - m) / sum;
    }
    for (int i = tid; i < 4 * KC * KF; i += 256) gpc[i] = 0.0f;
  }
  __syncthreads();
  for (int n0 = 0; n0 < N; n0 += 256) {
    const int n = n0 + tid;
    const bool live = n < N;
    // softmax of the node's fine column (recomputed as the forward computed it)
    float m = -INFINITY, sum = 0.0f;
    if (live) {
      for (int f = 0; f < KF; ++f) m = fmaxf(m, fl[f * N + n]);
      for (int f = 0; f < KF; ++f) sum = sum + expf(fl[f * N + n] - m);
    }
    float dot = 0.0f;
    if (live)
      for (int f = 0; f < KF; ++f) {
        const float p = expf(fl[f * N + n] - m) / sum;
        float g = g1 ? g1[f * N + n] : 0.0f;
        if (coarse)
          for (int c = 0; c < KC; ++c) g = fmaf(pc[c * KF + f], g2[c * N + n], g);
        dot = fmaf(p, g, dot);
      }
    if (live)
      for (int f = 0; f < KF; ++f) {
        const float p = expf(fl[f * N + n] - m) / sum;
        float g = g1 ? g1[f * N + n] : 0.0f;
        if (coarse)
          for (int c = 0; c < KC; ++c) g = fmaf(pc[c * KF + f], g2[c * N + n], g);
        g_fine[(int64_t)b * KF * N + f * N + n] = p * (g - dot);
      }
    if (coarse) {
      // g_pc[c][f] += sum over the 256 nodes of this step: wave reduction, one partial per wave
      for (int c = 0; c < KC; ++c)
        for (int f = 0; f < KF; ++f) {
          float v = live ? g2[c * N + n] * (expf(fl[f * N + n] - m) / sum) : 0.0f;
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if (lane == 0) gpc[(w * KC + c) * KF + f] += v;
        }
    }
  }
  if (!coarse) return;
  __syncthreads();
  for (int i = tid; i < KC * KF; i += 256) gpc[i] = ((gpc[i] + gpc[KC * KF + i]) + gpc[2 * KC * KF + i]) + gpc[3 * KC * KF + i];
  __syncthreads();
  for (int f = tid; f < KF; f += 256) {
    float dot = 0.0f;
    for (int c = 0; c < KC; ++c) dot = fmaf(pc[c * KF + f], gpc[c * KF + f], dot);
    for (int c = 0; c < KC; ++c)
      g_coarse[(int64_t)b * KC * KF + c * KF + f] = pc[c * KF + f] * (gpc[c * KF + f] - dot);
  }
}

// prototypes [B,C,N]; labels [B,N]; masks [B,N] (uint8, nullable: nothing
// padded); out [B,C,G].  Node n contributes to group labels[n] unless padded.
// One wave per (image, 64 channels): thread = channel walks the N nodes ONCE in ascending order and adds every
// value to its group's LDS cell (its own column: no conflicts) -- per (group, channel) the same sequential sum as a
// loop over groups, at 1 / G of the work and B * C / 64 workgroups instead of B (cfg4's 256 -> 64 grouping of four
// images: 0.98 -> ~0.03 ms).  Normalisation needs the whole channel range of a group: second, tiny kernel.
__global__ __launch_bounds__(64) void group_mean_kernel(
    const float *__restrict__ protos, const int64_t *__restrict__ labels,
    const uint8_t *__restrict__ masks, int C, int N, int G, float *__restrict__ out) {
  extern __shared__ float sm[];              // [G][64] sums, [G] counts, int labels [N]
  float *sums = sm;
  float *cnt = sm + (size_t)G * 64;
  int *lab = reinterpret_cast<int *>(cnt + G);
  const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int n = tid; n < N; n += 64) {
    const bool pad = masks ? masks[(int64_t)b * N + n] != 0 : false;
    const int64_t l = labels[(int64_t)b * N + n];
    lab[n] = (pad || l < 0 || l >= G) ? -1 : (int)l;
  }
  for (int i = tid; i < G * 64; i += 64) sums[i] = 0.0f;
  __syncthreads();
  for (int g = tid; g < G; g += 64) {
    float k = 0.0f;
    for (int n = 0; n < N; ++n) k = k + (lab[n] == g ? 1.0f : 0.0f);
    cnt[g] = k;
  }
  const int c = c0 + tid;
  if (c < C) {
    const float *p = protos + ((int64_t)b * C + c) * N;
    for (int n0 = 0; n0 < N; n0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[min(n0 + u, N - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = n0 + u;
        if (n < N) {
          const int g = lab[n];
          if (g >= 0) sums[g * 64 + tid] = sums[g * 64 + tid] + v[u];
        }
      }
    }
  }
  __syncthreads();
  if (c < C) {
    float *o = out + ((int64_t)b * C + c) * G;
    for (int g = 0; g < G; ++g) o[g] = sums[g * 64 + tid] / fmaxf(cnt[g], 1e-12f);
  }
}

// out [B][C][G] /= max(|column g|, eps), the norm as the C1 chain over ascending channels
__global__ __launch_bounds__(64) void group_normalize_kernel(float *__restrict__ out, int C, int G, float eps) {
  const int b = blockIdx.y, g = blockIdx.x * 64 + threadIdx.x;
  if (g >= G) return;
  float *o = out + (int64_t)b * C * G + g;
  float ss = 0.0f;
  for (int c = 0; c < C; ++c) ss = fmaf(o[(int64_t)c * G], o[(int64_t)c * G], ss);
  float nrm = sqrtf(ss);
  if (!(nrm >= eps)) nrm = eps;
  for (int c = 0; c < C; ++c) o[(int64_t)c * G] = o[(int64_t)c * G] / nrm;
}

__global__ void gather_labels_kernel(const int64_t *__restrict__ table, int M,
                                     const int64_t *__restrict__ img, const int64_t *__restrict__ seg,
                                     int64_t n, int64_t *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = table[img[i] * M + seg[i]];
}

// ---------------------------------------------------------------------------
// TransformerClustering.forward tail (hsg/models/embeddings/transformer_clusters.py:99-114):
//   logits[b,i,j] = (sum_c cent[b,c,i] * feat[b,c,j]) / sqrt(C)          (C1 chain over c)
//   order[b,:]    = the k rows i with the largest max_j logits[b,i,j], descending,
//                   lower i first on exact ties (NaN maxima first, like torch.topk)
//   outputs       = logits / cent / cfeat gathered in that order
// Two launches: (1) logits_all [B,tl,sl], one thread per output, lanes along j
// (coalesced node features, broadcast centroid element); (2) one workgroup per image:
// row maxima, rank by counting, gathers.  tl, sl are a few hundred at most: the op is
// launch bound in the reference (three einsum / max / topk / gather dispatches).
__global__ __launch_bounds__(256) void cluster_logits_kernel(
    const float *__restrict__ cent, const float *__restrict__ feat, int C, int tl, int sl,
    float divisor, float *__restrict__ logits_all) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= sl) return;
  const float *cb = cent + (int64_t)b * C * tl + i;
  const float *fb = feat + (int64_t)b * C * sl + j;
  float acc = 0.0f;
  for (int c = 0; c < C; ++c) acc = fmaf(cb[(int64_t)c * tl], fb[(int64_t)c * sl], acc);
  logits_all[((int64_t)b * tl + i) * sl + j] = acc / divisor;      // divisor = (float)sqrt((double)C), a true division like ATen on CPU
}

__global__ __launch_bounds__(256) void cluster_topk_kernel(
    const float *__restrict__ cent, const float *__restrict__ cfeat,
    const float *__restrict__ logits_all, int C, int tl, int sl, int k,
    int64_t *__restrict__ order, float *__restrict__ logits_sel, float *__restrict__ cent_sel,
    float *__restrict__ cfeat_sel) {
  extern __shared__ float smax[];              // [tl] row maxima, then int sel[k]
  int *sel = reinterpret_cast<int *>(smax + tl);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float *lg = logits_all + (int64_t)b * tl * sl;
  for (int i = w; i < tl; i += 4) {            // wave per row: max over j (NaN propagates, like torch.max)
    float m = -INFINITY;
    bool nan = false;
    for (int j = lane; j < sl; j += 64) {
      const float v = lg[(int64_t)i * sl + j];
      nan |= v != v;
      m = fmaxf(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
      m = fmaxf(m, __shfl_xor(m, off));
      nan |= __shfl_xor((int)nan, off) != 0;
    }
    if (lane == 0) smax[i] = nan ? NAN : m;
  }
  __syncthreads();
  for (int i = tid; i < tl; i += 256) {        // rank = number of rows that sort before row i
    const float v = smax[i];
    const bool vn = v != v;
    int rank = 0;
    for (int o = 0; o < tl; ++o) {
      const float u = smax[o];
      const bool un = u != u;
      const bool before = un ? (!vn || o < i) : (!vn && (u > v || (u == v && o < i)));
      rank += before ? 1 : 0;
    }
    if (rank < k) { sel[rank] = i; order[(int64_t)b * k + rank] = i; }
  }
  __syncthreads();
  for (int idx = tid; idx < k * sl; idx += 256) {
    const int r = idx / sl, j = idx - r * sl;
    logits_sel[((int64_t)b * k + r) * sl + j] = lg[(int64_t)sel[r] * sl + j];
  }
  for (int idx = tid; idx < C * k; idx += 256) {
    const int c = idx / k, r = idx - c * k;
    cent_sel[((int64_t)b * C + c) * k + r] = cent[((int64_t)b * C + c) * tl + sel[r]];
    cfeat_sel[((int64_t)b * C + c) * k + r] = cfeat[((int64_t)b * C + c) * tl + sel[r]];
  }
}


// ---------------------------------------------------------------------------------------------
// padded per-image prototype tables (resnet_fcn_hsg.py:499-577 / :1061-1136): the segments arrive sorted by
// image (the order of the tuple kernels of exchange.hip); a segment's table position is (dense image number,
// rank inside its image).  pad_scan: one workgroup walks the P image ids in blocks of 256 -- a max-scan gives
// every segment the first segment of its image, a sum-scan the dense image number -- and leaves the first
// segment of every image.  pad_fill: one workgroup per table slot copies the segment's rows (or zeros), writes
// mask / label / batch entries; the workgroups behind the slots map the pixels (segment -> rank, image).
__global__ __launch_bounds__(256) void pad_scan_kernel(const int64_t *__restrict__ seg_image, int P, int B, int M,
                                                       int32_t *__restrict__ seg_img, int32_t *__restrict__ seg_local,
                                                       int32_t *__restrict__ img_start, int64_t *__restrict__ seg_slot) {
  __shared__ int s_wmax[4], s_wsum[4], s_cstart, s_cimg;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { s_cstart = -1; s_cimg = 0; }
  __syncthreads();
  for (int i0 = 0; i0 < P; i0 += 256) {
    const int i = i0 + tid;
    const bool in = i < P;
    const bool first = in && (i == 0 || seg_image[i] != seg_image[i - 1]);
    int v = first ? i : -1, c = first ? 1 : 0;
    for (int off = 1; off < 64; off <<= 1) {
      const int ov = __shfl_up(v, off), oc = __shfl_up(c, off);
      if (lane >= off) { v = ov > v ? ov : v; c += oc; }
    }
    if (lane == 63) { s_wmax[wv] = v; s_wsum[wv] = c; }
    __syncthreads();
    int start = s_cstart, img = s_cimg;
    for (int q = 0; q < wv; ++q) { start = s_wmax[q] > start ? s_wmax[q] : start; img += s_wsum[q]; }
    start = v > start ? v : start;
    img += c;                                          // images up to and including this segment's
    if (in) {
      seg_img[i] = img - 1;
      seg_local[i] = i - start;
      if (seg_slot) seg_slot[i] = (int64_t)(img - 1) * M + (i - start);
      if (first && img - 1 < B) img_start[img - 1] = i;
    }
    __syncthreads();
    if (tid == 255) { s_cstart = start; s_cimg = img; }
    __syncthreads();
  }
  if (tid == 0) {
    const int nimg = s_cimg;
    for (int q = nimg; q <= B; ++q) img_start[q] = P;  // (images beyond the last: empty)
  }
}

__global__ __launch_bounds__(128) void pad_fill_kernel(
    const float *__restrict__ protos, int C, const float *__restrict__ pos, int Cp,
    const int64_t *__restrict__ seg_lab, const int64_t *__restrict__ seg_batch, const int64_t *__restrict__ pixel_seg,
    int64_t n, int slots, int M, const int32_t *__restrict__ seg_img, const int32_t *__restrict__ seg_local,
    const int32_t *__restrict__ img_start, float *__restrict__ table, float *__restrict__ pos_table,
    uint8_t *__restrict__ masks, int64_t *__restrict__ plabs, int64_t *__restrict__ pbatch,
    int64_t *__restrict__ by_image, int64_t *__restrict__ pixel_image) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < slots) {
    const int slot = blockIdx.x, img = slot / M, l = slot - img * M;
    const int s = img_start[img] + l;
    const bool used = s < img_start[img + 1];
    for (int c = tid; c < C; c += 128) table[(int64_t)slot * C + c] = used ? protos[(int64_t)s * C + c] : 0.0f;
    if (pos_table)
      for (int c = tid; c < Cp; c += 128) pos_table[(int64_t)slot * Cp + c] = used ? pos[(int64_t)s * Cp + c] : 0.0f;
    if (tid == 0) {
      masks[slot] = used ? 0 : 1;
      plabs[slot] = used ? seg_lab[s] : -1;
      pbatch[slot] = used ? seg_batch[s] : -1;
    }
    return;
  }
  const int64_t nb = gridDim.x - slots;
  for (int64_t i = ((int64_t)blockIdx.x - slots) * 128 + tid; i < n; i += nb * 128) {
    const int64_t sg = pixel_seg[i];
    by_image[i] = seg_local[sg];
    pixel_image[i] = seg_img[sg];
  }
}


// Backward of group_mean (+ the optional normalisation), one workgroup per image: the group sums are formed again
// in the forward's order (so that the mean and its norm are the forward's), the gradient of a group's mean vector
//   gm = normalised ? (norm >= eps ? (g - out <out, g>) / norm : g / eps) : g          (out = mean / max(norm, eps))
// is divided by the group's node count and handed to every unpadded node of the group.  Replaces the ATen
// restatement the mirror differentiated through autograd (~25 small kernels per call); a tolerance quantity.
__global__ __launch_bounds__(256) void group_mean_bwd_kernel(
    const float *__restrict__ protos, const int64_t *__restrict__ labels, const uint8_t *__restrict__ masks,
    int C, int N, int G, int normalized, float eps, const float *__restrict__ g_out, float *__restrict__ g_protos) {
  extern __shared__ float sm[];              // [G][C] sums -> gradients of the means, [G] counts, [G] norms, [G] dots, int labels [N]
  float *sums = sm;
  float *cnt = sm + (size_t)G * C;
  float *nrm = cnt + G;
  float *dot = nrm + G;
  int *lab = reinterpret_cast<int *>(dot + G);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int n = tid; n < N; n += 256) {
    const bool pad = masks ? masks[(int64_t)b * N + n] != 0 : false;
    const int64_t l = labels[(int64_t)b * N + n];
    lab[n] = (pad || l < 0 || l >= G) ? -1 : (int)l;
  }
  for (int i = tid; i < G * C; i += 256) sums[i] = 0.0f;
  __syncthreads();
  for (int g = tid; g < G; g += 256) {
    float k = 0.0f;
    for (int n = 0; n < N; ++n) k = k + (lab[n] == g ? 1.0f : 0.0f);
    cnt[g] = fmaxf(k, 1e-12f);
  }
  for (int c = tid; c < C; c += 256) {         // (a thread owns column c of every group: no conflicts)
    const float *p = protos + ((int64_t)b * C + c) * N;
    for (int n = 0; n < N; ++n) {
      const int g = lab[n];
      if (g >= 0) sums[g * C + c] = sums[g * C + c] + p[n];
    }
  }
  __syncthreads();
  // per group (one wave each, round-robin): the mean, its norm, <out, g>
  for (int g = w; g < G; g += 4) {
    float ss = 0.0f, dd = 0.0f;
    for (int c = lane; c < C; c += 64) {
      const float m = sums[g * C + c] / cnt[g];
      ss = fmaf(m, m, ss);
      dd = fmaf(m, g_out[((int64_t)b * C + c) * G + g], dd);
    }
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); dd += __shfl_xor(dd, off); }
    if (lane == 0) {
      float nn = sqrtf(ss);
      const bool clamped = !(nn >= eps);
      nrm[g] = clamped ? -eps : nn;            // (sign = the clamped branch, where out = mean / eps is linear)
      dot[g] = clamped ? 0.0f : dd / nn;       // <out, g> with out = mean / norm
    }
  }
  __syncthreads();
  for (int i = tid; i < G * C; i += 256) {
    const int g = i / C, c = i - g * C;
    const float go = g_out[((int64_t)b * C + c) * G + g];
    float gm = go;
    if (normalized) {
      const float nn = nrm[g];
      gm = nn < 0.0f ? go / (-nn) : (go - (sums[i] / cnt[g] / nn) * dot[g]) / nn;
    }
    sums[i] = gm / cnt[g];
  }
  __syncthreads();
  for (int64_t i = tid; i < (int64_t)C * N; i += 256) {
    const int c = (int)(i / N), n = (int)(i - (int64_t)c * N);
    const int g = lab[n];
    g_protos[(int64_t)b * C * N + i] = g >= 0 ? sums[g * C + c] : 0.0f;
  }
}

}  // namespace hsgk

using namespace hsgk;

extern "C" {

int hsgk_hier_assign(const float *fine_logits, const float *coarse_logits, int B, int KF, int KC,
                     int N, float *fine_prob, int64_t *fine_lab, float *coarse_prob,
                     int64_t *coarse_lab, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && KF >= 1 && N >= 1, "bad shape");
  HSGK_REQUIRE(fine_logits && fine_prob && fine_lab, "null argument");
  HSGK_REQUIRE(!coarse_logits || (KC >= 1 && coarse_prob && coarse_lab), "null coarse output");
  if (B == 0) return 0;
  (void)hipGetLastError();
  const size_t lds = (size_t)(coarse_logits ? KC : 0) * KF * 4 + 16;
  HSGK_REQUIRE(lds <= 64 * 1024, "hierarchy too large");
  hipLaunchKernelGGL(hier_assign_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream),
                     fine_logits, coarse_logits, KF, KC, N, fine_prob, fine_lab, coarse_prob, coarse_lab);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_hier_assign_bwd(const float *fine_logits, const float *coarse_logits, int B, int KF, int KC, int N,
                         const float *g_fine_prob, const float *g_coarse_prob, float *g_fine_logits,
                         float *g_coarse_logits, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && KF >= 1 && N >= 1, "bad shape");
  HSGK_REQUIRE(fine_logits && g_fine_logits, "null argument");
  const bool coarse = coarse_logits != nullptr && g_coarse_prob != nullptr;
  HSGK_REQUIRE(!coarse || (KC >= 1 && g_coarse_logits), "null coarse gradient");
  if (B == 0) return 0;
  (void)hipGetLastError();
  const size_t lds = (size_t)(coarse ? 5 * KC * KF : 0) * 4 + 16;
  HSGK_REQUIRE(lds <= 64 * 1024 && (!coarse || (int64_t)KC * KF <= 1024), "hierarchy too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (coarse_logits && !coarse && g_coarse_logits)
    HSGK_CHECK_HIP(hipMemsetAsync(g_coarse_logits, 0, sizeof(float) * (size_t)B * KC * KF, st));
  hipLaunchKernelGGL(hier_assign_bwd_kernel, dim3(B), dim3(256), lds, st, fine_logits, coarse ? coarse_logits : nullptr,
                     KF, KC, N, g_fine_prob, coarse ? g_coarse_prob : nullptr, g_fine_logits, g_coarse_logits);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_group_mean(const float *protos, const int64_t *labels, const uint8_t *masks, int B, int C,
                    int N, int G, int normalized, float eps, float *out, hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && C >= 1 && N >= 1 && G >= 1, "bad shape");
  if (B == 0) return 0;
  (void)hipGetLastError();
  const size_t lds = ((size_t)G * 64 + G) * 4 + (size_t)N * 4;
  HSGK_REQUIRE(lds <= 150 * 1024, "too many groups / nodes for one workgroup");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(group_mean_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(group_mean_kernel, dim3((C + 63) / 64, B), dim3(64), lds, st, protos, labels, masks, C, N, G, out);
  HSGK_LAUNCH_CHECK();
  if (normalized) {
    hipLaunchKernelGGL(group_normalize_kernel, dim3((G + 63) / 64, B), dim3(64), 0, st, out, C, G, eps);
    HSGK_LAUNCH_CHECK();
  }
  return 0;
}

int hsgk_group_mean_bwd(const float *protos, const int64_t *labels, const uint8_t *masks, int B, int C, int N,
                        int G, int normalized, float eps, const float *g_out, float *g_protos,
                        hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && C >= 1 && N >= 1 && G >= 1, "bad shape");
  if (B == 0) return 0;
  (void)hipGetLastError();
  const size_t lds = ((size_t)G * C + 3 * (size_t)G + (size_t)N) * 4;
  HSGK_REQUIRE(lds <= 150 * 1024, "too many groups x channels for one workgroup");
  HSGK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(group_mean_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(group_mean_bwd_kernel, dim3(B), dim3(256), lds, static_cast<hipStream_t>(stream), protos, labels,
                     masks, C, N, G, normalized, eps, g_out, g_protos);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_gather_labels(const int64_t *table, int M, const int64_t *img, const int64_t *seg, int64_t n,
                       int64_t *out, hsgk_stream_t stream) {
  HSGK_REQUIRE(n >= 0 && M >= 1, "bad shape");
  if (n == 0) return 0;
  (void)hipGetLastError();
  int64_t g = (n + 255) / 256;
  hipLaunchKernelGGL(gather_labels_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), table, M, img, seg, n, out);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_pad_prototype_tables(const int64_t *seg_image, int64_t P, const float *protos, int C, const float *pos,
                              int Cp, const int64_t *seg_lab, const int64_t *seg_batch, const int64_t *pixel_seg,
                              int64_t n, int B, int M, float *table, float *pos_table, uint8_t *masks,
                              int64_t *plabs, int64_t *pbatch, int64_t *by_image, int64_t *pixel_image,
                              int64_t *seg_slot, int32_t *work, hsgk_stream_t stream) {
  HSGK_REQUIRE(P >= 0 && P < (1ll << 31) && n >= 0 && B >= 0 && M >= 1 && C >= 1, "bad shape");
  HSGK_REQUIRE((int64_t)B * M < (1ll << 31), "table too large");
  HSGK_REQUIRE(P == 0 || (seg_image && protos && seg_lab && seg_batch && work), "null argument");
  HSGK_REQUIRE(B == 0 || (table && masks && plabs && pbatch), "null table");
  HSGK_REQUIRE(n == 0 || (pixel_seg && by_image && pixel_image), "null pixel vectors");
  HSGK_REQUIRE(!pos_table || (pos && Cp >= 1), "position rows required");
  (void)hipGetLastError();
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t *seg_img = work, *seg_local = work + P, *img_start = work + 2 * P;
  hipLaunchKernelGGL(pad_scan_kernel, dim3(1), dim3(256), 0, st, seg_image, (int)P, B, M, seg_img, seg_local, img_start,
                     seg_slot);
  HSGK_LAUNCH_CHECK();
  const int slots = B * M;
  int64_t pb = (n + 127) / 128;
  pb = pb > 2048 ? 2048 : pb;
  if (slots + pb == 0) return 0;
  hipLaunchKernelGGL(pad_fill_kernel, dim3((unsigned)(slots + pb)), dim3(128), 0, st, protos, C, pos, Cp, seg_lab,
                     seg_batch, pixel_seg, n, slots, M, seg_img, seg_local, img_start, table, pos_table, masks,
                     plabs, pbatch, by_image, pixel_image);
  HSGK_LAUNCH_CHECK();
  return 0;
}

int hsgk_cluster_topk(const float *centroids, const float *centroid_feats, const float *node_features,
                      int B, int C, int tl, int sl, int k, float *logits_all, int64_t *order,
                      float *logits_sel, float *centroids_sel, float *centroid_feats_sel,
                      hsgk_stream_t stream) {
  HSGK_REQUIRE(B >= 0 && C >= 1 && tl >= 1 && sl >= 1 && k >= 1 && k <= tl, "bad shape");
  HSGK_REQUIRE(centroids && centroid_feats && node_features && logits_all && order && logits_sel &&
                   centroids_sel && centroid_feats_sel, "null argument");
  HSGK_REQUIRE(tl <= 65535 && (size_t)(tl + k) * 4 <= 64 * 1024, "too many queries");
  if (B == 0) return 0;
  (void)hipGetLastError();
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cluster_logits_kernel, dim3((sl + 255) / 256, tl, B), dim3(256), 0, s, centroids,
                     node_features, C, tl, sl, (float)sqrt((double)C), logits_all);
  HSGK_LAUNCH_CHECK();
  hipLaunchKernelGGL(cluster_topk_kernel, dim3(B), dim3(256), (size_t)(tl + k) * 4, s, centroids,
                     centroid_feats, logits_all, C, tl, sl, k, order, logits_sel, centroids_sel,
                     centroid_feats_sel);
  HSGK_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
