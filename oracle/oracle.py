"""numpy/ctypes front-end of the CPU oracle (test infrastructure only).

Restates, on numpy arrays, the reference operators of the hot path
(/root/reference paths in every docstring).  Heavy loops live in
hsg_oracle.c (canonical summation order, see its header); the integer /
bookkeeping parts are plain numpy.

Parity status: PINNED by tests/golden/*.npz (generated from the reference by
tools/gen_golden.py, checked by tests/test_oracle_golden.py).

Nothing under hsg_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libhsg_oracle.so')

CHUNK = 2048      # canonical chunk length of order C2
EPS = np.float32(1e-12)

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_lib = None


def build():
  subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
  global _lib
  if _lib is None:
    if not os.path.exists(_SO):
      build()
    L = ctypes.CDLL(_SO)
    L.orc_prep.restype = ctypes.c_int64
    _lib = L
  return _lib


def _p(a, t):
  return a.ctypes.data_as(t) if a is not None else None


def _f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
  return np.ascontiguousarray(a, dtype=np.int64)


# --------------------------------------------------------------------------
def normalize_embedding(x, eps=EPS):
  """hsg/utils/general/common.py:101-120."""
  x = _f32(x)
  out = np.empty_like(x)
  d = x.shape[-1]
  lib().orc_normalize_rows(_p(x, _f32p), ctypes.c_int64(x.size // d), d,
                           ctypes.c_float(eps), _p(out, _f32p))
  return out


def linspace_f32(start, end, n):
  """torch.linspace(start, end, n) for float32 on the CPU, bit for bit, as ATen evaluates it
  (aten/src/ATen/native/cpu/RangeFactoriesKernel.cpp, linspace_kernel): step =
  (end - start) / (n - 1) in float32; element i is fma(step, i, start) in the lower half
  (i < n // 2) and fma(-step, n - 1 - i, end) in the upper half -- one rounding per element
  (the product of a float32 and a small integer is exact in float64, so is its sum with a
  float32 of similar magnitude).  Checked against torch.linspace for every grid size up to
  32 / image side up to 700 and for the location features in tests/test_cabi_and_host.py."""
  if n == 1:
    return np.full(1, np.float32(start))
  s0, e0 = np.float32(start), np.float32(end)
  step = np.float32((e0 - s0) / np.float32(n - 1))
  i = np.arange(n)
  lo = (np.float64(s0) + np.float64(step) * i).astype(np.float32)
  hi = (np.float64(e0) - np.float64(step) * (n - 1 - i)).astype(np.float32)
  return np.where(i < n // 2, lo, hi).astype(np.float32)


def grid_seed_axis(k, n):
  """linspace(0, k-1, n).round_().long() (hsg/utils/segsort/common.py:145-148): the float32
  linspace of ATen (linspace_f32), rounded half to even.  The float32 rounding is part of the
  reference's behaviour: the exactly rounded i*(k-1)/(n-1) differs from it for about 2 % of the
  (k, n) pairs (e.g. k = 4, n = 43), which tests/checkers/fuzz_parity.py found."""
  return np.rint(linspace_f32(0.0, float(k - 1), n)).astype(np.int64)


def initialize_cluster_labels(num_clusters, img_dimensions):
  """hsg/utils/segsort/common.py:129-153: y + (y.max()+1) * x."""
  y = grid_seed_axis(num_clusters[0], img_dimensions[0])
  x = grid_seed_axis(num_clusters[1], img_dimensions[1])
  return y[:, None] + (y.max() + 1) * x[None, :]


def generate_location_features(img_dimensions):
  """hsg/utils/segsort/common.py:156-189, feature_type='float': [H, W, 2] (y, x) in [0, 1]."""
  y = linspace_f32(0.0, 1.0, int(img_dimensions[0]))
  x = linspace_f32(0.0, 1.0, int(img_dimensions[1]))
  return np.stack(np.meshgrid(y, x, indexing='ij'), axis=2).astype(np.float32)


def dense_relabel(v):
  """torch.unique(v, return_inverse=True)[1]: rank among sorted unique."""
  return np.unique(v, return_inverse=True)[1].astype(np.int64).reshape(v.shape)


def prepare_prototype_labels(semantic_labels, instance_labels, offset=256):
  """hsg/utils/segsort/common.py:192-218."""
  pan = _i64(semantic_labels) + _i64(instance_labels) * int(offset)
  uniq, inv = np.unique(pan, return_inverse=True)
  return uniq % int(offset), inv.astype(np.int64)


def calculate_prototypes_from_labels(embeddings, labels, max_label=None,
                                     chunk=CHUNK, exact_sums=False):
  """hsg/utils/segsort/common.py:11-41 with summation order C2 (exact_sums: C2x, the
  M-step arithmetic of the Lloyd loop in segment_by_kmeans)."""
  x = _f32(embeddings).reshape(-1, embeddings.shape[-1])
  lab = _i64(labels).reshape(-1)
  P = int(lab.max()) + 1 if max_label is None else int(max_label)
  out = np.empty((P, x.shape[1]), np.float32)
  if exact_sums:
    lib().orc_prototypes_exact(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1],
                               _p(lab, _i64p), ctypes.c_int64(P), ctypes.c_float(EPS), _p(out, _f32p))
    return out
  lib().orc_prototypes(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1],
                       _p(lab, _i64p), ctypes.c_int64(P), chunk,
                       ctypes.c_float(EPS), _p(out, _f32p))
  return out



def exchange_prototypes(parts):
  """hsg/models/utils.py:127-217 gather_clustering_and_update_prototypes for a list of per-source
  pixel sets (dicts with emb [n,C], emb_loc [n,D], cluster / batch / sem / inst [n]): the reference
  concatenates the sources, makes (batch, cluster) dense, then `prepare_prototype_labels` on
  (batch, sem, inst) x cluster -- i.e. dense id = rank of (batch, cluster, sem, inst) among the distinct
  tuples in lexicographic order (:181-193) -- and normalises the per-id sums (:199-202).  Sums are
  taken per SOURCE in order C2 and added in source order from +0.0f (the all-reduce of the product;
  a segment normally lives on one source, so this only adds zeros).
  Returns protos, protos_loc, psem, pinst, pbatch, [updated ids per source]."""
  tup = [np.stack([_i64(p['batch']).reshape(-1), _i64(p['cluster']).reshape(-1),
                   _i64(p['sem']).reshape(-1), _i64(p['inst']).reshape(-1)], 1) for p in parts]
  allt = np.concatenate(tup, 0)
  uniq, inv = np.unique(allt, axis=0, return_inverse=True)
  inv = inv.reshape(-1).astype(np.int64)
  P = uniq.shape[0]
  C, D = parts[0]['emb'].shape[1], parts[0]['emb_loc'].shape[1]
  ta, tb = np.zeros((P, C), np.float32), np.zeros((P, D), np.float32)
  upd, o = [], 0
  for p, t in zip(parts, tup):
    ids = inv[o:o + t.shape[0]]
    o += t.shape[0]
    upd.append(ids)
    for x, tab in ((p['emb'], ta), (p['emb_loc'], tb)):
      x = _f32(x)
      part = np.zeros((P, x.shape[1]), np.float32)
      if x.shape[0]:
        lib().orc_segment_sums(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1], _p(ids, _i64p),
                               ctypes.c_int64(P), CHUNK, _p(part, _f32p))
      tab += part
  return (normalize_embedding(ta), normalize_embedding(tb), uniq[:, 2].copy(), uniq[:, 3].copy(),
          uniq[:, 0].copy(), upd)


def segment_mean(x, index, chunk=CHUNK):
  """hsg/utils/general/common.py:123-147."""
  x = _f32(x).reshape(-1, x.shape[-1])
  idx = _i64(index).reshape(-1)
  P = int(idx.max()) + 1
  out = np.empty((P, x.shape[1]), np.float32)
  lib().orc_segment_mean(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1],
                         _p(idx, _i64p), ctypes.c_int64(P), chunk,
                         _p(out, _f32p))
  return out


def find_nearest_prototypes(embeddings, prototypes, return_best=False):
  """hsg/utils/segsort/common.py:44-64."""
  c = _f32(prototypes)
  x = _f32(embeddings).reshape(-1, c.shape[-1])
  out = np.empty(x.shape[0], np.int32)
  best = np.empty(x.shape[0], np.float32) if return_best else None
  lib().orc_assign(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1],
                   _p(c, _f32p), c.shape[0], _p(out, _i32p), _p(best, _f32p))
  return (out.astype(np.int64), best) if return_best else out.astype(np.int64)


def kmeans_with_initial_labels(embeddings, initial_labels, max_label=None,
                               iterations=10, chunk=CHUNK, return_centroids=False,
                               exact_sums=False):
  """hsg/utils/segsort/common.py:67-97.  exact_sums: M-step with exact fixed-point segment
  sums (canonical order C2x, unit-norm rows -- what segment_by_kmeans uses), else the
  chunked fp32 order C2."""
  x = _f32(embeddings)
  init = np.ascontiguousarray(initial_labels, dtype=np.int32)
  K = int(init.max()) + 1 if max_label is None else int(max_label)
  out = np.empty(x.shape[0], np.int32)
  cent = np.empty((K, x.shape[1]), np.float32)
  lib().orc_kmeans_ex(_p(x, _f32p), ctypes.c_int64(x.shape[0]), x.shape[1],
                      _p(init, _i32p), K, int(iterations), chunk,
                      ctypes.c_float(EPS), int(bool(exact_sums)), _p(out, _i32p), _p(cent, _f32p))
  out = out.astype(np.int64)
  return (out, cent) if return_centroids else out


def segment_by_kmeans(embeddings, labels=None, num_clusters=(5, 5),
                      local_features=None, ignore_index=None, iterations=10,
                      gpu_id=0, chunk=CHUNK, cluster_indices=None):
  """hsg/utils/segsort/common.py:270-408.  cluster_indices [B,H,W]: the caller's initial labels
  (:320-323), made dense per image (:341-345); default: the grid seeds of num_clusters.

  embeddings [B,C,H,W] f32; labels [B,H,W] i64 or None; local_features
  [H,W,2] or [B,H,W,2] f32 (REQUIRED here: the float32 bit patterns of
  torch.linspace are input data).  Returns the reference's
  5-tuple as numpy arrays.
  """
  x = _f32(embeddings)
  B, C, H, W = x.shape
  D = C + 2
  loc = _f32(local_features)
  loc_sb = 0 if loc.ndim == 3 else H * W * 2
  if labels is None and ignore_index is not None:
    labels = np.zeros((B, H, W), np.int64)                 # common.py:326-329
  lab = _i64(labels) if labels is not None else None
  n_max = B * H * W
  emb = np.empty((n_max, C), np.float32)
  emb_loc = np.empty((n_max, D), np.float32)
  lab_out = np.empty(n_max, np.int64)
  counts = np.empty(B, np.int64)
  n = lib().orc_prep(_p(x, _f32p), B, C, H, W, _p(loc, _f32p),
                     ctypes.c_int64(loc_sb), _p(lab, _i64p),
                     int(ignore_index is not None),
                     ctypes.c_int64(0 if ignore_index is None else int(ignore_index)),
                     ctypes.c_float(EPS), _p(emb, _f32p), _p(emb_loc, _f32p),
                     _p(lab_out, _i64p), _p(counts, _i64p))
  emb, emb_loc, lab_out = emb[:n], emb_loc[:n], lab_out[:n]

  # common.py:320-323, 341-345: grid seeds, densely relabelled per image
  seeds = dense_relabel(initialize_cluster_labels(num_clusters, (H, W)).reshape(-1))
  K = int(seeds.max()) + 1
  cluster = np.empty(n, np.int64)
  batch = np.empty(n, np.int64)
  off = 0
  for b in range(B):
    cnt = int(counts[b])
    if cluster_indices is not None:
      seeds = dense_relabel(_i64(cluster_indices)[b].reshape(-1))
      K = int(seeds.max()) + 1
    if lab is not None and ignore_index is not None:
      keep = lab[b].reshape(-1) != int(ignore_index)
      init = seeds[keep]
    else:
      init = seeds
    if cnt > 0:                                             # common.py:368
      cluster[off:off + cnt] = kmeans_with_initial_labels(
          emb_loc[off:off + cnt], init, K, iterations, chunk, exact_sums=True)
    batch[off:off + cnt] = b + B * gpu_id                  # common.py:375-381
    off += cnt

  # common.py:398-405
  if n > 0:
    lab_div = int(cluster.max()) + 1
    cluster = dense_relabel(batch * lab_div + cluster)
    _, cluster = prepare_prototype_labels(lab_out, cluster, int(lab_out.max()) + 1)
  return emb, emb_loc, lab_out, cluster, batch


def segsort_nll(embeddings, semantic_labels, instance_labels, prototypes,
                prototype_semantic_labels, concentration, group_mode='segsort+',
                want_grads=False, gscale=None):
  """hsg/utils/segsort/loss.py:15-82 per-pixel negative log-likelihood (f64)."""
  e = _f32(embeddings).reshape(-1, embeddings.shape[-1])
  p = _f32(prototypes).reshape(-1, prototypes.shape[-1])
  sem = _i64(semantic_labels).reshape(-1)
  inst = _i64(instance_labels).reshape(-1)
  psem = _i64(prototype_semantic_labels).reshape(-1)
  n, c = e.shape
  nll = np.empty(n, np.float64)
  ge = np.zeros((n, c), np.float64) if want_grads else None
  gp = np.zeros((p.shape[0], c), np.float64) if want_grads else None
  if gscale is None:
    gscale = 1.0 / max(n, 1)
  lib().orc_segsort_nll(_p(e, _f32p), ctypes.c_int64(n), c, _p(sem, _i64p),
                        _p(inst, _i64p), _p(p, _f32p), ctypes.c_int64(p.shape[0]),
                        _p(psem, _i64p), ctypes.c_float(concentration),
                        int(group_mode == 'segsort+'), _p(nll, _f64p),
                        ctypes.c_double(gscale), _p(ge, _f64p), _p(gp, _f64p))
  return (nll, ge, gp) if want_grads else nll


def set_segsort_nll(embeddings, semantic_labels, instance_labels, prototypes,
                    prototype_semantic_labels, concentration, group_mode='segsort+',
                    want_grads=False, gscale=None):
  """hsg/utils/segsort/loss.py:85-130 (SetSegSortLoss): multi-hot labels [n,nc] / [P,nc],
  same / different by the label affinity sem @ psem.T (f64)."""
  e = _f32(embeddings).reshape(-1, embeddings.shape[-1])
  p = _f32(prototypes).reshape(-1, prototypes.shape[-1])
  sem = _i64(semantic_labels)
  psem = _i64(prototype_semantic_labels)
  nc = sem.shape[-1]
  sem, psem = sem.reshape(-1, nc), psem.reshape(-1, nc)
  inst = _i64(instance_labels).reshape(-1)
  n, c = e.shape
  nll = np.empty(n, np.float64)
  ge = np.zeros((n, c), np.float64) if want_grads else None
  gp = np.zeros((p.shape[0], c), np.float64) if want_grads else None
  if gscale is None:
    gscale = 1.0 / max(n, 1)
  lib().orc_set_segsort_nll(_p(e, _f32p), ctypes.c_int64(n), c, nc, _p(sem, _i64p),
                            _p(inst, _i64p), _p(p, _f32p), ctypes.c_int64(p.shape[0]),
                            _p(psem, _i64p), ctypes.c_float(concentration),
                            int(group_mode == 'segsort+'), _p(nll, _f64p),
                            ctypes.c_double(gscale), _p(ge, _f64p), _p(gp, _f64p))
  return (nll, ge, gp) if want_grads else nll


def segsort_loss(*args, reduction='mean', **kw):
  """hsg/utils/segsort/loss.py:149-190 SegSortLoss.forward."""
  nll = segsort_nll(*args, **kw)
  if reduction == 'mean':
    return float(nll.mean())
  if reduction == 'sum':
    return float(nll.sum())
  return nll


# ---- hierarchical grouping (hsg/models/embeddings/resnet_fcn_hsg.py) ----------
def _softmax_dim1(x):
  """float32 softmax over axis 1: exp(x - max) / sequential sum (canonical)."""
  x = x.astype(np.float32)
  e = np.exp(x - x.max(axis=1, keepdims=True), dtype=np.float32)
  s = np.zeros_like(e[:, 0])
  for c in range(e.shape[1]):
    s = (s + e[:, c]).astype(np.float32)
  return (e / s[:, None]).astype(np.float32)


def hierarchical_grouping_from_logits(fine_logits, coarse_logits=None):
  """resnet_fcn_hsg.py:638-672: (fine_labels, fine_probs, coarse_labels, coarse_probs)."""
  pf = _softmax_dim1(np.asarray(fine_logits))
  flab = pf.argmax(axis=1).astype(np.int64)
  if coarse_logits is None:
    return flab, pf, None, None
  pc = _softmax_dim1(np.asarray(coarse_logits))                    # [B,KC,KF]
  B, KC, KF = pc.shape
  out = np.zeros((B, KC, pf.shape[2]), np.float32)
  for c in range(KF):                                              # fmaf chain over fine clusters
    out = (pc[:, :, c, None].astype(np.float64) * pf[:, None, c, :].astype(np.float64)
           + out.astype(np.float64)).astype(np.float32)
  return flab, pf, out.argmax(axis=1).astype(np.int64), out


def collect_nd_coarser_prototype(prototypes, labels, masks=None, num_groups=None, normalized=True):
  """resnet_fcn_hsg.py:683-748: [B,C,N] -> [B,C,G] masked group means."""
  p = _f32(prototypes)
  B, C, N = p.shape
  lab = _i64(labels)
  G = int(lab.max()) + 1 if num_groups is None else int(num_groups)
  out = np.zeros((B, C, G), np.float32)
  for b in range(B):
    for g in range(G):
      sel = lab[b] == g
      if masks is not None:
        sel &= ~np.asarray(masks[b], bool)
      s = np.zeros(C, np.float32)
      for n in np.nonzero(sel)[0]:
        s = (s + p[b, :, n]).astype(np.float32)
      out[b, :, g] = s / np.float32(max(float(sel.sum()), 1e-12))
    if normalized:
      out[b] = normalize_embedding(out[b].T.copy()).T
  return out


def collect_pixel_hierarchical_clustering_indices(cluster_indices_by_batch, cluster_batch_indices,
                                                  grouping_labels):
  """resnet_fcn_hsg.py:751-780."""
  _, img = np.unique(_i64(cluster_batch_indices), return_inverse=True)
  return _i64(grouping_labels)[img, _i64(cluster_indices_by_batch)]


def calculate_kmeans_prototypes(emb, cluster_indices, cluster_batch_indices, pos_emb, labels,
                                label_divisor=256, max_num_clusters=256):
  """resnet_fcn_hsg.py:455-577, per-image loop as in the reference."""
  emb = _f32(emb)
  c, b, lab = _i64(cluster_indices), _i64(cluster_batch_indices), _i64(labels)
  M = int(max_num_clusters)
  outs = ([], [], [], [], [], [])
  for bi in np.unique(b):
    sel = np.nonzero(b == bi)[0]
    cl = b[sel] * label_divisor ** 2 + lab[sel]
    plab, ci = prepare_prototype_labels(cl, c[sel], int(cl.max()) + 1)
    n = plab.shape[0]
    outs[0].append(calculate_prototypes_from_labels(emb[sel], ci, M))
    if pos_emb is not None:
      pm = np.zeros((M, pos_emb.shape[1]), np.float32)
      pm[:n] = segment_mean(_f32(pos_emb)[sel], ci)
      outs[1].append(pm)
    mk = np.ones(M, bool)
    mk[:n] = False
    outs[2].append(mk)
    pl = np.full(M, -1, np.int64)
    pl[:n] = plab % label_divisor ** 2
    outs[3].append(pl)
    pb = np.full(M, -1, np.int64)
    pb[:n] = plab // label_divisor ** 2
    outs[4].append(pb)
    outs[5].append(ci)
  protos = np.stack(outs[0]).transpose(0, 2, 1)
  pos = np.stack(outs[1]).transpose(0, 2, 1) if outs[1] else None
  return (protos, pos, np.stack(outs[2]), np.stack(outs[3]), np.stack(outs[4]),
          np.concatenate(outs[5]))


def transformer_clustering_tail(centroids, centroid_feats, node_features, k):
  """hsg/models/embeddings/transformer_clusters.py:99-114: logits = cent^T feat / sqrt(C)
  (C1 chain over channels), the k rows with the largest row maximum (descending, lower
  index first on ties), gathers."""
  cen, cfe, nod = _f32(centroids), _f32(centroid_feats), _f32(node_features)
  B, C, tl = cen.shape
  sl = nod.shape[-1]
  logits_all = np.empty((B, tl, sl), np.float32)
  lib().orc_cluster_logits(_p(cen, _f32p), _p(nod, _f32p), B, C, tl, sl, _p(logits_all, _f32p))
  mx = logits_all.max(axis=2)
  order = np.stack([np.lexsort((np.arange(tl), -mx[b]))[:k] for b in range(B)]).astype(np.int64)
  logits = np.stack([logits_all[b, order[b]] for b in range(B)])
  cen_sel = np.stack([cen[b][:, order[b]] for b in range(B)])
  cfe_sel = np.stack([cfe[b][:, order[b]] for b in range(B)])
  return cen_sel, cfe_sel, logits, order


def find_majority_label_index(semantic_labels, cluster_labels):
  """hsg/utils/segsort/common.py:221-268."""
  sem = _i64(semantic_labels).reshape(-1)
  clu = _i64(cluster_labels).reshape(-1)
  nk, nc = int(clu.max()) + 1, int(sem.max()) + 1
  hist = np.zeros((nk, nc), np.int64)
  np.add.at(hist, (clu, sem), 1)
  majority = hist.argmax(axis=1).astype(np.int64)              # first maximum
  select = np.nonzero(majority[clu] == sem)[0].reshape(-1, 1).astype(np.int64)
  return select, majority


def overlap_average(crops, corners, channels, height, width, eps=EPS):
  """pyscripts/inference/prototype.py:141-177: canvas += normalize(crop) per crop (in the
  given order), counts += 1, canvas /= counts.  crops: list of [C,h,w] float32."""
  canvas = np.zeros((channels, height, width), np.float32)
  counts = np.zeros((height, width), np.float32)
  for crop, (sh, sw) in zip(crops, corners):
    c = _f32(crop)
    C, h, w = c.shape
    rows = normalize_embedding(np.ascontiguousarray(c.reshape(C, h * w).T), eps)   # [h*w, C], C1 chain
    canvas[:, sh:sh + h, sw:sw + w] = canvas[:, sh:sh + h, sw:sw + w] + rows.T.reshape(C, h, w)
    counts[sh:sh + h, sw:sw + w] += np.float32(1)
  return canvas / counts[None]


def affinity_matrix_as_attention(x, x_padding_mask=None, x_segment_labels=None, knn=None,
                                 remove_self_loop=True, binarize=True, concentration=5.0,
                                 affinity=None):
  """hsg/utils/graph/common.py:39-125, statement by statement (the per-image /
  per-segment loop with the k-th largest value taken from a sorted copy)."""
  x = _f32(x)
  B, C, N = x.shape
  if affinity is None:
    A = np.empty((B, N, N), np.float32)
    lib().orc_exp_affinity(_p(x, _f32p), B, C, N, ctypes.c_float(concentration), _p(A, _f32p))
  else:
    A = _f32(affinity).copy()
  pad = np.zeros((B, N), bool) if x_padding_mask is None else np.asarray(x_padding_mask, bool)
  seg = np.zeros((B, N), np.int64) if x_segment_labels is None else _i64(x_segment_labels)
  A[pad[:, :, None] | pad[:, None, :]] = 0                                   # :81-82
  if remove_self_loop:                                                        # :85-93
    for b in range(B):
      if (~pad[b]).sum() > 1:
        A[b][np.eye(N, dtype=bool)] = 0
  if knn is not None:                                                         # :96-118
    for b in range(B):
      cur = A[b]
      for lab in np.unique(seg[b]):
        m = (~pad[b]) & (seg[b] == lab)
        if not m.any():
          continue
        k = min(int(m.sum()), knn)
        adj = cur[:, m]
        kth = -np.sort(-adj, axis=1)[:, k - 1]
        cut = m[None, :] & (cur < kth[:, None])
        cur[cut] = 0
  if binarize:                                                                # :121-122
    A = np.where(A > 0, np.float32(1), np.float32(0)).astype(np.float32)
  return A
