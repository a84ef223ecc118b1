"""Drop-in for `hsg.utils.segsort.common` on MI355X.

Mirrors the reference module's call surface (hsg/utils/segsort/common.py):
`segment_by_kmeans`, `kmeans_with_initial_labels`, `find_nearest_prototypes`,
`calculate_prototypes_from_labels`, `prepare_prototype_labels`,
`initialize_cluster_labels`, `generate_location_features`.  The arithmetic is
done by libhsgk (hand-written gfx950 kernels, C ABI in include/hsgk.h); this
file only validates arguments, allocates outputs/workspace on the caller's
device and stream, and shapes the results like the reference does.

Thread-safe: no global mutable state apart from two read-only caches of tiny
per-(H,W) tables; every call allocates its own workspace and launches on the
calling thread's current device/stream (the reference's DataParallel shell
calls these from one Python thread per GPU).
"""
import ctypes
import threading

import torch

from hsg_amd import _lib, _torch_ops, ops
from hsg_amd.utils.general import common as common_utils

_cache_lock = threading.Lock()
_seed_cache = {}
_loc_cache = {}
_TABLE_CAP_MAX = 1 << 24


def _require_gpu(t, name):
  if not t.is_cuda:
    raise _lib.HsgkError('%s must be a ROCm device tensor (got %s); hsg_amd has '
                         'no CPU path' % (name, t.device))


def initialize_cluster_labels(num_clusters, img_dimensions, device):
  """Uniform grid seed labels (reference common.py:129-153).

  The float32 `linspace(...).round_()` is evaluated by torch on the host --
  its bit pattern is part of the reference's behaviour -- and the tiny [H,W]
  result is moved to `device`.
  """
  y = torch.linspace(0, num_clusters[0] - 1, img_dimensions[0]).round_().long()
  x = torch.linspace(0, num_clusters[1] - 1, img_dimensions[1]).round_().long()
  labels = y.view(-1, 1) + (y.max() + 1) * x.view(1, -1)
  return labels.to(device)


_locfeat_cache = {}


def generate_location_features(img_dimensions, device, feature_type='int'):
  """[H,W,2] (y,x) location features (reference common.py:156-189).  The table is a function of its arguments: a
  device result is built once per (shape, device, type) -- `linspace` on the host, as the reference's bits are ATen's
  CPU `linspace` bits, then one upload -- and every call returns a fresh COPY of it (callers shift it in place:
  `local_features -= 0.5`, common.py:316): one device copy per call instead of host work and an upload."""
  dev = torch.device(device)
  if dev.type == 'cuda' and feature_type in ('int', 'float'):
    if dev.index is None:
      dev = torch.device('cuda', torch.cuda.current_device())
    key = (int(img_dimensions[0]), int(img_dimensions[1]), str(dev), feature_type)
    with _cache_lock:
      hit = _locfeat_cache.get(key)
    if hit is None:
      hit = generate_location_features(img_dimensions, 'cpu', feature_type).to(dev)
      with _cache_lock:
        _locfeat_cache[key] = hit
    return hit.clone()
  if feature_type == 'int':
    y = torch.arange(img_dimensions[0])
    x = torch.arange(img_dimensions[1])
  elif feature_type == 'float':
    y = torch.linspace(0, 1, img_dimensions[0])
    x = torch.linspace(0, 1, img_dimensions[1])
  else:
    raise ValueError('Type of location features should be either int or float.')
  yy, xx = torch.meshgrid(y, x, indexing='ij')
  return torch.stack([yy, xx], dim=2).to(device)


def _seed_map(num_clusters, H, W, device):
  """Dense int32 seed label per pixel + cluster count (common.py:320-345)."""
  key = (int(num_clusters[0]), int(num_clusters[1]), H, W, str(device))
  with _cache_lock:
    hit = _seed_cache.get(key)
  if hit is None:
    grid = initialize_cluster_labels(num_clusters, (H, W), 'cpu').view(-1)
    _, dense = torch.unique(grid, return_inverse=True)
    hit = (dense.to(torch.int32).to(device), int(dense.max()) + 1)
    with _cache_lock:
      _seed_cache[key] = hit
  return hit


def _explicit_seeds(cluster_indices, B, H, W, device):
  """Initial labels given by the caller (reference common.py:320-323, 341-345): every image's
  map is made dense separately (`torch.unique(..., return_inverse=True)` per image) and the
  number of clusters of an image is its number of distinct initial labels.  libhsgk runs a
  batch with ONE cluster count (an image with fewer would gain empty clusters whose zero centroids
  compete in the argmax, which the reference's smaller table does not have): maps whose images
  differ in that number come back with the list of counts and are run group by group."""
  ci = cluster_indices.detach().to(torch.int64)
  if ci.dim() == 2:
    ci = ci.unsqueeze(0)
  if tuple(ci.shape[-2:]) != (H, W) or ci.shape[0] not in (1, B):
    raise ValueError('cluster_indices must be [batch, height, width]')
  ci = ci.to(device).expand(B, H, W).reshape(B, H * W)
  lo = ci.min()
  span = ci.max() - lo + 1
  if int(span) * B >= (1 << 62):
    # (batch * span + value would overflow int64 and merge seeds across images: rank-compress the values first)
    ci = torch.unique(ci, return_inverse=True)[1].view(B, H * W)
    lo = ci.min()
    span = ci.max() - lo + 1
  keys = (torch.arange(B, device=device).view(B, 1) * span + (ci - lo)).view(-1)
  uniq, inv = torch.unique(keys, return_inverse=True)
  first = torch.searchsorted(uniq, torch.arange(B, device=device) * span)          # first dense id of every image
  counts = torch.diff(torch.cat([first, torch.tensor([uniq.shape[0]], device=device)]))
  counts = counts.cpu().tolist()
  dense = (inv.view(B, H * W) - first.view(B, 1)).to(torch.int32).contiguous()
  if len(set(counts)) != 1:
    return dense, [int(c) for c in counts], H * W          # per-image counts: _segment_by_kmeans_grouped
  return dense, int(counts[0]), H * W


def _default_loc(H, W, device):
  key = (H, W, str(device))
  with _cache_lock:
    hit = _loc_cache.get(key)
  if hit is None:
    loc = generate_location_features((H, W), 'cpu', 'float')
    loc -= 0.5
    hit = loc.contiguous().to(device)
    with _cache_lock:
      _loc_cache[key] = hit
  return hit


def _batch_offset(B, dev):
  """First batch index of this call's images (reference common.py:375-377: `N * gpu_id`,
  one process driving every GPU).  With one process per GPU every rank may see its GPU as
  device 0, so under torch.distributed the offset follows the global rank instead."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized():
    return B * dist.get_rank()
  return B * (dev.index or 0)


class _SegmentByKmeans(torch.autograd.Function):
  """One libhsgk call; `embeddings` / `embeddings_with_loc` are differentiable
  w.r.t. the NCHW input (normalise -> concat -> normalise -> index_select),
  the three index outputs are not."""

  @staticmethod
  def forward(ctx, x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, iterations, batch_offset, seed_sb):
    dev = x.device
    B, C, H, W = x.shape
    n_max = B * H * W
    want_grad = x.requires_grad
    xd = x.detach()
    table_cap = B * K if lab is None else max(B * K, min(B * K * 4096, _TABLE_CAP_MAX))
    L = _lib.lib()

    def run(lab, ign, table_cap, flags=0):
      with torch.cuda.device(dev):
        out_emb = torch.empty((n_max, C), dtype=torch.float32, device=dev)
        out_loc = torch.empty((n_max, C + 2), dtype=torch.float32, device=dev)
        out_lab = torch.empty((n_max,), dtype=torch.int64, device=dev)
        out_cluster = torch.empty((n_max,), dtype=torch.int64, device=dev)
        out_batch = torch.empty((n_max,), dtype=torch.int64, device=dev)
        meta = torch.empty((8,), dtype=torch.int64, device=dev)
        norms = torch.empty((n_max, 2), dtype=torch.float32, device=dev) if want_grad else None
        rowmap = (torch.empty((n_max,), dtype=torch.int64, device=dev)
                  if want_grad and has_ignore else None)
        ws_bytes = L.hsgk_segment_by_kmeans_workspace_bytes(B, C, H, W, K, table_cap)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        args = _lib.SegkmArgs(
            embeddings=xd.data_ptr(), labels=lab.data_ptr() if lab is not None else None,
            loc=loc.data_ptr(), loc_batch_stride=loc_sb, seed_map=seed_map.data_ptr(),
            B=B, C=C, H=H, W=W, K=K, iterations=int(iterations), has_ignore=int(has_ignore),
            ignore_index=ign, batch_offset=batch_offset, table_cap=table_cap,
            out_embeddings=out_emb.data_ptr(), out_embeddings_loc=out_loc.data_ptr(),
            out_labels=out_lab.data_ptr(), out_cluster=out_cluster.data_ptr(),
            out_batch=out_batch.data_ptr(), meta=meta.data_ptr(),
            out_norms=norms.data_ptr() if norms is not None else None,
            out_rowmap=rowmap.data_ptr() if rowmap is not None else None,
            workspace=ws.data_ptr(), workspace_bytes=ws_bytes, seed_batch_stride=seed_sb, flags=flags)
        _lib.check(L.hsgk_segment_by_kmeans(ctypes.byref(args), _lib.stream_ptr()))
        if lab is None:
          # no label map: every pixel is kept and neither data-dependent error can occur, so the
          # row count is known on the host and the call stays asynchronous (no host sync at all)
          m = [n_max, 0, 0, 0, 0, 0, 0, 0]
          if flags == 0 and L.hsgk_small_map_groups(B, C, H, W, K) > 1:
            # several co-operating workgroups per image: a wait that timed out (error 3) is the one
            # data-independent failure of this path -- it travels to pinned host memory behind the
            # kernels and is raised by a later libhsgk call (or at once with HSGK_SYNC_ERRORS=1)
            _lib.defer_status(meta.view(torch.int32)[10:11],
                              'segment_by_kmeans: the co-operating workgroups of an image waited 10 s for each '
                              'other (labels invalid); set HSGK_SMALL_GROUPS=1')
        else:
          m = ops.read_i64(meta)         # the operator's single host sync
      return m, (out_emb, out_loc, out_lab, out_cluster, out_batch, norms, rowmap)

    m, outs = run(lab, ign, table_cap)
    if m[5] == 3:
      # the co-operating workgroups of an image timed out waiting for each other (another stream kept the
      # CUs for > 10 s): repeat with ONE workgroup per image, which waits for nobody
      m, outs = run(lab, ign, table_cap, flags=1)
    if m[5] == 2:
      # The label values (or their number) do not fit the library's presence table.  The final
      # ids only depend on the ORDER of the (image, cluster, label) triples (reference
      # common.py:398-405: two sorted `unique`s), so the call is repeated on the ranks of the
      # distinct label values -- a monotone map -- with a table sized for them, and the label
      # output is mapped back.  (Rare: label values >= 2^24 or thousands of distinct values.)
      kept = lab[lab != ign] if has_ignore else lab
      if kept.numel() and int(kept.min()) < 0:
        # (the rank compression below would hide negative labels from the library's own check)
        raise ValueError('segment_by_kmeans: negative labels are not supported')
      uniq, inv = torch.unique(lab, return_inverse=True)
      D = int(uniq.numel())
      ign_rank = D
      if has_ignore:
        pos = int(torch.searchsorted(uniq, torch.tensor([ign], dtype=uniq.dtype, device=dev)))
        if pos < D and int(uniq[pos]) == ign:
          ign_rank = pos
      m, outs = run(inv.view(B, H, W).contiguous(), ign_rank, max(table_cap, B * K * (D + 1)))
      if m[5] == 0:
        outs = outs[:2] + (uniq[outs[2].clamp_(0, D - 1)],) + outs[3:]
    out_emb, out_loc, out_lab, out_cluster, out_batch, norms, rowmap = outs
    n, err = m[0], m[5]
    if err == 1:
      raise ValueError('segment_by_kmeans: negative labels are not supported')
    if err == 3:
      raise _lib.HsgkError('segment_by_kmeans: the cooperating workgroups of an image were not co-resident '
                           '(inter-workgroup wait timed out); set HSGK_SMALL_GROUPS=1')
    if err == 2:
      raise _lib.HsgkError('segment_by_kmeans: label range too large for the relabel '
                           'table (label_max=%d)' % m[3])
    emb, eloc = out_emb[:n], out_loc[:n]
    labels, cluster, batch = out_lab[:n], out_cluster[:n], out_batch[:n]
    if want_grad:
      ctx.save_for_backward(emb, eloc, norms, rowmap)
      ctx.shape = (B, C, H, W)
    ctx.mark_non_differentiable(labels, cluster, batch)
    return emb, eloc, labels, cluster, batch

  @staticmethod
  def backward(ctx, g_emb, g_eloc, _gl, _gc, _gb):
    emb, eloc, norms, rowmap = ctx.saved_tensors
    B, C, H, W = ctx.shape
    dev = emb.device
    ge = g_emb.contiguous().to(torch.float32) if g_emb is not None else None
    gl = g_eloc.contiguous().to(torch.float32) if g_eloc is not None else None
    with torch.cuda.device(dev):
      gx = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
      _lib.check(_lib.lib().hsgk_segment_by_kmeans_bwd(
          ge.data_ptr() if ge is not None else None, gl.data_ptr() if gl is not None else None,
          emb.data_ptr(), eloc.data_ptr(), norms.data_ptr(),
          rowmap.data_ptr() if rowmap is not None else None, B, C, H, W,
          ctypes.c_float(_lib.EPS), gx.data_ptr(), _lib.stream_ptr()))
    return gx, None, None, None, None, None, None, None, None, None, None


def _segment_by_kmeans_torch(tops, x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, iterations, batch_offset, seed_sb):
  """The same call through the torch-extension binding (hsg_amd/csrc/torch_ops.cpp: hsgk::segment_by_kmeans, C++
  autograd node): one dispatch per attempt; the rare repeats are decided here exactly as in _SegmentByKmeans."""
  dev = x.device
  B, C, H, W = x.shape
  table_cap = B * K if lab is None else max(B * K, min(B * K * 4096, _TABLE_CAP_MAX))

  def run(lab, ign, table_cap, flags=0):
    r = tops.segment_by_kmeans(x, lab, loc, int(loc_sb), seed_map, int(seed_sb), int(K), bool(has_ignore), int(ign),
                               int(iterations), int(batch_offset), int(table_cap), int(flags))
    if lab is None and flags == 0 and _lib.lib().hsgk_small_map_groups(B, C, H, W, K) > 1:
      _lib.defer_status(r[6].view(torch.int32)[10:11],
                        'segment_by_kmeans: the co-operating workgroups of an image waited 10 s for each '
                        'other (labels invalid); set HSGK_SMALL_GROUPS=1')
    return r[5].tolist(), r[:5]

  m, outs = run(lab, ign, table_cap)
  if m[5] == 3:
    m, outs = run(lab, ign, table_cap, flags=1)
  if m[5] == 2:
    kept = lab[lab != ign] if has_ignore else lab
    if kept.numel() and int(kept.min()) < 0:
      raise ValueError('segment_by_kmeans: negative labels are not supported')
    uniq, inv = torch.unique(lab, return_inverse=True)
    D = int(uniq.numel())
    ign_rank = D
    if has_ignore:
      pos = int(torch.searchsorted(uniq, torch.tensor([ign], dtype=uniq.dtype, device=dev)))
      if pos < D and int(uniq[pos]) == ign:
        ign_rank = pos
    m, outs = run(inv.view(B, H, W).contiguous(), ign_rank, max(table_cap, B * K * (D + 1)))
    if m[5] == 0:
      outs = outs[:2] + (uniq[outs[2].clamp(0, D - 1)],) + outs[3:]
  err = m[5]
  if err == 1:
    raise ValueError('segment_by_kmeans: negative labels are not supported')
  if err == 3:
    raise _lib.HsgkError('segment_by_kmeans: the cooperating workgroups of an image were not co-resident '
                         '(inter-workgroup wait timed out); set HSGK_SMALL_GROUPS=1')
  if err == 2:
    raise _lib.HsgkError('segment_by_kmeans: label range too large for the relabel '
                         'table (label_max=%d)' % m[3])
  return tuple(outs)


def _segment_by_kmeans_any(x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, iterations, batch_offset, seed_sb):
  tops = _torch_ops.ops()
  if tops is not None:
    return _segment_by_kmeans_torch(tops, x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, iterations, batch_offset,
                                    seed_sb)
  return _SegmentByKmeans.apply(x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, int(iterations), int(batch_offset),
                                int(seed_sb))


def segment_by_kmeans(embeddings,
                      labels=None,
                      num_clusters=[5, 5],
                      cluster_indices=None,
                      local_features=None,
                      ignore_index=None,
                      iterations=10,
                      batch_offset=None):
  """Per-image spherical k-means over pixel embeddings.

  Contract of reference common.py:270-408.  `embeddings` is [B,C,H,W] float32
  on a ROCm device.  Returns (embeddings [N,C], embeddings_with_loc [N,C+2],
  labels [N], cluster_indices [N], batch_indices [N]) over the N pixels whose
  label differs from `ignore_index`, image-major, row-major inside an image.
  The two float outputs carry gradient back to `embeddings` (location
  features are treated as constants).  `batch_offset` (not in the reference:
  there it is `B * gpu_id`) overrides the first batch index; the default is
  `B * rank` under torch.distributed and `B * device.index` otherwise.
  """
  _require_gpu(embeddings, 'embeddings')
  if embeddings.dim() != 4:
    raise ValueError('embeddings must be [batch, channels, height, width]')
  if embeddings.dtype != torch.float32:
    raise TypeError('embeddings must be float32')
  dev = embeddings.device
  B, C, H, W = embeddings.shape
  x = embeddings.contiguous()
  if cluster_indices is None:
    seed_map, K = _seed_map(num_clusters, H, W, dev)
    seed_sb = 0
  else:
    seed_map, K, seed_sb = _explicit_seeds(cluster_indices, B, H, W, dev)

  if local_features is None:
    loc, loc_sb = _default_loc(H, W, dev), 0
  else:
    _require_gpu(local_features, 'local_features')
    lf = local_features.detach().to(torch.float32)
    if lf.dim() == 4 and lf.stride(0) == 0:
      lf = lf[0]
    loc = lf.contiguous()
    loc_sb = 0 if loc.dim() == 3 else H * W * 2
    if tuple(loc.shape[-3:]) != (H, W, 2):
      raise ValueError('local_features must be [batch, height, width, 2]')

  lab = None
  if labels is not None:
    _require_gpu(labels, 'labels')
    lab = labels.detach().to(torch.int64).contiguous()
    if tuple(lab.shape) != (B, H, W):
      raise ValueError('labels must be [batch, height, width]')
  elif ignore_index is not None:
    lab = torch.zeros((B, H, W), dtype=torch.int64, device=dev)   # common.py:326-329
  has_ignore = ignore_index is not None
  ign = int(ignore_index) if has_ignore else 0
  if batch_offset is None:
    batch_offset = _batch_offset(B, dev)
  if isinstance(K, list):
    return _segment_by_kmeans_grouped(x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, iterations, batch_offset)
  out = _segment_by_kmeans_any(x, lab, loc, loc_sb, seed_map, K, has_ignore, ign, int(iterations), int(batch_offset),
                               int(seed_sb))
  ops.note(out[4], 'ascending', True)          # rows leave image by image: downstream order checks need no read
  return out


def _segment_by_kmeans_grouped(x, lab, loc, loc_sb, dense, counts, has_ignore, ign, iterations,
                               batch_offset):
  """`cluster_indices=` maps whose images carry different numbers of distinct labels: one library
  call per group of images with the same count, then the reference's batch-wide bookkeeping on top
  -- rows image by image in batch order, ids = rank of the segment inside its image + the number
  of segments of the images before it (common.py:398-405: sorted `unique`s over (image, cluster,
  label)).  A few host syncs (rare path; no reference caller passes such maps)."""
  dev = x.device
  B = x.shape[0]
  HW = dense.shape[1]
  per_image = [None] * B
  for K in sorted(set(counts)):
    idx = [b for b in range(B) if counts[b] == K]
    it = torch.tensor(idx, dtype=torch.long, device=dev)
    xs = x.index_select(0, it)
    labs = lab.index_select(0, it).contiguous() if lab is not None else None
    locs = loc if loc_sb == 0 else loc.index_select(0, it).contiguous()
    seeds = dense.index_select(0, it).contiguous()
    emb, eloc, labels, cluster, batch = _segment_by_kmeans_any(
        xs, labs, locs, loc_sb, seeds, K, has_ignore, ign, int(iterations), 0, HW)
    bounds = torch.searchsorted(batch, torch.arange(len(idx) + 1, device=dev)).cpu().tolist()
    for j, b in enumerate(idx):
      s, e = bounds[j], bounds[j + 1]
      per_image[b] = (emb[s:e], eloc[s:e], labels[s:e], cluster[s:e])
  offset = 0
  clusters, batches = [], []
  for b in range(B):
    c = per_image[b][3]
    if c.numel():
      lo, hi = int(c.min()), int(c.max())
      clusters.append(c - lo + offset)
      offset += hi - lo + 1
    else:
      clusters.append(c)
    batches.append(torch.full_like(c, b + int(batch_offset)))
  return (torch.cat([p[0] for p in per_image]), torch.cat([p[1] for p in per_image]),
          torch.cat([p[2] for p in per_image]), torch.cat(clusters), torch.cat(batches))


def kmeans_with_initial_labels(embeddings, initial_labels, max_label=None, iterations=10):
  """Lloyd iterations from given labels (reference common.py:67-97)."""
  _require_gpu(embeddings, 'embeddings')
  x = embeddings.detach().to(torch.float32).contiguous()
  if x.dim() != 2:
    raise ValueError('embeddings must be [num_pixels, embedding_dim]')
  lab = initial_labels.detach().to(torch.int64).contiguous().clone()
  n, d = x.shape
  if n == 0:
    return lab
  K = int(initial_labels.max()) + 1 if max_label is None else int(max_label)
  L = _lib.lib()
  with torch.cuda.device(x.device):
    ws_bytes = L.hsgk_kmeans_workspace_bytes(n, d, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
    _lib.check(L.hsgk_kmeans_with_initial_labels(
        x.data_ptr(), n, d, lab.data_ptr(), K, int(iterations), ws.data_ptr(), ws_bytes,
        _lib.stream_ptr()))
  return lab


def kmeans(embeddings, num_clusters, iterations=10):
  """Grid-seeded Lloyd iterations over ONE `[batch, height, width, channels]` map (reference
  common.py:100-126).  The reference pairs the `[height * width]` seed labels with the flattened
  `[batch * height * width, channels]` embeddings, so it is only defined for batch = 1 (and, as written
  there, calls `initialize_cluster_labels` without its `device` argument); this mirror follows that
  contract with the device taken from `embeddings`.  Returns `[batch, height, width]` labels."""
  _require_gpu(embeddings, 'embeddings')
  shape = embeddings.shape
  if len(shape) != 4 or shape[0] != 1:
    raise ValueError('kmeans: embeddings must be [1, height, width, channels] (see the reference)')
  labels = initialize_cluster_labels(num_clusters, [shape[1], shape[2]], embeddings.device)
  labels = kmeans_with_initial_labels(embeddings.reshape(-1, shape[3]), labels.view(-1), iterations=iterations)
  return labels.view(shape[0], shape[1], shape[2])


def find_nearest_prototypes(embeddings, prototypes):
  """argmax_k <embedding, prototype_k> (reference common.py:44-64)."""
  _require_gpu(embeddings, 'embeddings')
  p = prototypes.detach().to(torch.float32).contiguous()
  x = embeddings.detach().to(torch.float32).contiguous().view(-1, p.shape[-1])
  n, d = x.shape
  out = torch.empty((n,), dtype=torch.int64, device=x.device)
  if n == 0:
    return out
  L = _lib.lib()
  with torch.cuda.device(x.device):
    ws_bytes = L.hsgk_assign_workspace_bytes(n, d, p.shape[0])
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=x.device)
    _lib.check(L.hsgk_find_nearest_prototypes(
        x.data_ptr(), n, d, p.data_ptr(), p.shape[0], out.data_ptr(), ws.data_ptr(),
        ws_bytes, _lib.stream_ptr()))
  return out


def calculate_prototypes_from_labels(embeddings, labels, max_label=None):
  """Mean direction per label (reference common.py:11-41): zeros ->
  scatter_add_ -> normalize_embedding.  Differentiable w.r.t. `embeddings`
  (prototypes carry gradient in the reference's training step)."""
  if max_label is None:
    # one host read, as the reference's `labels.max() + 1` (:33); the smallest label rides along, so a negative
    # label raises here and nothing is left to report later
    lo, hi = torch.aminmax(labels)
    lo, hi = (int(v) for v in torch.stack([lo, hi]).tolist())
    if lo < 0:
      raise _lib.HsgkError('calculate_prototypes_from_labels: negative label %d' % lo)
    return ops.segment_reduce(embeddings, labels, hi + 1, 0)
  # an explicit max_label: a label outside [0, max_label) raises AT THIS CALL, as scatter_add_ does on the
  # reference's CPU path (one host read of the kernel's flag; HSGK_SYNC_ERRORS=0: reported by a later call)
  return ops.segment_reduce(embeddings, labels, int(max_label), 0, strict='call')


def prepare_prototype_labels(semantic_labels, instance_labels, offset=256):
  """Reference common.py:192-218: dense ids of (instance, semantic) pairs in
  sorted order and the semantic label of every id."""
  panoptic = semantic_labels + instance_labels * offset
  proto_panoptic, unique_instance = torch.unique(panoptic, return_inverse=True)
  return proto_panoptic % offset, unique_instance


def find_majority_label_index(semantic_labels, cluster_labels):
  """Reference common.py:221-268: the majority semantic label of every cluster (first
  maximal class, like `torch.argmax` on CPU) and the indices `[M, 1]` of the pixels that
  carry their cluster's majority label.  One histogram + argmax + select pass in
  libhsgk (`hsgk_majority_labels`) instead of the [N, num_classes] one-hot scatter."""
  ops.require_gpu(semantic_labels, 'semantic_labels')
  sem = semantic_labels.reshape(-1).to(torch.int64).contiguous()
  clu = cluster_labels.reshape(-1).to(torch.int64).contiguous()
  n = sem.numel()
  if n == 0:
    raise ValueError('find_majority_label_index: empty input')      # the reference's .max() raises too
  lo = torch.stack([sem.min(), clu.min()])
  hi = torch.stack([sem.max(), clu.max()]).cpu()                    # the reference syncs on both maxima too
  if int(lo.min().item()) < 0:
    raise ValueError('find_majority_label_index: negative label')
  num_classes, num_clusters = int(hi[0].item()) + 1, int(hi[1].item()) + 1
  dev = sem.device
  with torch.cuda.device(dev):
    hist = torch.empty((num_clusters * num_classes,), dtype=torch.int32, device=dev)
    majority = torch.empty((num_clusters,), dtype=torch.int64, device=dev)
    select = torch.empty((n,), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().hsgk_majority_labels(
        sem.data_ptr(), clu.data_ptr(), n, num_clusters, num_classes, hist.data_ptr(),
        majority.data_ptr(), select.data_ptr(), _lib.stream_ptr()))
  return select.nonzero(), majority
