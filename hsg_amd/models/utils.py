"""Drop-in for the cross-GPU glue of `hsg.models.utils` (reference
hsg/models/utils.py:41-240), re-designed for one process per GPU.

The reference runs ONE process, gathers every pixel embedding of every GPU to
an anchor GPU (`scatter_gather.gather`), computes the batch-wide prototype
table there and copies it back (3x per iteration, train.py:190,219).  Here
each rank reduces its own pixels to per-segment SUMS with libhsgk
(segment_reduce, mode 2) and only the small segment table crosses xGMI:

    keys      ONE fixed-capacity all_gather of the ranks' distinct
              (image, cluster, sem, inst) tuples (row count in the buffer header)
    sums      ONE RCCL all_reduce(sum) over the zero-padded [P_total, C + D] table
    labels    decoded from the keys (identical on every rank)

A segment normally lives on one rank (an image is never split), so the
all_reduce only adds zeros to each row and the result is independent of the
reduction order; if the same image id does occur on two ranks its rows merge,
exactly as in the reference.  Payload ~ P_total * (2C+2) * 4 B (a few MB):
latency-bound on xGMI, which is why it is ONE collective.

Every function accepts what the reference accepts -- a list with one tensor per
GPU of this process -- and also a bare tensor (the natural form with one process
per GPU); it returns the same structure it was given.  With
`torch.distributed` initialised the exchange spans all ranks of `group`.
"""
import torch
import torch.distributed as dist

from hsg_amd import ops


# ---- hooks (replaced by the CPU gloo tests with oracle-backed versions) ------
def _segment_sums(rows, ids, count):
  """Raw per-segment sums [count, d] of the local rows (libhsgk, mode 2)."""
  return ops.segment_reduce(rows, ids, count, 2)


def _normalize(table):
  return ops.normalize_rows(table)


def _local_prototypes(rows, ids, count):
  """Normalised per-segment sums in one pass (libhsgk, mode 0): the single-rank path, where the
  sums need not leave the kernel before they are normalised."""
  return ops.segment_reduce(rows, ids, count, 0)


# ---- small helpers -----------------------------------------------------------
def _world(group):
  if dist.is_available() and dist.is_initialized():
    return dist.get_world_size(group)
  return 1


def _as_list(v):
  return (list(v), True) if isinstance(v, (list, tuple)) else ([v], False)


def _cat_to(tensors, device):
  return torch.cat([t.to(device) for t in tensors], 0)


# Rows per rank of the fixed-size gather buffers, per call site.  Every rank starts from the
# same value and grows it from the same gathered counts, so the sizes agree without a
# collective of their own.
_capacity = {}
_CAP_START = 4096
collective_calls = 0        # incremented per collective issued (tests / bench read it)


def _all_gather_rows(t, group, tag):
  """Concatenation over ranks (rank order) of tensors whose first dimension differs per
  rank, in ONE collective and ONE host read: every rank contributes a fixed-capacity byte
  buffer whose first 8 bytes hold its row count.  If some rank has more rows than the
  capacity, every rank sees that in the same counts and repeats the gather with the next
  power of two (kept for later calls).  Returns (rows, counts)."""
  global collective_calls
  world = _world(group)
  if world == 1:
    return t, [t.shape[0]]
  t = t.contiguous()
  n = t.shape[0]
  row_shape = tuple(t.shape[1:])
  row_bytes = t.element_size()
  for s in row_shape:
    row_bytes *= s
  key = (tag, row_bytes)
  while True:
    cap = _capacity.get(key, _CAP_START)
    send = torch.zeros((8 + cap * row_bytes,), dtype=torch.uint8, device=t.device)
    send[:8] = torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(t.device, non_blocking=True)
    m = min(n, cap)
    if m:
      send[8:8 + m * row_bytes] = t[:m].reshape(-1).view(torch.uint8)
    recv = torch.empty((world, 8 + cap * row_bytes), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    collective_calls += 1
    counts = recv[:, :8].contiguous().view(torch.int64).view(-1).tolist()     # the one host read
    need = max(counts)
    if need <= cap:
      break
    while cap < need:
      cap *= 2
    _capacity[key] = cap
  parts = [recv[r, 8:8 + c * row_bytes] for r, c in enumerate(counts) if c]
  if not parts:
    return t[:0], counts
  flat = torch.cat(parts, 0) if len(parts) > 1 else parts[0].contiguous()
  return flat.view(t.dtype).view((-1,) + row_shape), counts


class _AllReduceSum(torch.autograd.Function):
  """y = sum over ranks of x; dL/dx = sum over ranks of dL/dy (every rank's
  loss sees the whole table)."""

  @staticmethod
  def forward(ctx, x, group):
    global collective_calls
    ctx.group = group
    y = x.detach().clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
    collective_calls += 1
    return y

  @staticmethod
  def backward(ctx, g):
    g = g.contiguous().clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
    return g, None


def _compose(tuples, rc, rl):
  """(batch, cluster, semantic, instance) columns -> one sortable key."""
  return ((tuples[:, 0] * rc + tuples[:, 1]) * rl + tuples[:, 2]) * rl + tuples[:, 3]


# ---- hsg/models/utils.py:127-217 -----------------------------------------------
def exchange_prototypes(embeddings, embeddings_with_loc, cluster_indices, batch_indices,
                        semantic_labels, instance_labels, group=None):
  """Per-rank tensors in, batch-wide prototype tables out (same 6 results as
  the reference's gather_clustering_and_update_prototypes, un-listed).

  Two collectives per call: one fixed-capacity all_gather of the ranks' distinct
  (batch, cluster, semantic, instance) tuples and one all_reduce(sum) of the
  zero-padded [P_total, C + D] segment sums.  The reference's dense ids are the
  ranks of those tuples in lexicographic order (its two nested sorted
  `unique`s, utils.py:181-193), which does not depend on the radices used to
  pack them -- so each rank packs with its own maxima and no max-reduction is
  needed before the gather."""
  dev = cluster_indices.device
  c = cluster_indices.reshape(-1).long()
  b = batch_indices.reshape(-1).long()
  sem = semantic_labels.reshape(-1).long()
  inst = instance_labels.reshape(-1).long()
  world = _world(group)

  if c.numel():
    rc = c.max() + 1
    rl = torch.maximum(inst.max(), sem.max()) + 1
    keys = ((b * rc + c) * rl + sem) * rl + inst
    local_keys, local_ids = torch.unique(keys, return_inverse=True)
    rest, li = local_keys // rl, local_keys % rl
    rest, ls = rest // rl, rest % rl
    local_tuples = torch.stack([rest // rc, rest % rc, ls, li], 1)
  else:
    local_ids = torch.zeros((0,), dtype=torch.long, device=dev)
    local_tuples = torch.zeros((0, 4), dtype=torch.long, device=dev)

  if world > 1:
    tuples, _ = _all_gather_rows(local_tuples, group, 'proto_keys')
  else:
    tuples = local_tuples
  if tuples.shape[0]:
    # utils.py:181-189: key order = (batch, cluster, semantic, instance)
    divisor = tuples[:, 1].max() + 1
    lab_div = tuples[:, 2:].max() + 1
    if world > 1:
      global_keys = torch.unique(_compose(tuples, divisor, lab_div))
      slot = torch.searchsorted(global_keys, _compose(local_tuples, divisor, lab_div))
    else:
      global_keys = _compose(tuples, divisor, lab_div)        # already sorted and distinct
      slot = None
    updated_cluster_indices = slot[local_ids] if slot is not None else local_ids
    # utils.py:193-197: labels of every prototype, decoded from the keys
    prototype_instance_labels = global_keys % lab_div
    prototype_semantic_labels = (global_keys // lab_div) % lab_div
    prototype_batch_indices = (global_keys // (lab_div * lab_div)) // divisor
  else:
    global_keys = torch.zeros((0,), dtype=torch.long, device=dev)
    slot = global_keys if world > 1 else None
    updated_cluster_indices = local_ids
    prototype_instance_labels = prototype_semantic_labels = prototype_batch_indices = global_keys
  P = global_keys.shape[0]

  # utils.py:199-202: segment sums -> (exchange) -> normalise
  C = embeddings.shape[-1]
  D = embeddings_with_loc.shape[-1]
  n_local = local_tuples.shape[0]
  if world > 1:
    local = torch.cat([
        _segment_sums(embeddings.reshape(-1, C), local_ids, n_local),
        _segment_sums(embeddings_with_loc.reshape(-1, D), local_ids, n_local)], 1)
    table = torch.zeros((P, C + D), dtype=local.dtype, device=dev)
    table = table.index_add(0, slot, local)
    table = _AllReduceSum.apply(table, group)
    prototypes = _normalize(table[:, :C].contiguous())
    prototypes_with_loc = _normalize(table[:, C:].contiguous())
  else:
    prototypes = _local_prototypes(embeddings.reshape(-1, C), local_ids, n_local)
    prototypes_with_loc = _local_prototypes(embeddings_with_loc.reshape(-1, D), local_ids, n_local)
  return (prototypes, prototypes_with_loc, prototype_semantic_labels,
          prototype_instance_labels, prototype_batch_indices, updated_cluster_indices)


def gather_clustering_and_update_prototypes(embeddings, embeddings_with_loc, cluster_indices,
                                            batch_indices, semantic_labels, instance_labels,
                                            anchor_device=None, group=None):
  """Reference hsg/models/utils.py:127-217 (same arguments and results)."""
  embs, listed = _as_list(embeddings)
  embs_loc, _ = _as_list(embeddings_with_loc)
  c_inds, _ = _as_list(cluster_indices)
  b_inds, _ = _as_list(batch_indices)
  sems, _ = _as_list(semantic_labels)
  insts, _ = _as_list(instance_labels)
  devices = [t.device for t in c_inds]
  sections = [t.shape[0] for t in c_inds]
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  if len(c_inds) > 1:            # several GPUs driven by this one process
    args = [_cat_to(v, anchor) for v in (embs, embs_loc, c_inds, b_inds, sems, insts)]
  else:
    args = [embs[0], embs_loc[0], c_inds[0], b_inds[0], sems[0], insts[0]]
  protos, protos_loc, psem, pinst, pbatch, updated = exchange_prototypes(*args, group=group)
  if not listed:
    return protos, protos_loc, psem, pinst, pbatch, updated
  updated = [u.to(d) for u, d in zip(torch.split(updated, sections), devices)]
  fan = lambda t: [t.to(d) for d in devices]
  return fan(protos), fan(protos_loc), fan(psem), fan(pinst), fan(pbatch), updated


# ---- hsg/models/utils.py:41-74 ---------------------------------------------------
def reorder_image_indices(image_ids_all):
  """Dense image index by first-occurrence order over the gathered id list."""
  uniq, inv = torch.unique(image_ids_all, return_inverse=True)
  n = image_ids_all.shape[0]
  pos = torch.arange(n, dtype=torch.long, device=image_ids_all.device)
  first = torch.full((uniq.shape[0],), n, dtype=torch.long, device=image_ids_all.device)
  first = first.scatter_reduce(0, inv, pos, reduce='amin')
  rank_of = torch.empty_like(first)
  rank_of[torch.argsort(first)] = torch.arange(uniq.shape[0], dtype=torch.long,
                                               device=image_ids_all.device)
  return rank_of[inv]


def gather_and_reorder_image_indices(image_indices, anchor_device=None, group=None):
  """Reference hsg/models/utils.py:41-74: every GPU receives the WHOLE
  re-indexed vector (it is later indexed with `batch_index + B * gpu_id`,
  train.py:208-212)."""
  ids, listed = _as_list(image_indices)
  devices = [t.device for t in ids]
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  local = _cat_to(ids, anchor).long() if len(ids) > 1 else ids[0].long()
  gathered, _ = _all_gather_rows(local, group, 'image_ids')
  full = reorder_image_indices(gathered)
  if not listed:
    return full
  return [full.to(d) for d in devices]


# ---- hsg/models/utils.py:78-124 --------------------------------------------------
def gather_and_update_cluster_mappings(cluster_indices_1, cluster_indices_2,
                                       anchor_device=None, group=None):
  """mapping[i] = the cluster_indices_2 value co-occurring with index i of
  cluster_indices_1 (largest one if several), over all ranks."""
  c1, listed = _as_list(cluster_indices_1)
  c2, _ = _as_list(cluster_indices_2)
  devices = [t.device for t in c1]
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  a = _cat_to(c1, anchor).long() if len(c1) > 1 else c1[0].long()
  b = _cat_to(c2, anchor).long() if len(c2) > 1 else c2[0].long()
  if a.numel():
    local_max = b.max() + 1
    pk = torch.unique(a * local_max + b)
    local_pairs = torch.stack([pk // local_max, pk % local_max], 1)
  else:
    local_pairs = torch.zeros((0, 2), dtype=torch.long, device=a.device)
  # one collective: the ranks' distinct (index_1, index_2) pairs; the radices
  # (utils.py:112,116) are the maxima over the gathered pairs
  pairs2, _ = _all_gather_rows(local_pairs, group, 'cluster_pairs')
  if pairs2.shape[0] == 0:
    mapping = torch.zeros((0,), dtype=torch.long, device=a.device)
  else:
    max_ind = pairs2[:, 1].max() + 1
    pairs = torch.unique(pairs2[:, 0] * max_ind + pairs2[:, 1])   # ascending: later writes win
    size = int(pairs2[:, 0].max()) + 1
    mapping = torch.zeros((size,), dtype=torch.long, device=a.device)
    # sorted order + last-write-wins == the reference's advanced-index assignment on CPU
    keep = torch.ones_like(pairs, dtype=torch.bool)
    keep[:-1] = (pairs[1:] // max_ind) != (pairs[:-1] // max_ind)
    mapping[(pairs // max_ind)[keep]] = (pairs % max_ind)[keep]
  if not listed:
    return mapping
  return [mapping.to(d) for d in devices]


# ---- hsg/models/utils.py:220-240 -------------------------------------------------
def gather_and_update_datas(datas, anchor_device=None, group=None):
  """Concatenation along dim 0 over all GPUs, replicated everywhere."""
  items, listed = _as_list(datas)
  devices = [t.device for t in items]
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  local = _cat_to(items, anchor) if len(items) > 1 else items[0]
  flat = local.reshape(local.shape[0], -1)
  gathered, _ = _all_gather_rows(flat, group, 'datas')
  out = gathered.reshape((gathered.shape[0],) + tuple(local.shape[1:]))
  if not listed:
    return out
  return [out.to(d) for d in devices]


# ---- hsg/models/utils.py:243-309 -------------------------------------------------
def gather_multiset_labels_per_batch_by_nearest_neighbor(
    embeddings, prototypes, semantic_prototype_labels, batch_embedding_labels, batch_prototype_labels,
    num_classes=21, top_k=3, threshold=0.95, label_divisor=255):
  """Multi-hot labels [num_pixels, num_classes] of every pixel from its `top_k` nearest LABELLED
  segments of the same image (similarity >= threshold), as the reference computes them with an
  [N, P] similarity matrix, a masked `topk` and a one-hot sum.  Here the grouped top-k kernel
  (hsgk_topk_prototypes_grouped) keeps only prototypes of the pixel's image with a valid class."""
  from hsg_amd.utils.segsort import eval as segsort_eval
  emb = embeddings.reshape(-1, embeddings.shape[-1])
  proto = prototypes.reshape(-1, emb.shape[-1])
  n = emb.shape[0]
  plab = semantic_prototype_labels.reshape(-1).long()
  qgroup = batch_embedding_labels.reshape(-1).long()
  # prototypes without a valid class never match a pixel's image id
  pgroup = torch.where(plab < num_classes, batch_prototype_labels.reshape(-1).long(),
                       torch.full_like(plab, torch.iinfo(torch.int64).min))
  idx, val = segsort_eval.top_k_indices(emb, proto, top_k, qgroup, pgroup)
  labs = plab[idx.reshape(-1)].view(n, top_k)
  labs = labs.masked_fill(val < threshold, num_classes)          # utils.py:296-297 (unfilled slots: -inf)
  hot = torch.zeros((n, num_classes + 1), dtype=torch.long, device=emb.device)
  hot.scatter_(1, labs, 1)
  return hot[:, :num_classes]
