"""Drop-in for the cross-GPU glue of `hsg.models.utils` (reference
hsg/models/utils.py:41-240), re-designed so that only TABLES cross the interconnect.

The reference runs ONE process, gathers every pixel embedding of every GPU to
an anchor GPU (`scatter_gather.gather`), computes the batch-wide prototype
table there and copies it back (3x per iteration, train.py:190,219).  Here every
device reduces its own pixels to per-segment SUMS with libhsgk (hsg_amd/csrc/exchange.hip:
no sort over the pixel keys, both row sets summed in one fused pass) and only the small
segment table travels:

    keys      ONE fixed-capacity all_gather of the sources' sorted distinct
              (image, cluster, sem, inst) tuples (row count in the block header)
    sums      ONE all_reduce(sum) over the zero-padded [P_total, C + D] table
    labels    decoded from the tuples (identical on every source)

in both multi-GPU modes:

  * one process per GPU (`torch.distributed` initialised): the two collectives run over the
    process group (RCCL under the `nccl` backend), or in-stream on libhsgk's own RCCL
    communicator (`use_library_comm`, the C entry hsgk_exchange_begin / _finish);
  * ONE process driving several GPUs, the reference's DataParallel convention that
    `pyscripts/train/train.py` uses (lists with one tensor per GPU): keys and sums are computed
    on each tensor's OWN device, the tuple blocks and the sum tables (a few MB) are copied to the
    anchor and merged / added there -- pixel rows never leave their device.

A segment normally lives on one source (an image is never split), so the reduction only adds
zeros to each row and the result is independent of its order; if the same image id does occur
on two sources its rows merge, exactly as in the reference.

The per-device work is behind a small backend interface (`keys` / `merge` / `sums` / `finish`,
the phases of include/hsgk.h's exchange section); the CPU tests swap in an oracle-backed backend
and run this very orchestration over `gloo`.
"""
import ctypes
import os
import threading

import torch
import torch.distributed as dist

from hsg_amd import _lib, _torch_ops, ops

HDR = 8                      # int64 words in front of a tuple block (hsg_amd/csrc/exchange.hip kXHdr)
ERR_NEGATIVE, ERR_CAPACITY, ERR_OVERFLOW, ERR_ROWS = 1, 2, 4, 8


# ---- small helpers -----------------------------------------------------------
def _world(group):
  if dist.is_available() and dist.is_initialized():
    return dist.get_world_size(group)
  return 1


def _rank(group):
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(group)
  return 0


def _as_list(v):
  return (list(v), True) if isinstance(v, (list, tuple)) else ([v], False)


def _cat_to(tensors, device):
  return torch.cat([t.to(device) for t in tensors], 0)


def _pow2(v):
  p = 1
  while p < v:
    p *= 2
  return p


# Rows per source of the fixed-size tuple / gather blocks, per (process group, call site).  Every
# rank starts from the same value and grows it from the same gathered counts, so the sizes agree
# without a collective of their own.
_state_lock = threading.Lock()
_capacity = {}
_CAP_START = 4096
collective_calls = 0        # incremented per collective issued (tests / bench read it)


def _group_key(group):
  """A token of the process group that outlives the Python object: c10d names every group it creates (unique in
  the process, the same on all ranks); `id(group)` -- reused by the interpreter once a group has been collected,
  so a NEW group could inherit a dead one's capacities or communicator -- only where no name exists."""
  if group is None:
    return None
  name = getattr(group, 'group_name', None)
  return ('name', name) if name else ('id', id(group))


def _cap_get(group, tag, default=None):
  with _state_lock:
    return _capacity.get((_group_key(group), tag), _CAP_START if default is None else default)


def _cap_set(group, tag, value):
  with _state_lock:
    _capacity[(_group_key(group), tag)] = value


def _count_collective():
  global collective_calls
  with _state_lock:
    collective_calls += 1


def _all_gather_rows(t, group, tag, cap_start=None):
  """Concatenation over ranks (rank order) of tensors whose first dimension differs per
  rank, in ONE collective and ONE host read: every rank contributes a fixed-capacity byte
  buffer whose first 8 bytes hold its row count.  If some rank has more rows than the
  capacity, every rank sees that in the same counts and repeats the gather with the next
  power of two (kept for later calls).  Returns (rows, counts)."""
  world = _world(group)
  if world == 1:
    return t, [t.shape[0]]
  t = t.contiguous()
  n = t.shape[0]
  row_shape = tuple(t.shape[1:])
  row_bytes = t.element_size()
  for s in row_shape:
    row_bytes *= s
  key = '%s/%d' % (tag, row_bytes)
  while True:
    cap = _cap_get(group, key, cap_start)       # (cap_start: whole tensors as rows -- every rank passes the same hint)
    send = torch.zeros((8 + cap * row_bytes,), dtype=torch.uint8, device=t.device)
    send[:8] = torch.tensor([n], dtype=torch.int64).view(torch.uint8).to(t.device, non_blocking=True)
    m = min(n, cap)
    if m:
      send[8:8 + m * row_bytes] = t[:m].reshape(-1).view(torch.uint8)
    recv = torch.empty((world, 8 + cap * row_bytes), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(recv.view(-1), send, group=group)
    _count_collective()
    counts = recv[:, :8].contiguous().view(torch.int64).view(-1).tolist()     # the one host read
    need = max(counts)
    if need <= cap:
      break
    _cap_set(group, key, _pow2(need))
  parts = [recv[r, 8:8 + c * row_bytes] for r, c in enumerate(counts) if c]
  if not parts:
    return t[:0], counts
  flat = torch.cat(parts, 0) if len(parts) > 1 else parts[0].contiguous()
  return flat.view(t.dtype).view((-1,) + row_shape), counts


# ---- transports of the two collectives ----------------------------------------
class _DistTransport:
  """torch.distributed process group (RCCL under the `nccl` backend, gloo in the CPU tests and the
  single-device dry runs)."""

  def __init__(self, group):
    self.group = group
    self.world = _world(group)
    self.rank = _rank(group)

  def all_gather(self, send, recv):
    dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
    _count_collective()

  def all_reduce(self, t):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
    _count_collective()


class _LibraryTransport:
  """libhsgk's own RCCL communicator: both collectives are enqueued on the caller's current
  stream between the kernels (no stream hop, no host sync)."""

  def __init__(self, comm, rank, world):
    self.comm, self.rank, self.world = comm, rank, world

  def all_gather(self, send, recv):
    _lib.check(_lib.lib().hsgk_comm_all_gather_bytes(
        send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(), self.comm, _lib.stream_ptr()))
    _count_collective()

  def all_reduce(self, t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    _lib.check(_lib.lib().hsgk_comm_all_reduce_f32(t.data_ptr(), t.numel(), self.comm, _lib.stream_ptr()))
    _count_collective()


_library_comms = {}
use_library_comm = False     # one process per GPU: in-stream RCCL through libhsgk instead of the process group


def library_transport(group=None):
  """Creates (once per process group) libhsgk's RCCL communicator over the ranks of `group`: rank 0
  draws the unique id, the process group carries its 128 bytes to the others."""
  key = _group_key(group)
  with _state_lock:
    hit = _library_comms.get(key)
  if hit is not None:
    return hit
  world, rank = _world(group), _rank(group)
  L = _lib.lib()
  ident = (ctypes.c_uint8 * 128)()
  if rank == 0:
    _lib.check(L.hsgk_comm_unique_id(ident, 128))
  box = [bytes(ident)]
  if world > 1:
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
  comm = ctypes.c_void_p()
  buf = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
  _lib.check(L.hsgk_comm_init_rank(ctypes.byref(comm), world, rank, buf, 128))
  hit = _LibraryTransport(comm, rank, world)
  with _state_lock:
    _library_comms[key] = hit
  return hit


def _transport(group):
  if _world(group) == 1:
    return None
  if use_library_comm:
    return library_transport(group)
  return _DistTransport(group)


_pin_pool = threading.local()


def _pinned_meta(dev_index):
  """(pinned [8] int64, its event or None) of this thread and device: the buffer travels with the event that guards
  it (an event belongs to the device it was first recorded on; creating one per exchange was ~5 us of host time)."""
  pools = getattr(_pin_pool, 'free', None)
  if pools is None:
    pools = _pin_pool.free = {}
  free = pools.setdefault(dev_index, [])
  return free.pop() if free else (torch.empty((8,), dtype=torch.int64).pin_memory(), None)


def _pinned_release(dev_index, pin, ev):
  _pin_pool.free[dev_index].append((pin, ev))


early_meta = os.environ.get('HSGK_EXCHANGE_EARLY_META', '1') != '0'     # (A/B switch of read_meta_begin)
last_exchange = threading.local()          # .image_stats of this thread's last single-rank exchange (hierarchy.py)


# ---- per-device backend: the phases of include/hsgk.h's exchange section -----
class HsgkExchangeBackend:
  """One source (device) of an exchange: keys -> [gather] -> merge -> sums -> [reduce] -> finish,
  all enqueued on the device's current stream; `read_meta` is the only host read."""

  def __init__(self, emb, emb_loc, c, b, sem, inst, cap_local, cap_total, world):
    ops.require_gpu(emb, 'embeddings')
    self.dev = emb.device
    self.emb = ops.as_rows(emb, torch.float32, 2)
    self.emb_loc = ops.as_rows(emb_loc, torch.float32, 2)
    self.keys_in = [ops.as_rows(t, torch.int64, 1, self.dev) for t in (c, b, sem, inst)]
    n, C = self.emb.shape
    D = self.emb_loc.shape[1]
    if self.emb_loc.shape[0] != n or any(k.shape[0] != n for k in self.keys_in):
      raise ValueError('embeddings, embeddings_with_loc and the index vectors disagree on the number of pixels')
    self.n, self.C, self.D = n, C, D
    self.cap, self.cap_total, self.world = int(cap_local), int(cap_total), int(world)
    nch = (n + _lib.CHUNK - 1) // _lib.CHUNK
    self.pool_rows = max(1, min(n, nch * min(self.cap_total, 256)))
    L = _lib.lib()
    with torch.cuda.device(self.dev):
      wsb = L.hsgk_exchange_workspace_bytes(n, C, D, self.cap, self.cap_total, self.world, self.pool_rows)
      self.ws = torch.empty((wsb,), dtype=torch.uint8, device=self.dev)
      self.table = torch.empty((self.cap_total, C + D), dtype=torch.float32, device=self.dev)
      self.upd = torch.empty((n,), dtype=torch.int64, device=self.dev)
      self.plab = torch.empty((3, self.cap_total), dtype=torch.int64, device=self.dev)
      self.meta = torch.empty((8,), dtype=torch.int64, device=self.dev)
    self.args = _lib.ExchangeArgs(
        embeddings=self.emb.data_ptr(), embeddings_loc=self.emb_loc.data_ptr(),
        cluster=self.keys_in[0].data_ptr(), batch=self.keys_in[1].data_ptr(),
        semantic=self.keys_in[2].data_ptr(), instance=self.keys_in[3].data_ptr(),
        n=n, C=C, D=D, cap_local=self.cap, cap_total=self.cap_total, pool_rows=self.pool_rows, eps=_lib.EPS,
        table=self.table.data_ptr(), prototypes=None, prototypes_loc=None, norms=None,
        proto_semantic=self.plab[0].data_ptr(), proto_instance=self.plab[1].data_ptr(),
        proto_batch=self.plab[2].data_ptr(), updated_cluster=self.upd.data_ptr(), meta=self.meta.data_ptr(),
        workspace=self.ws.data_ptr(), workspace_bytes=wsb)

  def _view(self, ptr, words):
    off = ptr - self.ws.data_ptr()
    return self.ws[off:off + 8 * words].view(torch.int64)

  def keys(self):
    L = _lib.lib()
    with torch.cuda.device(self.dev):
      _lib.check(L.hsgk_exchange_keys(ctypes.byref(self.args), self.world, _lib.stream_ptr()))
    nbytes = ctypes.c_size_t(0)
    ptr = L.hsgk_exchange_send_block(ctypes.byref(self.args), self.world, ctypes.byref(nbytes))
    return self._view(ptr, HDR + 4 * self.cap)

  def recv_blocks(self):
    ptr = _lib.lib().hsgk_exchange_recv_blocks(ctypes.byref(self.args), self.world)
    return self._view(ptr, self.world * (HDR + 4 * self.cap)).view(self.world, HDR + 4 * self.cap)

  def merge(self, my_rank):
    """-> slots int32 [world, cap] (my_rank < 0: every source's row is filled)."""
    with torch.cuda.device(self.dev):
      slots = torch.empty((self.world, self.cap), dtype=torch.int32, device=self.dev)
      _lib.check(_lib.lib().hsgk_exchange_merge(ctypes.byref(self.args), my_rank, self.world,
                                                 None if (self.world == 1 and my_rank == 0) else slots.data_ptr(),
                                                 _lib.stream_ptr()))
    self._slots = None if (self.world == 1 and my_rank == 0) else slots
    return slots

  def sums(self, my_rank, slots_row=None):
    """ids + fused segment sums of both row sets into self.table (raw sums at the global rows)."""
    if slots_row is None and self._slots is not None:
      slots_row = self._slots[my_rank]
    if slots_row is not None:
      slots_row = slots_row.contiguous()
    self._keep = slots_row
    with torch.cuda.device(self.dev):
      _lib.check(_lib.lib().hsgk_exchange_sums(ctypes.byref(self.args), max(my_rank, 0), self.world,
                                                slots_row.data_ptr() if slots_row is not None else None,
                                                _lib.stream_ptr()))

  def read_meta_begin(self):
    """After `merge`: the meta block (complete from there on) starts its way to pinned host memory, so that the
    one host read of the exchange waits for keys + merge only and the sums run behind it."""
    pin, ev = _pinned_meta(self.dev.index)
    with torch.cuda.device(self.dev):
      pin.copy_(self.meta, non_blocking=True)
      if ev is None:
        ev = torch.cuda.Event()
      ev.record()
    self._pending = (pin, ev)

  def read_meta(self):
    pending = getattr(self, '_pending', None)
    if pending is not None:
      pin, ev = pending
      ev.synchronize()
      m = pin.tolist()
      _pinned_release(self.dev.index, pin, ev)
      self._pending = None
    else:
      m = self.meta.cpu().tolist()
    self.image_stats = (m[4], m[5])       # distinct first keys, longest run (-1: not computed)
    return m[0], m[1], m[2], m[3]

  def finish(self, table_rows):
    """table_rows [rows, C + D] (raw, already reduced) -> prototypes, prototypes_with_loc, norms [rows, 2]."""
    rows = table_rows.shape[0]
    with torch.cuda.device(self.dev):
      pa = torch.empty((rows, self.C), dtype=torch.float32, device=self.dev)
      pb = torch.empty((rows, self.D), dtype=torch.float32, device=self.dev)
      norms = torch.empty((rows, 2), dtype=torch.float32, device=self.dev)
      if rows:
        a = _lib.ExchangeArgs(C=self.C, D=self.D, cap_total=rows, eps=_lib.EPS, table=table_rows.data_ptr(),
                              prototypes=pa.data_ptr(), prototypes_loc=pb.data_ptr(), norms=norms.data_ptr())
        _lib.check(_lib.lib().hsgk_exchange_finish(ctypes.byref(a), rows, None, 1, _lib.stream_ptr()))
    return pa, pb, norms

  # backward pieces (mirror of finish and of the row -> segment map)
  @staticmethod
  def finish_bwd(g_pa, g_pb, pa, pb, norms):
    """d(loss)/d(raw sums) [rows, C + D] from the gradients of the two normalised tables."""
    rows, C = pa.shape
    D = pb.shape[1]
    dev = pa.device
    L = _lib.lib()
    with torch.cuda.device(dev):
      g = torch.zeros((rows, C + D), dtype=torch.float32, device=dev)
      for gp, out, col, d, o in ((g_pa, pa, 0, C, 0), (g_pb, pb, 1, D, C)):
        if gp is None or rows == 0:
          continue
        gseg = torch.empty((rows, d), dtype=torch.float32, device=dev)
        aux = norms[:, col].contiguous()
        _lib.check(L.hsgk_segment_reduce_bwd(gp.contiguous().data_ptr(), out.data_ptr(), aux.data_ptr(), None,
                                             0, d, rows, 0, ctypes.c_float(_lib.EPS), gseg.data_ptr(), None,
                                             _lib.stream_ptr()))
        g[:, o:o + d] = gseg
    return g

  @staticmethod
  def rows_bwd(g_table, upd, C, D, need):
    """gradient of the pixel rows: row i receives the gradient of its segment's sum."""
    dev = upd.device
    n, rows = upd.shape[0], g_table.shape[0]
    L = _lib.lib()
    outs = []
    with torch.cuda.device(dev):
      for want, d, o in ((need[0], C, 0), (need[1], D, C)):
        if not want:
          outs.append(None)
          continue
        gseg = g_table[:, o:o + d].contiguous()
        gx = torch.empty((n, d), dtype=torch.float32, device=dev)
        _lib.check(L.hsgk_segment_reduce_bwd(gseg.data_ptr(), gseg.data_ptr(), None, upd.data_ptr(), n, d, rows, 2,
                                             ctypes.c_float(_lib.EPS), gseg.data_ptr(), gx.data_ptr(),
                                             _lib.stream_ptr()))
        outs.append(gx)
    return outs


backend_class = HsgkExchangeBackend       # the CPU tests install an oracle-backed class with the same methods


def _raise_for(err):
  if err & ERR_NEGATIVE:
    raise ValueError('prototype exchange: negative cluster / batch / label values are not supported')
  if err & ERR_OVERFLOW:
    raise _lib.HsgkError('prototype exchange: (batch, cluster, semantic, instance) does not pack into 62 bits')
  raise _lib.HsgkError('prototype exchange: device-side error %d' % err)


class _Exchange(torch.autograd.Function):
  """One process per GPU (or a single GPU): this rank's rows in, the batch-wide tables out."""

  @staticmethod
  def forward(ctx, emb, emb_loc, c, b, sem, inst, group, tag, local=False):
    tr = None if local else _transport(group)
    world = tr.world if tr is not None else 1
    rank = tr.rank if tr is not None else 0
    while True:
      cap = _cap_get(group, tag)
      be = backend_class(emb, emb_loc, c, b, sem, inst, cap, cap * world, world)
      send = be.keys()
      if tr is not None:
        tr.all_gather(send, be.recv_blocks())
      be.merge(rank)
      if early_meta and hasattr(be, 'read_meta_begin'):
        be.read_meta_begin()
      be.sums(rank)
      n_local, rows, err, need = be.read_meta()          # the exchange's one host read
      last_exchange.image_stats = getattr(be, 'image_stats', None) if world == 1 else None
      if err & (ERR_CAPACITY | ERR_ROWS):
        # some rank has more distinct tuples than the blocks hold: every rank sees the same
        # counts in the gathered headers and regrows alike
        _cap_set(group, tag, _pow2(max(need, 2 * cap)))
        continue
      if err:
        _raise_for(err)
      break
    table = be.table[:rows]
    if tr is not None and rows:
      tr.all_reduce(table)
    pa, pb, norms = be.finish(table)
    psem, pinst, pbatch = be.plab[0, :rows], be.plab[1, :rows], be.plab[2, :rows]
    ctx.save_for_backward(pa, pb, norms, be.upd)
    ctx.tr, ctx.be_cls = tr, type(be)
    ctx.mark_non_differentiable(psem, pinst, pbatch, be.upd)
    return pa, pb, psem, pinst, pbatch, be.upd

  @staticmethod
  def backward(ctx, g_pa, g_pb, _g2, _g3, _g4, _g5):
    pa, pb, norms, upd = ctx.saved_tensors
    g = ctx.be_cls.finish_bwd(g_pa, g_pb, pa, pb, norms)
    if ctx.tr is not None and g.numel():
      ctx.tr.all_reduce(g)              # every rank's loss sees the whole table
    gx, gl = ctx.be_cls.rows_bwd(g, upd, pa.shape[1], pb.shape[1], ctx.needs_input_grad[:2])
    return gx, gl, None, None, None, None, None, None, None


# ---- hsg/models/utils.py:127-217 -----------------------------------------------
def exchange_prototypes(embeddings, embeddings_with_loc, cluster_indices, batch_indices,
                        semantic_labels, instance_labels, group=None, tag='proto', local=False):
  """Per-rank tensors in, batch-wide prototype tables out (same 6 results as
  the reference's gather_clustering_and_update_prototypes, un-listed).

  Two collectives per call: one fixed-capacity all_gather of the ranks' sorted distinct
  (batch, cluster, semantic, instance) tuples and one all_reduce(sum) of the
  zero-padded [P_total, C + D] segment sums.  The reference's dense ids are the
  ranks of those tuples in lexicographic order (its two nested sorted
  `unique`s, utils.py:181-193), which does not depend on the radices used to
  pack them -- so each rank packs with its own maxima and no max-reduction is
  needed before the gather.  `local`: this rank's rows only, no collective (the per-GPU prototype tables
  of predictions/segsort.py:224-244)."""
  C = embeddings.shape[-1]
  D = embeddings_with_loc.shape[-1]
  tops = _torch_ops.ops() if backend_class is HsgkExchangeBackend else None
  if tops is not None and (local or _world(group) == 1):
    # one rank: the whole exchange is ONE op of the torch-extension binding (hsg_amd/csrc/torch_ops.cpp: keys ->
    # merge -> meta on its way to the host -> sums -> finish, backward as a C++ autograd node) instead of the ~25
    # dispatches of the phase-by-phase mirror below, which the multi-rank exchange needs for its two collectives
    ops.require_gpu(embeddings, 'embeddings')
    while True:
      cap = _cap_get(group, tag)
      pa, pb, psem, pinst, pbatch, upd, meta = tops.exchange_local(
          embeddings, embeddings_with_loc, cluster_indices, batch_indices, semantic_labels, instance_labels, cap)
      m = meta.tolist()
      last_exchange.image_stats = (m[4], m[5])
      if m[2] & (ERR_CAPACITY | ERR_ROWS):
        _cap_set(group, tag, _pow2(max(m[3], 2 * cap)))
        continue
      if m[2]:
        _raise_for(m[2])
      return pa, pb, psem, pinst, pbatch, upd
  return _Exchange.apply(embeddings.reshape(-1, C), embeddings_with_loc.reshape(-1, D), cluster_indices,
                         batch_indices, semantic_labels, instance_labels, group, tag, bool(local))


class _ExchangeList(torch.autograd.Function):
  """ONE process driving several GPUs (lists of per-GPU tensors, train.py:190-223): keys and sums on
  every tensor's own device; only tuple blocks and [P, C + D] tables move to the anchor."""

  @staticmethod
  def forward(ctx, ndev, anchor, tag, *tensors):
    embs, locs = tensors[:ndev], tensors[ndev:2 * ndev]
    cs, bs, sems, insts = (tensors[(2 + j) * ndev:(3 + j) * ndev] for j in range(4))
    ai = _anchor_index(embs, anchor)
    while True:
      cap = _cap_get(None, tag)
      # the anchor's workspace holds the `ndev` gathered tuple blocks and the merge; the others only their own
      bes = [backend_class(embs[g], locs[g], cs[g], bs[g], sems[g], insts[g], cap, cap * ndev,
                           ndev if g == ai else 1) for g in range(ndev)]
      sends = [be.keys() for be in bes]
      if ndev > 1:
        recv = bes[ai].recv_blocks()
        for g in range(ndev):
          recv[g].copy_(sends[g], non_blocking=True)     # a tuple block: 8 + 4 cap words
        slots = bes[ai].merge(-1)
        for g in range(ndev):
          bes[g].sums(-1, slots[g].to(embs[g].device, non_blocking=True))
      else:
        bes[0].merge(0)
        if early_meta and hasattr(bes[0], 'read_meta_begin'):
          bes[0].read_meta_begin()
        bes[0].sums(0)
      # one host read: the error bits of every device travel in its block header and are merged here
      _n, rows, err, need = bes[ai].read_meta()
      if err & (ERR_CAPACITY | ERR_ROWS):
        _cap_set(None, tag, _pow2(max(need, 2 * cap)))
        continue
      if err:
        _raise_for(err)
      break
    table = bes[ai].table[:rows]
    for g in range(ndev):                                  # the tables, not the pixels, cross the links
      if g != ai:
        table += bes[g].table[:rows].to(anchor, non_blocking=True)
    pa, pb, norms = bes[ai].finish(table)
    plab = bes[ai].plab
    upds = [be.upd for be in bes]
    ctx.save_for_backward(pa, pb, norms, *upds)
    ctx.ndev, ctx.be_cls = ndev, type(bes[0])
    outs = (pa, pb, plab[0, :rows], plab[1, :rows], plab[2, :rows]) + tuple(upds)
    ctx.mark_non_differentiable(*outs[2:])
    return outs

  @staticmethod
  def backward(ctx, g_pa, g_pb, *_rest):
    pa, pb, norms = ctx.saved_tensors[:3]
    upds = ctx.saved_tensors[3:]
    ndev = ctx.ndev
    g = ctx.be_cls.finish_bwd(g_pa, g_pb, pa, pb, norms)
    grads_e, grads_l = [], []
    for k in range(ndev):
      gd = g.to(upds[k].device)
      ge, gl = ctx.be_cls.rows_bwd(gd, upds[k], pa.shape[1], pb.shape[1],
                                   (ctx.needs_input_grad[3 + k], ctx.needs_input_grad[3 + ndev + k]))
      grads_e.append(ge)
      grads_l.append(gl)
    return (None, None, None) + tuple(grads_e) + tuple(grads_l) + (None,) * (4 * ndev)


def _anchor_index(tensors, anchor):
  for i, t in enumerate(tensors):
    if t.device == anchor:
      return i
  return 0


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
  # (one GPU per process -- also when it arrives as one-element lists, the way pyscripts/train/train.py:190,219 call
  #  it: the single-tensor exchange, i.e. ONE op of the torch-extension binding when this process is the world)
  if not listed or len(c_inds) == 1:
    res = exchange_prototypes(embs[0], embs_loc[0], c_inds[0], b_inds[0], sems[0], insts[0], group=group)
    if _world(group) == 1:
      ops.note(res[5], 'index_count', int(res[0].shape[0]))
    if not listed:
      return res
    return tuple([r] for r in res)
  if _world(group) > 1:
    # several GPUs per process AND several processes (not a reference configuration): the local
    # tensors are joined on the anchor first, then the ranks exchange as usual
    anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
    sections = [t.shape[0] for t in c_inds]
    args = [_cat_to(v, anchor) for v in (embs, embs_loc, c_inds, b_inds, sems, insts)]
    protos, protos_loc, psem, pinst, pbatch, updated = exchange_prototypes(*args, group=group)
    updated = [u.to(d) for u, d in zip(torch.split(updated, sections), devices)]
    fan = lambda t: [t.to(d) for d in devices]
    return fan(protos), fan(protos_loc), fan(psem), fan(pinst), fan(pbatch), updated
  ndev = len(c_inds)
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  if anchor.type == 'cuda' and anchor.index is None:
    anchor = torch.device('cuda', torch.cuda.current_device())
  if anchor not in devices:
    anchor = devices[0]
  C, D = embs[0].shape[-1], embs_loc[0].shape[-1]
  outs = _ExchangeList.apply(ndev, anchor, 'proto_list',
                             *[e.reshape(-1, C) for e in embs], *[e.reshape(-1, D) for e in embs_loc],
                             *c_inds, *b_inds, *sems, *insts)
  protos, protos_loc, psem, pinst, pbatch = outs[:5]
  if ndev == 1:
    ops.note(outs[5], 'index_count', int(protos.shape[0]))
  fan = lambda t: [t.to(d) for d in devices]
  return fan(protos), fan(protos_loc), fan(psem), fan(pinst), fan(pbatch), list(outs[5:])


# ---- hsg/models/utils.py:12-38 ---------------------------------------------------
def get_params(model, prefixs, suffixes, exclude=None):
  """The trainable parameters of the sub-modules named in `prefixs` whose (last component of the) name starts or
  ends with one of `suffixes`; `exclude`: a list of full parameter names, or a substring.  Optimiser bookkeeping
  of the training script (reference :12-38), kept so that the module swap of INTEGRATION.md is complete."""
  wanted = set(prefixs)
  for mod_name, module in model.named_modules():
    if mod_name not in wanted:
      continue
    for par_name, par in module.named_parameters():
      full = mod_name + '.' + par_name
      if isinstance(exclude, list) and full in exclude:
        continue
      if isinstance(exclude, str) and exclude in full:
        continue
      last = full.split('.')[-1]
      # (one yield per matching suffix, as the reference's nested loop does)
      for suffix in suffixes:
        if par.requires_grad and (last.startswith(suffix) or full.endswith(suffix)):
          yield par


# ---- hsg/models/utils.py:41-74 ---------------------------------------------------
def gather_and_reorder_image_indices(image_indices, anchor_device=None, group=None):
  """Reference hsg/models/utils.py:41-74: every GPU receives the WHOLE
  re-indexed vector (it is later indexed with `batch_index + B * gpu_id`,
  train.py:208-212)."""
  ids, listed = _as_list(image_indices)
  devices = [t.device for t in ids]
  anchor = torch.device(anchor_device) if anchor_device is not None else devices[0]
  local = _cat_to(ids, anchor).long() if len(ids) > 1 else ids[0].long()
  gathered, _ = _all_gather_rows(local, group, 'image_ids')
  # The vector has two entries per image of the batch: its dense re-indexing (first-occurrence order, :60-72) is
  # done on the host copy the order checks of generate_clusters want anyway -- one read and one small upload
  # instead of `unique` (itself a host synchronisation), a scatter-min, a sort and an index_put on the device.
  first = {}
  host = [first.setdefault(v, len(first)) for v in gathered.tolist()]
  full = torch.tensor(host, dtype=torch.long, device=gathered.device)
  if not listed:
    return ops.note(full, 'host', host)
  return [ops.note(full.to(d), 'host', host) for d in devices]


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
  if _world(group) == 1:
    # one process: mapping[i] = the LARGEST partner of index i -- what the reference's duplicate-index
    # assignment leaves behind on sorted pairs (utils.py:112-121) -- is one scatter-max
    if a.numel():
      # (the reference reads this maximum on the host; an index vector that comes straight from the prototype
      #  exchange carries its table length -- every table row has a pixel -- and needs no read)
      size = ops.noted(a, 'index_count')
      if size is None:
        size = int(a.max()) + 1
      mapping = torch.zeros((size,), dtype=torch.long, device=a.device).scatter_reduce(0, a, b, 'amax', include_self=True)
    else:
      mapping = torch.zeros((0,), dtype=torch.long, device=a.device)
    if not listed:
      return mapping
    return [mapping.to(d) for d in devices]
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
  gathered, _ = _all_gather_rows(flat, group, 'datas', cap_start=_pow2(max(int(flat.shape[0]), 16)))
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
