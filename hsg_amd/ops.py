"""torch.autograd glue over the libhsgk C ABI (internal; the public surface is
the reference-shaped modules under hsg_amd/utils and hsg_amd/models)."""
import ctypes
import os
import threading

import torch

from hsg_amd import _lib, _torch_ops

EPS = 1e-12


# Host-side facts about a device tensor, remembered on the tensor OBJECT by the function that had them on the
# host anyway (an exchange knows its table length, the image-id gather its id list) and read back by the next
# function that would otherwise fetch them from the device with a stalling read.  Valid for that object and its
# current version only: any op makes a new object without the note, an in-place write bumps `_version`.
# NOT covered (the caveat of every `_version` check): writes through `.data`, `set_()` or a foreign kernel writing
# through `data_ptr()` -- this package only notes tensors it has just produced and hands them straight to the next
# function of the same step.  HSGK_NO_NOTES=1 turns the notes off (every consumer then reads the device).
_notes_on = os.environ.get('HSGK_NO_NOTES', '') in ('', '0')


def note(t, name, value):
  if not _notes_on:
    return t
  try:
    setattr(t, '_hsg_' + name, (value, t._version))
  except (AttributeError, RuntimeError):
    pass
  return t


def noted(t, name):
  h = getattr(t, '_hsg_' + name, None) if torch.is_tensor(t) else None
  return h[0] if h is not None and h[1] == t._version else None


_pin_pool = threading.local()


def read_i64(dev_tensor):
  """Host list of a small int64 device vector through pinned memory (a pageable `.cpu()` stages the copy and costs
  ~50 us more); the stream is waited for up to the copy only."""
  pools = getattr(_pin_pool, 'free', None)
  if pools is None:
    pools = _pin_pool.free = {}
  free = pools.setdefault(dev_tensor.device.index, [])      # (an event belongs to the device it was first recorded on)
  n = dev_tensor.numel()
  # (the pinned buffer travels with its event: creating one per read was ~5 us of the read)
  pin, ev = free.pop() if free else (torch.empty((64,), dtype=torch.int64).pin_memory(), None)
  with torch.cuda.device(dev_tensor.device):
    pin[:n].copy_(dev_tensor.view(-1), non_blocking=True)
    if ev is None:
      ev = torch.cuda.Event()
    ev.record()
  ev.synchronize()
  out = pin[:n].tolist()
  free.append((pin, ev))
  return out


def as_rows(t, dtype, ndim, device=None):
  """`t` detached, contiguous, of `dtype`, flattened to ndim 1 or 2 (rows x last dimension): the tensor ITSELF when
  it already is all of that -- the usual case on the hot path, where the four-op chain
  detach / reshape / to / contiguous costs ~6 us of dispatch per operand and a training step has ~60 operands."""
  if t.requires_grad:
    t = t.detach()
  if t.dim() != ndim:
    t = t.reshape(-1) if ndim == 1 else t.reshape(-1, t.shape[-1])
  if t.dtype is not dtype or (device is not None and t.device != device):
    t = t.to(device=device if device is not None else t.device, dtype=dtype)
  return t if t.is_contiguous() else t.contiguous()


def require_gpu(t, name):
  if not t.is_cuda:
    raise _lib.HsgkError('%s must be a ROCm device tensor (got %s); hsg_amd has '
                         'no CPU path' % (name, t.device))


def _segment_reduce_fwd(x, labels, P, mode, strict=False):
  n, d = x.shape
  _lib.poll_deferred()
  L = _lib.lib()
  dev = x.device
  with torch.cuda.device(dev):
    out = torch.empty((P, d), dtype=torch.float32, device=dev)
    aux = torch.empty((max(P, 1),), dtype=torch.float32, device=dev)
    status = torch.empty((1,), dtype=torch.int32, device=dev)
    wsb = L.hsgk_segment_reduce_workspace_bytes(n, d, P)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    _lib.check(L.hsgk_segment_reduce(
        x.data_ptr(), n, d, labels.data_ptr(), P, mode, ctypes.c_float(EPS), out.data_ptr(),
        aux.data_ptr(), status.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()))
    if strict:      # the reference's scatter_add_ rejects such labels: raised here ('call') or by a later call
      _lib.defer_status(status, 'segment_reduce: a label lies outside [0, %d)' % P, at_call=(strict == 'call'))
  return out, aux


class SegmentReduce(torch.autograd.Function):
  """mode 0 prototypes (normalised sums), 1 means, 2 raw sums."""

  @staticmethod
  def forward(ctx, x, labels, P, mode, strict=False):
    x = x.contiguous()
    out, aux = _segment_reduce_fwd(x.detach(), labels, P, mode, strict)
    ctx.save_for_backward(out, aux, labels)
    ctx.meta = (x.shape[0], x.shape[1], P, mode)
    return out

  @staticmethod
  def backward(ctx, gout):
    out, aux, labels = ctx.saved_tensors
    n, d, P, mode = ctx.meta
    gout = gout.contiguous().to(torch.float32)
    dev = gout.device
    with torch.cuda.device(dev):
      gseg = torch.empty((max(P, 1), d), dtype=torch.float32, device=dev)
      gx = torch.empty((n, d), dtype=torch.float32, device=dev)
      _lib.check(_lib.lib().hsgk_segment_reduce_bwd(
          gout.data_ptr(), out.data_ptr(), aux.data_ptr(), labels.data_ptr(), n, d, P, mode,
          ctypes.c_float(EPS), gseg.data_ptr(), gx.data_ptr(), _lib.stream_ptr()))
    return gx, None, None, None, None


def segment_reduce(x, labels, P, mode, strict=False):
  """Rows whose label lies outside [0, P) are skipped; with `strict` that is an error: strict='call' raises at this
  call (one host read; HSGK_SYNC_ERRORS=0: as True), strict=True by a later libhsgk call once the kernels have run
  (at once with HSGK_SYNC_ERRORS=1)."""
  require_gpu(x, 'x')
  if x.dtype != torch.float32:
    raise TypeError('x must be float32')
  x2 = x.reshape(-1, x.shape[-1])
  lab = labels.reshape(-1).to(torch.int64).contiguous()
  if lab.shape[0] != x2.shape[0]:
    raise ValueError('labels and rows disagree: %d vs %d' % (lab.shape[0], x2.shape[0]))
  tops = _torch_ops.ops()
  if tops is not None:                 # the torch-extension binding: one dispatch, C++ autograd node
    _lib.poll_deferred()
    out, status = tops.segment_reduce(x2, lab, int(P), int(mode))
    if strict:
      _lib.defer_status(status, 'segment_reduce: a label lies outside [0, %d)' % P, at_call=(strict == 'call'))
    return out
  return SegmentReduce.apply(x2, lab, int(P), int(mode), strict if strict == 'call' else bool(strict))


class NormalizeRows(torch.autograd.Function):
  """normalize_embedding with the norm-clamp semantics of the reference."""

  @staticmethod
  def forward(ctx, x, eps):
    x2 = x.detach().contiguous()
    d = x2.shape[-1]
    n = x2.numel() // d if d else 0
    dev = x2.device
    with torch.cuda.device(dev):
      out = torch.empty_like(x2)
      norms = torch.empty((max(n, 1),), dtype=torch.float32, device=dev)
      _lib.check(_lib.lib().hsgk_normalize_rows(
          x2.data_ptr(), n, d, ctypes.c_float(eps), out.data_ptr(), norms.data_ptr(),
          _lib.stream_ptr()))
    ctx.save_for_backward(out, norms)
    ctx.eps = eps
    return out

  @staticmethod
  def backward(ctx, gout):
    out, norms = ctx.saved_tensors
    d = out.shape[-1]
    n = out.numel() // d if d else 0
    g = gout.contiguous().to(torch.float32)
    with torch.cuda.device(out.device):
      gx = torch.empty_like(out)
      # a normalised row is a prototype row: reuse the mode-0 segment backward
      _lib.check(_lib.lib().hsgk_segment_reduce_bwd(
          g.data_ptr(), out.data_ptr(), norms.data_ptr(), None, 0, d, n, 0,
          ctypes.c_float(ctx.eps), gx.data_ptr(), None, _lib.stream_ptr()))
    return gx, None


def normalize_rows(x, eps=EPS):
  require_gpu(x, 'embeddings')
  if x.dtype != torch.float32:
    raise TypeError('embeddings must be float32')
  return NormalizeRows.apply(x, float(eps))
