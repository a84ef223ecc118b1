"""Drop-in for `hsg.utils.segsort.loss.SegSortLoss` / `SetSegSortLoss` on MI355X.

Same constructors and forward signatures as the reference module
(hsg/utils/segsort/loss.py:133-190, 193-251).  The [N,P] similarity matrix of the
reference is never built in the forward pass: libhsgk streams the pixels
against 64-prototype blocks on fp32 MFMA and folds exp(kappa * cos) straight
into per-pixel own / same / different sums (csrc/loss.hip), for up to three
label sets per pass (`segsort_losses`: the three losses of
hsg/models/predictions/hsg.py:78-155 share E P^T).  Backward recomputes the
score tiles and contracts them in place (g_emb = W P, g_proto = W^T E) -- no
[N,P] storage there either.
"""
import ctypes

import torch
from torch.nn.modules.loss import _Loss

from hsg_amd import _lib, _torch_ops, ops


MAX_LABEL_SETS = 3


def _make_sets(sems, psems, kappas, modes):
  arr = (_lib.LossSet * len(sems))()
  for i, (a, b, k, m) in enumerate(zip(sems, psems, kappas, modes)):
    arr[i].sem, arr[i].psem, arr[i].kappa, arr[i].mode = a.data_ptr(), b.data_ptr(), float(k), int(m)
  return arr


class _SegSortNLL(torch.autograd.Function):
  """Per-pixel negative log-likelihood (reference loss.py:15-82) of up to three label sets
  over the same embeddings / own-prototype indices / prototype table: [L, n]."""

  @staticmethod
  def forward(ctx, embeddings, instance_labels, prototypes, kappas, modes, groups, *labels):
    L = len(kappas)
    sems, psems = labels[:L], labels[L:]
    qg, pg = groups if groups is not None else (None, None)
    emb = embeddings.detach().contiguous()
    proto = prototypes.detach().contiguous()
    n, c = emb.shape
    P = proto.shape[0]
    dev = emb.device
    lib = _lib.lib()
    with torch.cuda.device(dev):
      nll = torch.empty((L, n), dtype=torch.float32, device=dev)
      num = torch.empty((L, n), dtype=torch.float32, device=dev)
      den = torch.empty((L, n), dtype=torch.float32, device=dev)
      use_same = torch.empty((L, n), dtype=torch.int32, device=dev)
      wsb = lib.hsgk_segsort_loss_workspace_bytes(n, c, P, L)
      ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
      sets = _make_sets(sems, psems, kappas, modes)
      _lib.check(lib.hsgk_segsort_loss_fwd(
          emb.data_ptr(), n, c, instance_labels.data_ptr(), proto.data_ptr(), P, L, sets,
          qg.data_ptr() if qg is not None else None, pg.data_ptr() if pg is not None else None,
          nll.data_ptr(), num.data_ptr(), den.data_ptr(), use_same.data_ptr(), ws.data_ptr(), wsb,
          _lib.stream_ptr()))
    ctx.save_for_backward(emb, proto, instance_labels, num, den, use_same, *labels)
    ctx.cfg = (tuple(float(k) for k in kappas), tuple(int(m) for m in modes))
    ctx.groups = (qg, pg)
    return nll

  @staticmethod
  def backward(ctx, gnll):
    emb, proto, inst, num, den, use_same = ctx.saved_tensors[:6]
    labels = ctx.saved_tensors[6:]
    kappas, modes = ctx.cfg
    qg, pg = ctx.groups
    L = len(kappas)
    n, c = emb.shape
    P = proto.shape[0]
    dev = emb.device
    lib = _lib.lib()
    gscale = gnll.detach().to(torch.float32).contiguous()
    want_e, want_p = ctx.needs_input_grad[0], ctx.needs_input_grad[2]
    g_emb = g_proto = None
    with torch.cuda.device(dev):
      if want_e:
        g_emb = torch.empty((n, c), dtype=torch.float32, device=dev)
      if want_p:
        g_proto = torch.empty((P, c), dtype=torch.float32, device=dev)
      if want_e or want_p:
        wsb = lib.hsgk_segsort_loss_bwd_workspace_bytes(n, c, P, L)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        sets = _make_sets(labels[:L], labels[L:], kappas, modes)
        _lib.check(lib.hsgk_segsort_loss_bwd(
            emb.data_ptr(), n, c, inst.data_ptr(), proto.data_ptr(), P, L, sets,
            qg.data_ptr() if qg is not None else None, pg.data_ptr() if pg is not None else None, num.data_ptr(),
            den.data_ptr(), use_same.data_ptr(), gscale.data_ptr(),
            g_emb.data_ptr() if want_e else None, g_proto.data_ptr() if want_p else None,
            ws.data_ptr(), wsb, _lib.stream_ptr()))
    return (g_emb, None, g_proto, None, None, None) + (None,) * len(labels)


def _nll_sets(embeddings, instance_labels, prototypes, label_sets, groups=None):
  """label_sets: list of (semantic_labels [n], prototype_semantic_labels [P], concentration,
  mode) with mode bit 0 = 'segsort+', bit 1 = class-mask labels -> nll [L, n].  groups = (pixel_groups [n],
  prototype_groups [P]): a prototype only takes part in the sums of the pixels of its own group."""
  ops.require_gpu(embeddings, 'embeddings')
  if not 1 <= len(label_sets) <= MAX_LABEL_SETS:
    raise ValueError('1..%d label sets per pass' % MAX_LABEL_SETS)
  emb = embeddings if embeddings.dim() == 2 and embeddings.dtype is torch.float32 else \
      embeddings.reshape(-1, embeddings.shape[-1]).to(torch.float32)              # (differentiable: no detach here)
  proto = prototypes if prototypes.dim() == 2 and prototypes.dtype is torch.float32 else \
      prototypes.reshape(-1, prototypes.shape[-1]).to(torch.float32)
  inst = ops.as_rows(instance_labels, torch.int64, 1)
  sems, psems = [], []
  for ls in label_sets:
    words = (ls[3] >> 8) or 1                       # class-mask words per row (1 for plain labels)
    a = ops.as_rows(ls[0], torch.int64, 1)
    b = ops.as_rows(ls[1], torch.int64, 1)
    if a.shape[0] != emb.shape[0] * words or b.shape[0] != proto.shape[0] * words:
      raise ValueError('label vectors do not match the embeddings / prototypes')
    sems.append(a)
    psems.append(b)
  if groups is not None:
    qg = ops.as_rows(groups[0], torch.int64, 1)
    pg = ops.as_rows(groups[1], torch.int64, 1)
    if qg.shape[0] != emb.shape[0] or pg.shape[0] != proto.shape[0]:
      raise ValueError('group vectors do not match the embeddings / prototypes')
    groups = (qg, pg)
  tops = _torch_ops.ops()
  if tops is not None:                 # the torch-extension binding: one dispatch, C++ autograd node
    return tops.segsort_nll(emb, inst, proto, sems, psems, [float(ls[2]) for ls in label_sets],
                            [int(ls[3]) for ls in label_sets], groups[0] if groups is not None else None,
                            groups[1] if groups is not None else None)
  return _SegSortNLL.apply(emb, inst, proto, tuple(ls[2] for ls in label_sets),
                           tuple(ls[3] for ls in label_sets), groups, *sems, *psems)


def segsort_nll(embeddings, semantic_labels, instance_labels, prototypes, prototype_semantic_labels,
                concentration, group_mode='segsort+', pixel_groups=None, prototype_groups=None):
  """Per-pixel negative log-likelihood [num_pixels] of SegSortLoss (reference loss.py:15-82) with an
  optional restriction of every pixel to the prototypes of its own group (not in the reference: it
  compacts / loops instead, predictions/segsort.py:181-196, 224-244).  A pixel whose group holds none of
  its prototypes yields inf / nan -- mask it with `torch.where` before reducing."""
  groups = None if pixel_groups is None else (pixel_groups, prototype_groups)
  return _nll_sets(embeddings, instance_labels, prototypes,
                   [(semantic_labels, prototype_semantic_labels, float(concentration),
                     1 if group_mode == 'segsort+' else 0)], groups)[0]


def segsort_losses(embeddings, instance_labels, prototypes, label_sets, reduction='mean'):
  """Several SegSortLoss values in ONE pass over E P^T: `label_sets` is a list of up to three
  (semantic_labels, prototype_semantic_labels, concentration, group_mode) tuples that share the
  embeddings, the own-prototype indices (`instance_labels`) and the prototype table -- the three
  contrastive losses of hsg/models/predictions/hsg.py:78-155.  Returns one loss per set, each
  equal to SegSortLoss(concentration, group_mode, reduction)(embeddings, semantic_labels,
  instance_labels, prototypes, prototype_semantic_labels)."""
  sets = [(s, p, float(k), 1 if g == 'segsort+' else 0) for s, p, k, g in label_sets]
  nll = _nll_sets(embeddings, instance_labels, prototypes, sets)
  if reduction == 'mean':
    return [nll[i].mean() for i in range(len(sets))]
  if reduction == 'sum':
    return [nll[i].sum() for i in range(len(sets))]
  return [nll[i].view(-1, 1) for i in range(len(sets))]


def _calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                              prototype_semantic_labels, concentration, group_mode):
  """Reference loss.py:15-82; returns [num_pixels, 1] like the reference."""
  nll = _nll_sets(embeddings, instance_labels, prototypes,
                  [(semantic_labels, prototype_semantic_labels, float(concentration),
                    1 if group_mode == 'segsort+' else 0)])
  return nll[0].view(-1, 1)


class SegSortLoss(_Loss):
  """NCA-style pixel-to-segment contrastive loss (reference loss.py:133-190)."""

  def __init__(self, concentration=10, group_mode='segsort+', size_average=None,
               reduce=None, reduction='mean'):
    super(SegSortLoss, self).__init__(size_average, reduce, reduction)
    self.concentration = concentration
    self.group_mode = group_mode

  def __repr__(self):
    return 'SegSortLoss(concentration={:.2f}, group_mode={})'.format(
        self.concentration, self.group_mode)

  def forward(self, embeddings, semantic_labels, instance_labels, prototypes,
              prototype_semantic_labels, prototype_weights=None):
    log_likelihood = _calculate_log_likelihood(
        embeddings, semantic_labels, instance_labels, prototypes,
        prototype_semantic_labels, self.concentration, self.group_mode)
    if self.reduction == 'mean':
      return torch.mean(log_likelihood)
    if self.reduction == 'sum':
      return torch.sum(log_likelihood)
    return log_likelihood


MASK_BITS = 63          # classes per int64 mask word (the sign bit stays clear)
MAX_MASK_WORDS = 4      # include/hsgk.h HSGK_LOSS_MASK_WORDS


def _class_masks(multi_hot, what):
  """[n, num_classes] non-negative multi-hot labels -> [n, W] int64 bit masks, 63 classes
  per word (bit c % 63 of word c // 63 set iff column c is non-zero).  With non-negative
  entries the reference's label affinity `sem @ psem.T` (loss.py:108-110) is > 0 exactly
  where two masks meet and == 0 where they do not."""
  if multi_hot.dim() != 2:
    raise ValueError('%s must be [num_rows, num_classes]' % what)
  nc = multi_hot.shape[1]
  words = max(1, -(-nc // MASK_BITS))
  if words > MAX_MASK_WORDS:
    raise ValueError('%s: at most %d classes are supported (got %d)' % (what, MASK_BITS * MAX_MASK_WORDS, nc))
  if bool((multi_hot < 0).any()):
    raise ValueError('%s must be non-negative' % what)
  dev = multi_hot.device
  bits = (multi_hot != 0).to(torch.int64)
  pad = words * MASK_BITS - nc
  if pad:
    bits = torch.cat([bits, torch.zeros((bits.shape[0], pad), dtype=torch.int64, device=dev)], 1)
  weights = torch.ones((), dtype=torch.int64, device=dev) << torch.arange(MASK_BITS, dtype=torch.int64, device=dev)
  return (bits.view(-1, words, MASK_BITS) * weights).sum(dim=2).contiguous()


def _one_hot_calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                                      prototype_semantic_labels, concentration, group_mode):
  """Reference loss.py:85-130 (multi-label variant: "same semantic label" = non-zero
  label affinity); returns [num_pixels, 1] like the reference.  Same kernels as
  `_calculate_log_likelihood` with the class masks in place of the labels."""
  sem = _class_masks(semantic_labels.reshape(-1, semantic_labels.shape[-1]), 'semantic_labels')
  psem = _class_masks(prototype_semantic_labels.reshape(-1, prototype_semantic_labels.shape[-1]),
                      'prototype_semantic_labels')
  # mode bit 0 = 'segsort+', bit 1 = set mode, bits 8.. = mask words per row (include/hsgk.h)
  mode = (1 if group_mode == 'segsort+' else 0) | 2 | (sem.shape[1] << 8)
  nll = _nll_sets(embeddings, instance_labels, prototypes, [(sem, psem, float(concentration), mode)])
  return nll[0].view(-1, 1)


class SetSegSortLoss(_Loss):
  """Multi-label NCA loss (reference loss.py:193-251): semantic labels are multi-hot
  `[num_pixels, num_classes]` / `[num_prototypes, num_classes]` long tensors."""

  def __init__(self, concentration=10, group_mode='segsort+', size_average=None,
               reduce=None, reduction='mean'):
    super(SetSegSortLoss, self).__init__(size_average, reduce, reduction)
    self.concentration = concentration
    self.group_mode = group_mode

  def __repr__(self):
    return 'SetSegSortLoss(concentration={:.2f}, group_mode={})'.format(
        self.concentration, self.group_mode)

  def forward(self, embeddings, semantic_labels, instance_labels, prototypes,
              prototype_semantic_labels, prototype_weights=None):
    log_likelihood = _one_hot_calculate_log_likelihood(
        embeddings, semantic_labels, instance_labels, prototypes,
        prototype_semantic_labels, self.concentration, self.group_mode)
    if self.reduction == 'mean':
      return torch.mean(log_likelihood)
    if self.reduction == 'sum':
      return torch.sum(log_likelihood)
    return log_likelihood
