"""Drop-in for `hsg.utils.segsort.loss.SegSortLoss` / `SetSegSortLoss` on MI355X.

Same constructors and forward signatures as the reference module
(hsg/utils/segsort/loss.py:133-190, 193-251).  The [N,P] similarity matrix of the
reference is never built in the forward pass: libhsgk streams the pixels
against 64-prototype blocks on fp32 MFMA and folds exp(kappa * cos) straight
into per-pixel own / same / different sums (csrc/loss.hip).  Backward
recomputes the scores, writes the per-pair weights transposed and finishes
with two library GEMMs.
"""
import ctypes

import torch
from torch.nn.modules.loss import _Loss

from hsg_amd import _lib, ops


class _SegSortNLL(torch.autograd.Function):
  """Per-pixel negative log-likelihood (reference loss.py:15-82)."""

  @staticmethod
  def forward(ctx, embeddings, semantic_labels, instance_labels, prototypes,
              prototype_semantic_labels, concentration, group_plus):
    emb = embeddings.detach().contiguous()
    proto = prototypes.detach().contiguous()
    n, c = emb.shape
    P = proto.shape[0]
    dev = emb.device
    L = _lib.lib()
    with torch.cuda.device(dev):
      nll = torch.empty((n,), dtype=torch.float32, device=dev)
      num = torch.empty((n,), dtype=torch.float32, device=dev)
      den = torch.empty((n,), dtype=torch.float32, device=dev)
      use_same = torch.empty((n,), dtype=torch.int32, device=dev)
      wsb = L.hsgk_segsort_loss_workspace_bytes(n, c, P)
      ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
      _lib.check(L.hsgk_segsort_loss_fwd(
          emb.data_ptr(), n, c, semantic_labels.data_ptr(), instance_labels.data_ptr(),
          proto.data_ptr(), P, prototype_semantic_labels.data_ptr(),
          ctypes.c_float(concentration), int(group_plus), nll.data_ptr(), num.data_ptr(),
          den.data_ptr(), use_same.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()))
    ctx.save_for_backward(emb, proto, semantic_labels, instance_labels,
                          prototype_semantic_labels, num, den, use_same)
    ctx.cfg = (float(concentration), int(group_plus))
    return nll

  @staticmethod
  def backward(ctx, gnll):
    emb, proto, sem, inst, psem, num, den, use_same = ctx.saved_tensors
    kappa, group_plus = ctx.cfg
    n, c = emb.shape
    P = proto.shape[0]
    dev = emb.device
    gscale = gnll.detach().to(torch.float32).contiguous()
    with torch.cuda.device(dev):
      wt = torch.empty((P, n), dtype=torch.float32, device=dev)
      _lib.check(_lib.lib().hsgk_segsort_loss_bwd_weights(
          emb.data_ptr(), n, c, sem.data_ptr(), inst.data_ptr(), proto.data_ptr(), P,
          psem.data_ptr(), ctypes.c_float(kappa), group_plus, num.data_ptr(), den.data_ptr(),
          use_same.data_ptr(), gscale.data_ptr(), wt.data_ptr(), _lib.stream_ptr()))
      g_emb = torch.mm(wt.t(), proto) if ctx.needs_input_grad[0] else None
      g_proto = torch.mm(wt, emb) if ctx.needs_input_grad[3] else None
    return g_emb, None, None, g_proto, None, None, None


def _calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                              prototype_semantic_labels, concentration, group_mode):
  """Reference loss.py:15-82; returns [num_pixels, 1] like the reference."""
  ops.require_gpu(embeddings, 'embeddings')
  emb = embeddings.reshape(-1, embeddings.shape[-1]).to(torch.float32)
  proto = prototypes.reshape(-1, prototypes.shape[-1]).to(torch.float32)
  sem = semantic_labels.reshape(-1).to(torch.int64).contiguous()
  inst = instance_labels.reshape(-1).to(torch.int64).contiguous()
  psem = prototype_semantic_labels.reshape(-1).to(torch.int64).contiguous()
  nll = _SegSortNLL.apply(emb, sem, inst, proto, psem, float(concentration),
                          group_mode == 'segsort+')
  return nll.view(-1, 1)


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


def _class_masks(multi_hot, what):
  """[n, num_classes] non-negative multi-hot labels -> one int64 bit mask per row
  (bit c set iff column c is non-zero).  With non-negative entries the reference's
  label affinity `sem @ psem.T` (loss.py:108-110) is > 0 exactly where two masks meet
  and == 0 where they do not."""
  if multi_hot.dim() != 2:
    raise ValueError('%s must be [num_rows, num_classes]' % what)
  nc = multi_hot.shape[1]
  if nc > 63:
    raise ValueError('%s: at most 63 classes are supported (got %d)' % (what, nc))
  if bool((multi_hot < 0).any()):
    raise ValueError('%s must be non-negative' % what)
  weights = (torch.ones((), dtype=torch.int64, device=multi_hot.device) << torch.arange(
      nc, dtype=torch.int64, device=multi_hot.device))
  return ((multi_hot != 0).to(torch.int64) * weights).sum(dim=1).contiguous()


def _one_hot_calculate_log_likelihood(embeddings, semantic_labels, instance_labels, prototypes,
                                      prototype_semantic_labels, concentration, group_mode):
  """Reference loss.py:85-130 (multi-label variant: "same semantic label" = non-zero
  label affinity); returns [num_pixels, 1] like the reference.  Same kernels as
  `_calculate_log_likelihood` with the class masks in place of the labels."""
  ops.require_gpu(embeddings, 'embeddings')
  emb = embeddings.reshape(-1, embeddings.shape[-1]).to(torch.float32)
  proto = prototypes.reshape(-1, prototypes.shape[-1]).to(torch.float32)
  sem = _class_masks(semantic_labels.reshape(-1, semantic_labels.shape[-1]), 'semantic_labels')
  psem = _class_masks(prototype_semantic_labels.reshape(-1, prototype_semantic_labels.shape[-1]),
                      'prototype_semantic_labels')
  inst = instance_labels.reshape(-1).to(torch.int64).contiguous()
  # group_plus bit 0 = 'segsort+', bit 1 = set mode (include/hsgk.h)
  nll = _SegSortNLL.apply(emb, sem, inst, proto, psem, float(concentration),
                          (1 if group_mode == 'segsort+' else 0) | 2)
  return nll.view(-1, 1)


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
