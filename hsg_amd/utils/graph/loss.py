"""Drop-in for `hsg.utils.graph.loss` (DMon clustering losses) on MI355X.

The adjacency comes from the fused k-NN affinity kernel (graph/common.py); it is binary,
so no gradient flows through it and the pooling terms below are a handful of small
batched GEMMs on [B, num_nodes, num_clusters] tensors, left to rocBLAS through torch.
"""
import ctypes
import math

import torch
from torch.nn.modules.loss import _Loss

from hsg_amd import _lib, _torch_ops
from hsg_amd.utils.graph import common as graph_common

MAX_FUSED_CLUSTERS = 32


class _DmonPool(torch.autograd.Function):
  """Per image t = (Tr(S^T A S) - |S^T d|^2 / 2m) / 2m and c = |sum_i S_i|_2 for an adjacency without gradient:
  one libhsgk launch forward, one backward (`hsgk_dmon_pool_fwd / _bwd`, csrc/graph.hip) in place of the
  reference's chain of batched GEMMs and reductions (loss.py:62-94: 55 device operations per level and as many
  again in its backward)."""

  @staticmethod
  def forward(ctx, adj, s, valid):
    B, N, K = s.shape
    dev = s.device
    L = _lib.lib()
    with torch.cuda.device(dev):
      t = torch.empty((B,), dtype=torch.float32, device=dev)
      c = torch.empty((B,), dtype=torch.float32, device=dev)
      nbytes = L.hsgk_dmon_pool_workspace_bytes(B, N, K)
      saved = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
      _lib.check(L.hsgk_dmon_pool_fwd(adj.data_ptr(), s.data_ptr(), valid.data_ptr() if valid is not None else None,
                                      B, N, K, t.data_ptr(), c.data_ptr(), saved.data_ptr(), nbytes,
                                      _lib.stream_ptr()))
    ctx.save_for_backward(adj, s, valid, saved)
    return t, c

  @staticmethod
  def backward(ctx, gt, gc):
    adj, s, valid, saved = ctx.saved_tensors
    B, N, K = s.shape
    with torch.cuda.device(s.device):
      grad = torch.empty_like(s)
      gt = gt.to(torch.float32).contiguous()
      gc = gc.to(torch.float32).contiguous()
      _lib.check(_lib.lib().hsgk_dmon_pool_bwd(adj.data_ptr(), s.data_ptr(),
                                               valid.data_ptr() if valid is not None else None, B, N, K,
                                               saved.data_ptr(), gt.data_ptr(), gc.data_ptr(), grad.data_ptr(),
                                               _lib.stream_ptr()))
    return None, grad, None


def dmon_pool_loss(x, adj, s, mask=None, softmax=False):
  """Reference loss.py:27-96.  adj [B,N,N], s [B,N,K] (cluster assignments), mask
  [B,N] valid nodes.  Returns (dmon_loss, collapse_loss), both batch means.
      dmon     = 1 - Tr(S^T A S - S^T d d^T S / 2m) / 2m,   d = A 1,  2m = 2 sum(d)
      collapse = |sum_i S_i|_2 * sqrt(K) / N
  An adjacency without gradient (DMonLoss: the binary k-NN graph) on the GPU takes the fused kernels; one that
  carries a gradient (the pooled adjacencies of HierarchicalDMonLoss) the formulas below."""
  adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
  s = s.unsqueeze(0) if s.dim() == 2 else s
  B, N, K = s.shape
  if softmax:
    s = torch.softmax(s, dim=-1)
  kt = 4 if K <= 4 else 8 if K <= 8 else 16 if K <= 16 else 32        # the kernels' padded cluster count (LDS: 64 KiB)
  if s.is_cuda and not adj.requires_grad and K <= MAX_FUSED_CLUSTERS and (N + 3) * kt <= 14000:
    valid = None if mask is None else mask.reshape(B, N).to(torch.uint8).contiguous()
    tops = _torch_ops.ops()
    if tops is not None:               # the torch-extension binding: one dispatch, C++ autograd node
      t, c = tops.dmon_pool(adj, s, valid)
    else:
      t, c = _DmonPool.apply(adj.detach().to(torch.float32).contiguous(), s.to(torch.float32).contiguous(), valid)
    return torch.mean(1 - t), torch.mean(c) * (math.sqrt(K) / N)
  if mask is not None:
    s = s * mask.view(B, N, 1).to(s.dtype)
  st = s.transpose(1, 2)
  pooled_adj = torch.matmul(torch.matmul(st, adj), s)                    # S^T A S
  deg = adj.sum(dim=2)                                                    # d
  ddt = deg.unsqueeze(2) * deg.unsqueeze(1)                               # d d^T, as the reference forms it
  pooled_deg = torch.matmul(torch.matmul(st, ddt), s)
  two_m = 2 * deg.sum(dim=1)
  trace = torch.diagonal(pooled_adj - pooled_deg / two_m.view(-1, 1, 1), dim1=1, dim2=2).sum(-1)
  dmon_loss = torch.mean(1 - trace / two_m)
  eye_norm = torch.norm(torch.eye(K, dtype=s.dtype, device=s.device))
  collapse_loss = torch.mean(torch.norm(s.sum(dim=1), dim=1) / (N / eye_norm))
  return dmon_loss, collapse_loss


class DMonLoss(_Loss):
  """Reference loss.py:99-145."""

  def __init__(self, adj_knn=None, size_average=None, reduce=None, reduction='mean'):
    super(DMonLoss, self).__init__(size_average, reduce, reduction)
    self._knn = adj_knn

  def __repr__(self):
    return 'DMonLoss(adj_knn={})'.format(self._knn)

  def adjacency(self, x, x_padding_mask=None, x_segment_labels=None):
    """The binary k-NN graph the loss scores assignments against (no gradient): a caller that scores several
    sets of logits on the same nodes builds it once and passes it as `adjacency=`."""
    return graph_common.affinity_matrix_as_attention(
        x, x_padding_mask, x_segment_labels, self._knn, True, True, concentration=5)

  def forward(self, logits, x, x_padding_mask=None, x_segment_labels=None, adjacency=None):
    adj = self.adjacency(x, x_padding_mask, x_segment_labels) if adjacency is None else adjacency
    return dmon_pool_loss(x.transpose(1, 2), adj, logits.transpose(1, 2), ~x_padding_mask)


class HierarchicalDMonLoss(_Loss):
  """Reference loss.py:148-226: the level-l adjacency is P_{l-1} A_{l-1} P_{l-1}^T with the
  previous level's (masked) assignment probabilities."""

  def __init__(self, adj_knn=None, size_average=None, reduce=None, reduction='mean'):
    super(HierarchicalDMonLoss, self).__init__(size_average, reduce, reduction)
    self._knn = adj_knn

  def __repr__(self):
    return 'DMonLoss(adj_knn={})'.format(self._knn)

  def forward(self, probs, x, padding_masks=None, x_segment_labels=None):
    x_padding_mask = None if not padding_masks else padding_masks[0]
    adj = graph_common.affinity_matrix_as_attention(
        x, x_padding_mask, x_segment_labels, self._knn, True, True, concentration=5)
    prev_probs, prev_masks = None, None
    dmon_losses, collapse_losses = [], []
    for cur_probs, cur_masks in zip(probs, padding_masks):
      if prev_probs is not None:
        pt = prev_probs.transpose(1, 2)
        if prev_masks is not None:
          pt = pt * (~prev_masks).unsqueeze(2).to(pt.dtype)
        adj = torch.matmul(torch.matmul(pt.transpose(1, 2), adj), pt)
      d, c = dmon_pool_loss(x.transpose(1, 2), adj, cur_probs.transpose(1, 2), ~cur_masks, False)
      dmon_losses.append(d)
      collapse_losses.append(c)
      prev_probs, prev_masks = cur_probs, cur_masks
    return dmon_losses, collapse_losses


def ncut_pool_loss(x, adj, s, mask=None):
  """Reference loss.py:234-288 (normalised cut + entropy terms; no model of the reference uses it).  s [B,N,K]
  logits (the softmax is applied here), mask [B,N] valid nodes.  Returns (ncut_loss, self_loss):
      ncut = mean_b sum_k  S_k^T A (1 - S_k) / (d^T S_k + 1e-2),   d = A 1
      self = mean over (b, k) of  sum_i -S_ik log S_ik   with S clamped to [1e-5, 1]
  Since A (1 - S) = d 1^T - A S, the numerator of cluster k is d^T S_k - S_k^T (A S)_k: one batched product
  with the adjacency instead of the reference's two and its [K, K] intermediate."""
  adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
  s = s.unsqueeze(0) if s.dim() == 2 else s
  B, N, K = s.shape
  s = torch.softmax(s, dim=-1)
  if mask is not None:
    s = s * mask.view(B, N, 1).to(s.dtype)
  deg = adj.sum(dim=2, keepdim=True)                                     # d [B,N,1]
  reach = (deg * s).sum(dim=1)                                           # d^T S_k [B,K]
  inside = (s * torch.matmul(adj, s)).sum(dim=1)                         # S_k^T A S_k [B,K]
  ncut_loss = torch.mean(((reach - inside) / (reach + 1e-2)).sum(dim=1))
  safe = torch.clamp(s, min=1e-5, max=1)
  self_loss = torch.mean((-safe * torch.log(safe)).sum(dim=1))
  return ncut_loss, self_loss


class NCutLoss(_Loss):
  """Reference loss.py:291-345: the normalised-cut objective on the symmetrised binary k-NN graph."""

  def __init__(self, adj_knn=None, size_average=None, reduce=None, reduction='mean'):
    super(NCutLoss, self).__init__(size_average, reduce, reduction)
    self._knn = adj_knn

  def __repr__(self):
    return 'NCutLoss(adj_knn={:.2f})'.format(self._knn)

  def forward(self, logits, x, x_padding_mask=None, x_segment_labels=None):
    adj = graph_common.affinity_matrix_as_attention(x, x_padding_mask, x_segment_labels, self._knn, True, True)
    adj = (adj + adj.transpose(1, 2)) * 0.5
    return ncut_pool_loss(x.transpose(1, 2), adj, logits.transpose(1, 2), ~x_padding_mask)
