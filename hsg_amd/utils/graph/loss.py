"""Drop-in for `hsg.utils.graph.loss` (DMon clustering losses) on MI355X.

The adjacency comes from the fused k-NN affinity kernel (graph/common.py); it is binary,
so no gradient flows through it and the pooling terms below are a handful of small
batched GEMMs on [B, num_nodes, num_clusters] tensors, left to rocBLAS through torch.
"""
import torch
from torch.nn.modules.loss import _Loss

from hsg_amd.utils.graph import common as graph_common


def dmon_pool_loss(x, adj, s, mask=None, softmax=False):
  """Reference loss.py:27-96.  adj [B,N,N], s [B,N,K] (cluster assignments), mask
  [B,N] valid nodes.  Returns (dmon_loss, collapse_loss), both batch means.
      dmon     = 1 - Tr(S^T A S - S^T d d^T S / 2m) / 2m,   d = A 1,  2m = 2 sum(d)
      collapse = |sum_i S_i|_2 * sqrt(K) / N"""
  adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
  s = s.unsqueeze(0) if s.dim() == 2 else s
  B, N, K = s.shape
  if softmax:
    s = torch.softmax(s, dim=-1)
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

  def forward(self, logits, x, x_padding_mask=None, x_segment_labels=None):
    adj = graph_common.affinity_matrix_as_attention(
        x, x_padding_mask, x_segment_labels, self._knn, True, True, concentration=5)
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
