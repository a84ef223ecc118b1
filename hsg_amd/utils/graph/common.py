"""Drop-in for `hsg.utils.graph.common` on MI355X (the DMon affinity graph).

`affinity_matrix_as_attention` keeps the reference's signature
(hsg/utils/graph/common.py:39-125); the per-image / per-segment Python loop of the
reference (mask, masked_select, topk, compare, masked_fill for every segment) is one
libhsgk launch (`hsgk_knn_affinity`, csrc/graph.hip).
"""
import ctypes

import torch

from hsg_amd import _lib, ops


def inner_product_kernel(x):
  """Reference common.py:8-20: sim(i, j) = x_i^T x_j over the last two dimensions."""
  return torch.einsum('...ij,...jk->...ik', x.transpose(-2, -1), x)


def exp_inner_product_kernel(x, concentration=5):
  """Reference common.py:23-36: exp(concentration * x_i^T x_j)."""
  return inner_product_kernel(x).mul(concentration).exp()


def affinity_matrix_as_attention(x, x_padding_mask=None, x_segment_labels=None, knn=None,
                                 remove_self_loop=True, binarize=True,
                                 kernel_fn=exp_inner_product_kernel, concentration=None):
  """Reference common.py:39-125.  x `[batch, channels, num_nodes]` -> affinity
  `[batch, num_nodes, num_nodes]`.

  kernel_fn: the module's `exp_inner_product_kernel` (default, concentration 5) is
  evaluated inside the fused kernel; pass `concentration=` to change its scale (what the
  reference does with `lambda x: exp_inner_product_kernel(x, 5)`).  Any other callable is
  evaluated as given (on the GPU) and only the masking / k-NN / binarisation is fused.
  With `binarize=False` the result carries no gradient (the reference's DMon losses
  always binarise)."""
  ops.require_gpu(x, 'x')
  if not binarize and x.requires_grad:
    raise NotImplementedError('affinity_matrix_as_attention: binarize=False is forward-only here')
  B, C, N = x.shape
  dev = x.device
  xf = x.detach().to(torch.float32).contiguous()
  a_in = None
  if kernel_fn is not exp_inner_product_kernel:
    if concentration is not None:
      raise ValueError('concentration= only applies to the built-in exp_inner_product_kernel')
    a_in = kernel_fn(xf).detach().to(torch.float32).contiguous()
  conc = 5.0 if concentration is None else float(concentration)
  pad = None if x_padding_mask is None else x_padding_mask.to(torch.uint8).contiguous()
  seg = None if x_segment_labels is None else x_segment_labels.to(torch.int64).contiguous()
  with torch.cuda.device(dev):
    out = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    tmp = None if a_in is not None else torch.empty((B, N, N), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().hsgk_knn_affinity(
        xf.data_ptr(), a_in.data_ptr() if a_in is not None else None, B, C, N,
        ctypes.c_float(conc), pad.data_ptr() if pad is not None else None,
        seg.data_ptr() if seg is not None else None, int(knn) if knn is not None else 0,
        int(bool(remove_self_loop)), int(bool(binarize)),
        tmp.data_ptr() if tmp is not None else None, out.data_ptr(), _lib.stream_ptr()))
  return out
