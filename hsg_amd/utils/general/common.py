"""Drop-in for `hsg.utils.general.common` (hot-path members only).

Same names, argument meaning and return conventions as the reference
(hsg/utils/general/common.py); the arithmetic runs in libhsgk's gfx950
kernels.  Tensors must live on a ROCm device: like the reference's multi-GPU
code path, CPU tensors are not a supported device here.
"""
import torch

from hsg_amd import _lib, ops


def _require_gpu(t, name):
  if not t.is_cuda:
    raise _lib.HsgkError('%s must be a ROCm device tensor (got %s); hsg_amd has '
                         'no CPU path' % (name, t.device))


def normalize_embedding(embeddings, eps=1e-12):
  """L2-normalises the last dimension (reference general/common.py:101-120):
  x / where(||x|| >= eps, ||x||, eps).  Differentiable."""
  return ops.normalize_rows(embeddings, eps)


def segment_mean(x, index):
  """tf.segment_mean look-alike (reference general/common.py:123-147):
  per-index mean of the rows of x, empty indices give zero rows."""
  max_index = int(index.max()) + 1
  return ops.segment_reduce(x, index, max_index, 1)


def one_hot(labels, max_label=None):
  """Reference general/common.py:76-98 (bookkeeping helper, stays on ATen)."""
  if max_label is None:
    max_label = int(labels.max()) + 1
  flat = labels.reshape(-1, 1)
  out = torch.zeros((flat.shape[0], max_label), dtype=torch.long, device=labels.device)
  out.scatter_(1, flat, 1)
  return out.view(list(labels.shape) + [max_label])
