"""Drop-in for `hsg.utils.general.common`: the hot-path members on libhsgk, the bookkeeping / visualisation
helpers (`one_hot`, `resize_labels`, `pca`) as ATen so that the import swap of INTEGRATION.md is complete.

Same names, argument meaning and return conventions as the reference
(hsg/utils/general/common.py); the arithmetic runs in libhsgk's gfx950
kernels.  Tensors must live on a ROCm device: like the reference's multi-GPU
code path, CPU tensors are not a supported device here.
"""
import torch
import torch.nn.functional as F

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


def resize_labels(labels, size):
  """Reference general/common.py:11-26: nearest-neighbour resize of [batch, height, width] long label maps to
  `size` (what `generate_clusters`' callers do to bring the annotation to the embedding resolution, :316-321).
  Out of the hot path: ATen."""
  b, h, w = labels.shape
  resized = F.interpolate(labels.reshape(b, 1, h, w).float(), size=size, mode='nearest')
  return resized[:, 0].long()


def calculate_principal_components(embeddings, num_components=3):
  """Reference general/common.py:29-42: the leading right singular vectors [embedding_dims, num_components] of
  the centred [num_pixels, embedding_dims] matrix (visualisation helper; ATen)."""
  centred = embeddings - embeddings.mean(dim=0, keepdim=True)
  _, _, v = torch.svd(centred)
  return v[:, :num_components]


def pca(embeddings, num_components=3, principal_components=None):
  """Reference general/common.py:45-73: projects the last dimension onto `principal_components` (computed from the
  input when not given); the leading dimensions are kept."""
  lead = list(embeddings.shape[:-1])
  flat = embeddings.reshape(-1, embeddings.shape[-1])
  if principal_components is None:
    principal_components = calculate_principal_components(flat, num_components)
  return torch.mm(flat, principal_components).reshape(lead + [num_components])
