"""Full-resolution inference around k-means (pyscripts/inference/prototype.py:130-177,
inference.py:155-196): the sliding-window patch grid and the overlap-averaged
accumulation of the per-crop, per-pixel L2-normalised embeddings.

The reference keeps a `[1, C, H, W]` canvas and a counter plane on the GPU and, per
crop, runs permute -> normalize_embedding (3 temporaries) -> permute -> slice `+=`; here
one kernel per crop normalises and accumulates in place (`hsgk_overlap_accumulate`), and
one divides at the end.  The result feeds `segment_by_kmeans` at full resolution.
"""
import ctypes
import math

import numpy as np
import torch

from hsg_amd import _lib, ops


def patch_end_indices(pad_size, crop_size, stride):
  """prototype.py:131-138: end index of every patch along one axis
  (`np.linspace(crop, pad, npatches, dtype=np.int32)`)."""
  npatches = math.ceil(1.0 * (pad_size - crop_size) / stride) + 1
  return np.linspace(crop_size, pad_size, npatches, dtype=np.int32)


class OverlapAverager(object):
  """canvas += normalize(crop) over a sliding window, counts += 1, canvas /= counts
  (prototype.py:141-177).  `add` takes the raw crop embedding `[1, C, h, w]` (or
  `[C, h, w]`) and the top-left corner; `result()` returns the averaged `[1, C, H, W]`."""

  def __init__(self, channels, height, width, device, eps=1e-12):
    self.C, self.H, self.W = int(channels), int(height), int(width)
    self.eps = float(eps)
    self.canvas = torch.zeros((1, self.C, self.H, self.W), dtype=torch.float32, device=device)
    self.counts = torch.zeros((1, 1, self.H, self.W), dtype=torch.float32, device=device)
    self._done = False

  def add(self, crop_emb, sh, sw):
    ops.require_gpu(crop_emb, 'crop_emb')
    if self._done:
      raise RuntimeError('OverlapAverager: result() was already taken')
    x = crop_emb.detach().to(torch.float32)
    if x.dim() == 4:
      if x.shape[0] != 1:
        raise ValueError('one crop at a time (the reference feeds batch 1)')
      x = x[0]
    x = x.contiguous()
    C, h, w = x.shape
    if C != self.C:
      raise ValueError('channel mismatch')
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().hsgk_overlap_accumulate(
          x.data_ptr(), C, h, w, self.canvas.data_ptr(), self.counts.data_ptr(), self.H, self.W,
          int(sh), int(sw), ctypes.c_float(self.eps), _lib.stream_ptr()))

  def result(self):
    if not self._done:
      with torch.cuda.device(self.canvas.device):
        _lib.check(_lib.lib().hsgk_overlap_finish(
            self.canvas.data_ptr(), self.counts.data_ptr(), self.C, self.H, self.W,
            _lib.stream_ptr()))
      self._done = True
    return self.canvas
