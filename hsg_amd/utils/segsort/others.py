"""Prototype memory-bank files: the wire format between `pyscripts/inference/prototype.py`
(writer, :204-208) and `hsg/utils/segsort/others.py:11-41` (reader).

One `.npy` per image holding a pickled dict `{'prototype': float32 [P, C],
'prototype_label': int64 [P]}`; a memory bank is the concatenation of all files of a
directory in sorted file-name order.  Host-side I/O only -- the tensors come from /
go to the k-means and retrieval operators of this package.
"""
import glob
import os

import numpy as np
import torch


def save_prototypes(path, prototypes, prototype_labels):
  """`np.save(name, {'prototype': ..., 'prototype_label': ...})` of
  pyscripts/inference/prototype.py:204-208 (arrays moved to the host as float32 / int64)."""
  proto = prototypes.detach().cpu().numpy() if torch.is_tensor(prototypes) else np.asarray(prototypes)
  labs = (prototype_labels.detach().cpu().numpy() if torch.is_tensor(prototype_labels)
          else np.asarray(prototype_labels))
  np.save(path, {'prototype': proto, 'prototype_label': labs})


def load_memory_banks(memory_dir):
  """Reference others.py:11-41: `[num_prototypes, C]` float tensor and `[num_prototypes]`
  long tensor over all `*.npy` files of the directory, in sorted file-name order."""
  paths = sorted(glob.glob(os.path.join(memory_dir, '*.npy')))
  if not paths:
    raise AssertionError('No memory stored in the directory')        # the reference asserts
  banks = [np.load(path, allow_pickle=True).item() for path in paths]
  protos = np.concatenate([bank['prototype'] for bank in banks], axis=0)
  labels = np.concatenate([bank['prototype_label'] for bank in banks], axis=0)
  return torch.from_numpy(protos).float(), torch.from_numpy(labels).long()
