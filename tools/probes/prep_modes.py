"""Is the prep kernel's per-process time scatter (6.2 .. 6.9 ms, profiles/r06_prep_ab.txt) a property of WHERE its
buffers lie?  One process, six rounds: every round frees everything, keeps a dummy allocation of a different size
(so that the caching allocator hands out other addresses), regenerates the input and times five calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
keep = []
for rnd, pad_mb in enumerate((0, 3, 64, 513, 1100, 2049)):
  torch.cuda.empty_cache()
  if pad_mb:
    keep.append(torch.empty((pad_mb << 20,), dtype=torch.uint8, device=dev))
  x = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
  for _ in range(2):
    out = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  _lib.profile_enable(True); _lib.profile_collect()
  for _ in range(5):
    out = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  p = _lib.profile_collect()
  print('round %d (+%d MB kept): prep %.3f ms   x @ %#x  emb @ %#x  emb_loc @ %#x' %
        (rnd, pad_mb, p['prep'][0] / p['prep'][1], x.data_ptr(), out[0].data_ptr(), out[1].data_ptr()), flush=True)
  del x, out
