"""prep kernel and step time of the cfg2 call (HIP events of the library's profiler), 3 x 5 calls."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
x = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
for _ in range(2):
  sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
torch.cuda.synchronize()
for rep in range(3):
  _lib.profile_enable(True); _lib.profile_collect()
  t0 = time.perf_counter()
  for _ in range(5):
    sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 5
  p = _lib.profile_collect()
  print('prep %.3f ms  E %.3f  M %.3f  step %.2f ms' % (p['prep'][0] / p['prep'][1], p['assign'][0] / 5, p['accumulate'][0] / 5, dt * 1e3))
