"""Wall time of the small-map segment_by_kmeans calls of test_small_maps_several_workgroups... under both host bindings."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
os.environ['HSGK_SMALL'] = '1'; os.environ['HSGK_M0'] = '0'
dev = torch.device('cuda:0')
shape, grid, iters = (5, 256, 28, 28), (8, 8), 10
B, C, H, W = shape
x = torch.from_numpy(synth.embeddings_nchw(synth.SEED_BASE + 5 * C + W, shape, 'iid')).to(dev)
lab = torch.from_numpy(synth.overseg_labels(synth.SEED_BASE + 11, B, H, W, regions=4, ignore_rows=3)).to(dev)
for binding in (sys.argv[1:] or ['torch', 'ctypes']):
  os.environ['HSGK_BINDING'] = binding
  for l, ign in ((lab, 255), (None, None)):
    for rep in range(3):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      o = sc.segment_by_kmeans(x, l, list(grid), ignore_index=ign, iterations=iters)
      torch.cuda.synchronize()
      print(binding, 'labels' if l is not None else 'nolabels', rep, '%.1f ms' % ((time.perf_counter() - t0) * 1e3), o[0].shape[0], flush=True)
from oracle import oracle
xs = synth.embeddings_nchw(synth.SEED_BASE + 5 * C + W, shape, 'iid')
loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
for rep in range(3):
  t0 = time.perf_counter()
  oracle.segment_by_kmeans(xs, None, grid, loc, None, iters)
  print('oracle call %.2f s' % (time.perf_counter() - t0), 'torch threads', torch.get_num_threads(), flush=True)
