"""Per-image phase times of the cfg2-shaped call for batch sizes 1..48 (round 4):
if the fp16 row copy of a small group of images stays in the 256 MiB Infinity Cache between Lloyd
iterations, the E-step per image-iteration of a small batch is cheaper than 1/48 of the full batch's."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
xall = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
print('%3s %9s %9s %9s %9s %9s   per image (ms); step = wall per call / B' % ('B', 'prep', 'E', 'M', 'final', 'step'))
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 48):
  x = xall[:B]
  reps = max(3, 48 // B)
  for _ in range(2):
    sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  _lib.profile_enable(True); _lib.profile_collect()
  t0 = time.perf_counter()
  for _ in range(reps):
    sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  p = _lib.profile_collect(); _lib.profile_enable(False)
  f = 1.0 / (reps * B)
  print('%3d %9.4f %9.4f %9.4f %9.4f %9.4f' % (B, p['prep'][0] * f, p['assign'][0] * f, p['accumulate'][0] * f,
                                              p['finalize'][0] * f, dt * 1e3 / B))
