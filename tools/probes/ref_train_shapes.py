"""Calls at the reference's own training hyper-parameters (bashscripts/*/train.sh: EMBEDDING_DIM = 128,
KMEANS_NUM_CLUSTERS = 4 (x4), KMEANS_ITERATIONS = 15, 448 crops -> 56 x 56 or 28 x 28 maps, a few images
per GPU): ms per call of segment_by_kmeans on both routes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
for B, C, H, W, grid, it in ((4, 128, 56, 56, [4, 4], 15), (8, 128, 56, 56, [4, 4], 15), (4, 128, 28, 28, [4, 4], 15),
                            (8, 128, 28, 28, [4, 4], 15), (24, 128, 28, 28, [4, 4], 15), (4, 128, 64, 128, [4, 4], 15)):
  x = torch.randn((B, C, H, W), device=dev)
  lab = torch.randint(0, 19, (B, H, W), device=dev) * 255 + torch.randint(0, 4, (B, H, W), device=dev)
  for route in ('0', '1'):
    os.environ['HSGK_SMALL'] = route
    res = []
    for l, ign in ((None, None), (lab, 255 * 19)):
      for _ in range(3):
        out = sc.segment_by_kmeans(x, l, grid, ignore_index=ign, iterations=it)
      torch.cuda.synchronize()
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for _ in range(20):
        out = sc.segment_by_kmeans(x, l, grid, ignore_index=ign, iterations=it)
      b.record(); torch.cuda.synchronize()
      res.append(a.elapsed_time(b) / 20)
    print('%dx%dx%dx%d grid %s %d iterations, HSGK_SMALL=%s: %.3f ms (no labels) %.3f ms (labels)' % (B, C, H, W, grid, it, route, res[0], res[1]))
del os.environ['HSGK_SMALL']
