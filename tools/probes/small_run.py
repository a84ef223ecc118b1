"""Training-resolution calls (input / 16): ms per call of segment_by_kmeans, with and without labels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
for B, C, H, W, grid in ((48, 256, 28, 28, [8, 8]), (16, 256, 14, 14, [8, 8]), (24, 384, 14, 14, [8, 16]), (4, 256, 48, 48, [16, 16]), (48, 256, 56, 56, [8, 8])):
  x = torch.randn((B, C, H, W), device=dev)
  lab = torch.randint(0, 21, (B, H, W), device=dev)
  for name, l, ign in (('no labels', None, None), ('labels', lab, 255)):
    for _ in range(3):
      out = sc.segment_by_kmeans(x, l, grid, ignore_index=ign, iterations=10)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
      out = sc.segment_by_kmeans(x, l, grid, ignore_index=ign, iterations=10)
    b.record(); torch.cuda.synchronize()
    print('%dx%dx%dx%d grid %s %-9s: %.3f ms per call (%d px)' % (B, C, H, W, grid, name, a.elapsed_time(b) / 20, B * H * W))
