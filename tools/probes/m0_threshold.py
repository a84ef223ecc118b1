"""Where the fused first M-step starts to pay: ms per call by seed-cell width (run with HSGK_M0=0 / 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
for B, S in ((48, 64), (32, 96), (24, 128), (16, 160), (16, 192), (16, 224), (8, 320)):
  x = torch.randn((B, 256, S, S), device='cuda:0')
  for _ in range(3): sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10): sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  b.record(); torch.cuda.synchronize()
  print('%dx256x%dx%d, cells %4.1f px wide: %.3f ms' % (B, S, S, S / 8.0, a.elapsed_time(b) / 10))
