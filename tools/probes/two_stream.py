"""Potential of overlapping independent calls: K calls of segment_by_kmeans (cfg2) on one stream vs the
same K calls alternating between two streams (the prep of one call can overlap the Lloyd loop of the other)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
x = torch.randn((48, 256, 448, 448), device=dev)
K = 8
def run(streams):
  outs = [None] * len(streams)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(K):
    s = streams[i % len(streams)]
    with torch.cuda.stream(s):
      outs[i % len(streams)] = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / K * 1e3
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for name, st in (('one stream', [s0]), ('two streams', [s0, s1]), ('one stream', [s0]), ('two streams', [s0, s1])):
  run(st)
  print('%-12s %.2f ms per call' % (name, run(st)))
