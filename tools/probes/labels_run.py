"""cfg2 with a label map (over-segmentation, some ignored rows): ms per call vs the label-free call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
from hsg_amd.utils import synth
dev = torch.device('cuda:0')
B, C, H, W = 48, 256, 448, 448
x = torch.randn((B, C, H, W), device=dev)
lab = torch.from_numpy(synth.overseg_labels(5, B, H, W, regions=21, ignore_rows=12)).to(dev)
for name, kw in (('no labels', dict(labels=None, ign=None)), ('labels, ignore_index=255', dict(labels=lab, ign=255)),
                 ('labels, no ignore', dict(labels=lab, ign=None))):
  for _ in range(2):
    out = sc.segment_by_kmeans(x, kw['labels'], [8, 8], ignore_index=kw['ign'], iterations=10)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(4):
    out = sc.segment_by_kmeans(x, kw['labels'], [8, 8], ignore_index=kw['ign'], iterations=10)
  b.record(); torch.cuda.synchronize()
  print('%-26s %.2f ms per call, %d rows, %d segments' % (name, a.elapsed_time(b) / 4, out[0].shape[0], int(out[3].max()) + 1))
