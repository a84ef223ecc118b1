"""cfg2 forward + backward of segment_by_kmeans (training use): ms per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
from hsg_amd.utils import synth
dev = torch.device('cuda:0')
B, C, H, W = 48, 256, 448, 448
x = torch.randn((B, C, H, W), device=dev, requires_grad=True)
lab = torch.from_numpy(synth.overseg_labels(5, B, H, W, regions=21, ignore_rows=12)).to(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
for name, labels, ign in (('no labels', None, None), ('labels + ignore', lab, 255)):
  for rep in range(3):
    a, m, b = ev(), ev(), ev()
    a.record()
    emb, eloc, _, _, _ = sc.segment_by_kmeans(x, labels, [8, 8], ignore_index=ign, iterations=10)
    m.record()
    loss = (emb * 0.5).sum() + (eloc * 0.25).sum()
    loss.backward()
    b.record(); torch.cuda.synchronize()
    x.grad = None
  print('%-16s forward %.2f ms, loss + backward %.2f ms' % (name, a.elapsed_time(m), m.elapsed_time(b)))
