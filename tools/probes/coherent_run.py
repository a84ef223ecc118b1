"""ms per segment_by_kmeans call at cfg2 on i.i.d. input (the bench) and on a spatially coherent
input (low-resolution random field, upsampled, plus noise: closer to backbone feature maps)."""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc   # noqa: E402

dev = torch.device('cuda:0')
B, C, H, W = 48, 256, 448, 448
gen = torch.Generator(device=dev)
gen.manual_seed(7)
for name in ('iid', 'coherent'):
  if name == 'iid':
    x = torch.randn((B, C, H, W), device=dev, generator=gen)
  else:
    x = torch.empty((B, C, H, W), device=dev)
    for b in range(B):
      low = torch.randn((1, C, H // 32, W // 32), device=dev, generator=gen)
      x[b] = F.interpolate(low, size=(H, W), mode='bilinear', align_corners=False)[0]
    x += 0.05 * torch.randn((B, C, H, W), device=dev, generator=gen)
  for _ in range(2):
    sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  torch.cuda.synchronize()
  a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(5):
    out = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  b_.record()
  torch.cuda.synchronize()
  print('%-9s %.2f ms per call' % (name, a.elapsed_time(b_) / 5))
  del x, out
