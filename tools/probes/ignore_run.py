"""cfg2 with a large ignored fraction (checkerboard of ignored 16x16 blocks / half image): ms per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
B, C, H, W = 48, 256, 448, 448
x = torch.randn((B, C, H, W), device=dev)
yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
cases = {
  'half image ignored': torch.where(yy < H // 2, 255, (xx // 64) % 7),
  'checkerboard 16x16 ignored': torch.where(((yy // 16) + (xx // 16)) % 2 == 0, 255, (xx // 64) % 7),
  'random 50 % ignored': torch.where(torch.rand((H, W), device=dev) < 0.5, 255, 3),
}
for name, m in cases.items():
  lab = m.to(torch.int64).unsqueeze(0).expand(B, H, W).contiguous()
  for _ in range(2):
    out = sc.segment_by_kmeans(x, lab, [8, 8], ignore_index=255, iterations=10)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(4):
    out = sc.segment_by_kmeans(x, lab, [8, 8], ignore_index=255, iterations=10)
  b.record(); torch.cuda.synchronize()
  print('%-28s %.2f ms per call, %d rows kept' % (name, a.elapsed_time(b) / 4, out[0].shape[0]))
