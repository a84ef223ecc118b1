"""segment_reduce (raw sums) at the benchmark size under different id patterns: how much of its time depends
on the number of distinct ids per 2048-row chunk."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import ops
dev = torch.device('cuda:0')
n, d = 48 * 448 * 448, 258
x = torch.randn((n, d), device=dev)
r = torch.arange(n, device=dev)
pats = {
    '1 id per image (48 ids)': r // (448 * 448),
    '4 ids per chunk, sorted': r // 512,
    '16 ids per chunk (random of 16 per image)': (r // (448 * 448)) * 16 + torch.randint(0, 16, (n,), device=dev),
    '64 ids per chunk (random of 64 per image)': (r // (448 * 448)) * 64 + torch.randint(0, 64, (n,), device=dev),
}
for name, lab in pats.items():
  P = int(lab.max()) + 1
  for _ in range(2):
    ops.segment_reduce(x, lab, P, 2)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5):
    ops.segment_reduce(x, lab, P, 2)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 5
  print('%-45s P = %6d: %.3f ms = %.2f TB/s' % (name, P, dt * 1e3, n * d * 4 / dt / 1e12))
