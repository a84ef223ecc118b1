"""Training-side operators at the sizes of SURVEY 8(a): ms per call (forward, forward + backward)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
from hsg_amd.utils.segsort.loss import SegSortLoss
dev = torch.device('cuda:0')

def timeit(fn, n=5):
  fn(); torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / n

g = torch.Generator(device=dev); g.manual_seed(1)
for N, P, C in ((9408, 6144, 256), (9408, 1536, 256), (200704, 3072, 256), (150528, 3072, 256)):
  e = torch.nn.functional.normalize(torch.randn((N, C), device=dev, generator=g), dim=1).requires_grad_(True)
  p = torch.nn.functional.normalize(torch.randn((P, C), device=dev, generator=g), dim=1).requires_grad_(True)
  inst = torch.randint(0, P, (N,), device=dev, generator=g)
  psem = torch.randint(0, 21, (P,), device=dev, generator=g)
  sem = psem[inst]
  loss = SegSortLoss(10, 'segsort+')
  f = timeit(lambda: loss(e.detach(), sem, inst, p.detach(), psem))
  def fb():
    l = loss(e, sem, inst, p, psem); l.backward(); e.grad = None; p.grad = None
  fbt = timeit(fb)
  fl = 2.0 * N * P * C
  print('SegSortLoss N=%d P=%d C=%d: forward %.3f ms (%.1f TFLOP/s), forward+backward %.3f ms' % (N, P, C, f, fl / f / 1e9, fbt))
# prototypes forward + backward at one image / whole batch
for N, P, C in ((200704, 64, 258), (9633792, 3072, 256)):
  x = torch.randn((N, C), device=dev, generator=g).requires_grad_(True)
  lab = (torch.arange(N, device=dev) * P // N)
  f = timeit(lambda: sc.calculate_prototypes_from_labels(x.detach(), lab, P))
  def fb():
    sc.calculate_prototypes_from_labels(x, lab, P).sum().backward(); x.grad = None
  print('calculate_prototypes_from_labels N=%d P=%d C=%d: forward %.3f ms, forward+backward %.3f ms' % (N, P, C, f, timeit(fb, 3)))
# nearest prototype / Lloyd loop / segment_mean at large sizes
from hsg_amd.utils.general import common as gc
for N, P, C in ((9633792, 64, 258), (1204224, 3072, 256), (200704, 256, 258)):
  x = torch.nn.functional.normalize(torch.randn((N, C), device=dev, generator=g), dim=1)
  p = torch.nn.functional.normalize(torch.randn((P, C), device=dev, generator=g), dim=1)
  t = timeit(lambda: sc.find_nearest_prototypes(x, p), 3)
  print('find_nearest_prototypes N=%d P=%d C=%d: %.3f ms (%.1f TFLOP/s, %.2f TB/s)' % (N, P, C, t, 2.0 * N * P * C / t / 1e9, N * C * 4 / t / 1e9))
x = torch.nn.functional.normalize(torch.randn((200704, 258), device=dev, generator=g), dim=1)
init = (torch.arange(200704, device=dev) * 64 // 200704)
print('kmeans_with_initial_labels N=200704 K=64 10 iterations: %.3f ms' % timeit(lambda: sc.kmeans_with_initial_labels(x, init, 64, 10), 3))
x = torch.randn((9633792, 32), device=dev, generator=g)
idx = (torch.arange(9633792, device=dev) * 3072 // 9633792)
print('segment_mean N=9633792 C=32 P=3072: %.3f ms' % timeit(lambda: gc.segment_mean(x, idx), 3))
sem = torch.randint(0, 21, (9633792,), device=dev, generator=g); ins = torch.randint(0, 3072, (9633792,), device=dev, generator=g)
print('prepare_prototype_labels N=9633792: %.3f ms (torch.unique)' % timeit(lambda: sc.prepare_prototype_labels(sem, ins), 2))
