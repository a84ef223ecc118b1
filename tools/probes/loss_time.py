"""SegSortLoss forward and forward+backward times at (N, C, P): python tools/probes/loss_time.py N C P [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hsg_amd.utils.segsort.loss import SegSortLoss
n, c, P = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
pr = torch.nn.functional.normalize(torch.randn((P, c), device=dev, generator=g), dim=1)
ins = torch.randint(0, P, (n,), device=dev, generator=g)
e = torch.nn.functional.normalize(pr[ins] + 0.35 * torch.randn((n, c), device=dev, generator=g), dim=1)
ps = torch.arange(P, device=dev) % 21
se = ps[ins]
loss = SegSortLoss(16, 'segsort+')
def timed(fn):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / reps
def fwd():
  with torch.no_grad(): loss(e, se, ins, pr, ps)
def both():
  a = e.detach().requires_grad_(True); b = pr.detach().requires_grad_(True)
  loss(a, se, ins, b, ps).backward()
  return a.grad, b.grad
tf, tb = timed(fwd), timed(both)
ga, gb = both()
print(f'N={n} C={c} P={P} fwd {tf:.3f} ms  fwd+bwd {tb:.3f} ms  bwd {tb - tf:.3f} ms  ({4 * 2.0 * n * P * c / (tb - tf) / 1e9:.1f} TF-equiv)  '
      f'grad sums {ga.double().abs().sum().item():.9e} {gb.double().abs().sum().item():.9e}')
