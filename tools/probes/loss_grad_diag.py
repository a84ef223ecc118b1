import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hsg_amd.utils.segsort import loss as sl
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
dev = torch.device('cuda:0')
n, c, P = 9408, 128, 1536
g = torch.Generator(device=dev).manual_seed(n + P)
proto = torch.nn.functional.normalize(torch.randn((P, c), device=dev, generator=g), dim=1)
inst = torch.randint(0, P, (n,), device=dev, generator=g)
e = torch.nn.functional.normalize(proto[inst] + 0.35 * torch.randn((n, c), device=dev, generator=g), dim=1)
psem = torch.arange(P, device=dev) % 21
sem = psem[inst].clone()
flip = torch.rand((n,), device=dev, generator=g) < 0.1
sem[flip] = (sem[flip] + 3) % 21
def ref_nll(e, sem, inst, p, psem, kappa, plus):
  s = torch.exp(torch.mm(e, p.t()) * kappa)
  same = (sem.view(-1, 1) == psem.view(1, -1)).to(e.dtype)
  own = torch.gather(s, 1, inst.view(-1, 1)).view(-1)
  same_sum = (s * same).sum(1)
  diff = (s * (1.0 - same)).sum(1)
  num = own
  if plus:
    wo = same_sum - own
    num = torch.where(wo > 0, wo, own)
  return -torch.log(num / (num + diff)), num, diff
for kappa, mode in ((16.0, 'segsort+'), (10.0, 'segsort')):
  et, pt = e.clone().requires_grad_(True), proto.clone().requires_grad_(True)
  nll = sl.segsort_nll(et, sem, inst, pt, psem, kappa, mode)
  nll.mean().backward()
  e2, p2 = e.double().requires_grad_(True), proto.double().requires_grad_(True)
  rn, num, diff = ref_nll(e2, sem, inst, p2, psem, kappa, mode == 'segsort+')
  rn.mean().backward()
  de = (et.grad.double() - e2.grad).abs()
  dp = (pt.grad.double() - p2.grad).abs()
  print(mode, 'nll max err', float((nll.double() - rn).abs().max()), 'g_emb err', float(de.max()), 'scale', float(e2.grad.abs().max()),
        'g_proto err', float(dp.max()), 'scale', float(p2.grad.abs().max()))
  row = int(de.max(1).values.argmax())
  print('   worst pixel', row, 'flipped', bool(flip[row]), 'num', float(num[row]), 'diff', float(diff[row]), 'nll', float(rn[row]), float(nll[row]))
  prow = int(dp.max(1).values.argmax())
  print('   worst proto', prow, 'pixels owning it', int((inst == prow).sum()))
